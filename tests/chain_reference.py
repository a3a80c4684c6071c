"""numpy restatement of the chain KKT solver's block cyclic reduction (csrc/iem_kkt_device.h: kkt_eliminate / kkt_update /
kkt_forward / kkt_backward, level by level) — TEST INFRASTRUCTURE: checks kkt_chain.ChainLayout's grouping and scatter plan
on CPU, and is what the GPU kernels are compared with block for block."""
import numpy as np


def fill_blocks(layout, rows, cols, vals):
    """Dense D | B | E | G of the layout (the device keeps B restricted to its rows R / columns C: expanded here)."""
    src, dest = layout.scatter_plan(rows, cols)          # (first: it may widen the coupling, which moves the offsets)
    oD, oB, oE, oG, total = layout.offsets()
    flat = np.zeros(total)
    flat[layout.pad_positions()] = 1.0
    assert np.unique(dest).size == dest.size, "every kept KKT entry has a dense position of its own"
    flat[dest] = vals[src]
    S, nb, ne, nc = layout.S, layout.nb, layout.ne, layout.nc
    Bt = flat[oB:oE].reshape(S, nc, nc)
    B = np.zeros((S, nb, nb))
    R, C = layout.rowsR, layout.colsC
    if R.size and C.size:
        B[:, R[:, None], C[None, :]] = Bt[:, :R.size, :C.size]
    return (flat[oD:oB].reshape(S, nb, nb).copy(), B, flat[oE:oG].reshape(S, nb, ne).copy(), flat[oG:total].reshape(ne, ne).copy())


def factor(D, B, E):
    """In place on copies: returns (Dinv, X, Y, Z, Gp, negative pivots)."""
    S, nb, _ = D.shape
    ne = E.shape[2]
    D, B, E = D.copy(), B.copy(), E.copy()
    X, Y, Z, Gp = np.zeros_like(B), np.zeros_like(B), np.zeros_like(E), np.zeros((S, ne, ne))
    neg = 0

    def eliminate(i, s, final):
        nonlocal neg
        # pivots of the un-pivoted elimination = those of LDL': count the negative ones
        M = D[i].copy()
        for k in range(nb):
            neg += M[k, k] < 0
            M[k + 1:, k + 1:] -= np.outer(M[k + 1:, k], M[k, k + 1:]) / M[k, k]
        Di = np.linalg.inv(D[i])
        D[i] = Di
        if not final:
            X[i] = Di @ B[i]
            if i + s < S:
                Y[i] = Di @ B[i + s].T
        if ne:
            Z[i] = Di @ E[i]
            Gp[i] = E[i].T @ Z[i]

    s = 1
    while s < S:
        for i in range(s, S, 2 * s):
            eliminate(i, s, False)
        newB = {}
        for j in range(0, S, 2 * s):
            p, q = j - s, j + s
            if j > 0:
                D[j] -= B[j] @ Y[p]
                E[j] -= B[j] @ Z[p]
                newB[j] = -B[j] @ X[p]
            if q < S:
                D[j] -= B[q].T @ X[q]
                E[j] -= B[q].T @ Z[q]
        for j, v in newB.items():
            B[j] = v
        s *= 2
    eliminate(0, s, True)
    return D, X, Y, Z, Gp, int(neg)


def solve(Dinv, X, Y, Z, G, Gp, r, rB):
    S, nb, _ = Dinv.shape
    ne = Z.shape[2]
    r = r.copy()
    rBp = np.zeros((S, ne))
    levels = []
    s = 1
    while s < S:
        levels.append(s)
        for j in range(0, S, 2 * s):
            p, q = j - s, j + s
            if j > 0:
                r[j] -= Y[p].T @ r[p]
            if q < S:
                r[j] -= X[q].T @ r[q]
        for i in range(s, S, 2 * s):
            rBp[i] = Z[i].T @ r[i]
        s *= 2
    rBp[0] = Z[0].T @ r[0]
    xB = np.linalg.solve(G - Gp.sum(0), rB - rBp.sum(0)) if ne else np.zeros(0)
    r[0] = Dinv[0] @ r[0] - Z[0] @ xB
    for s in reversed(levels):
        for i in range(s, S, 2 * s):
            v = Dinv[i] @ r[i] - X[i] @ r[i - s] - Z[i] @ xB
            if i + s < S:
                v -= Y[i] @ r[i + s]
            r[i] = v
    return r, xB


class HubLevels:
    """Dense restatement (torch on the CPU) of what ``iem_kkt_chain_level`` / ``iem_kkt_chain_solve`` do to the chain part of
    ``kkt_chain.HubChainKKT`` — so that the hub pipeline around them (span-sparse border columns, Schur complement of the hubs)
    can be checked without a device.  ``__call__(hub, s, what)``: 0 eliminate level s, 1 update level s, 2 last block, 3 reset;
    ``__call__(hub, r, "solve")``: K_c y = r with the ORIGINAL chain matrix (kept at reset)."""

    def __call__(self, hub, s, what):
        import torch
        S, nb, nc = hub.S, hub.nb, hub.nc
        D, Bt, BR = hub.D.view(S, nb, nb), hub.Bt.view(S, nc, nc), hub.BR.view(S, nc, nc)
        R, C = hub._R, hub._C
        nR, nC = R.numel(), C.numel()
        if isinstance(what, str):
            return self._solve(hub, s)
        if what == 3:
            hub.info.zero_()
            self.D0, self.B0 = D.clone(), Bt.clone()
            return

        def invert(i):
            M = D[i].clone().numpy()
            for k in range(nb):
                hub.info[0] += int(M[k, k] < 0)
                M[k + 1:, k + 1:] -= np.outer(M[k + 1:, k], M[k, k + 1:]) / M[k, k]
            D[i] = torch.linalg.inv(D[i])
        T = hub.Tp                                   # every lane of T blocks is reduced by itself (iem_kkt_chain_level's lane_len)
        if what == 2:
            for lane in range(S // T):
                invert(lane * T)
        elif what == 0:
            for lane in range(S // T):
                for tt in range(s, T, 2 * s):
                    i = lane * T + tt
                    if tt + s < T:
                        BR[i] = Bt[i + s]
                    invert(i)
        else:
            new = {}
            for j in (lane * T + tt for lane in range(S // T) for tt in range(0, T, 2 * s)):
                p, q = j - s, j + s
                tt = j % T
                if tt > 0:
                    Bj = Bt[j][:nR, :nC]
                    D[j][R[:, None], R[None, :]] -= Bj @ D[p][C[:, None], C[None, :]] @ Bj.T
                    nb_ = torch.zeros(nc, nc, dtype=torch.float64)
                    if tt - 2 * s >= 0:
                        nb_[:nR, :nC] = -Bj @ D[p][C[:, None], R[None, :]] @ Bt[p][:nR, :nC]
                    new[j] = nb_
                if tt + s < T:
                    Bq = Bt[q][:nR, :nC]
                    D[j][C[:, None], C[None, :]] -= Bq.T @ D[q][R[:, None], R[None, :]] @ Bq
            for j, v in new.items():
                Bt[j] = v

    def _solve(self, hub, r):
        import scipy.sparse as sp
        from scipy.sparse.linalg import spsolve
        import torch
        S, nb = hub.S, hub.nb
        R, C = hub._R.numpy(), hub._C.numpy()
        blocks = [[None] * S for _ in range(S)]
        for k in range(S):
            blocks[k][k] = sp.csr_matrix(self.D0[k].numpy())
            if k > 0:
                B = np.zeros((nb, nb))
                B[R[:, None], C[None, :]] = self.B0[k][:R.size, :C.size].numpy()
                if np.any(B):
                    blocks[k][k - 1] = sp.csr_matrix(B)
                    blocks[k - 1][k] = sp.csr_matrix(B.T)
        K = sp.bmat(blocks, format="csc")
        return torch.as_tensor(spsolve(K, r.numpy()))

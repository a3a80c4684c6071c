// iem_kkt_device.h — hand-written gfx950 kernels of the CHAIN KKT solver (the step right after
// jac_coord!/hess_coord! in every solver iteration; reference: README.md:36-37 hands it to MadNLPGPU + CUDSS).
//
// The augmented system K = [H + Sigma + dw I, J'; J, -dc I] of a transcription whose supports couple only through a
// derivative stencil (src/transform.jl:535-557: reach 1 for backward differences) is BLOCK TRIDIAGONAL once its
// unknowns are grouped by support — block k = the variables and constraint rows of support(s) k, NB unknowns, ordered
// variables first — plus a small dense BORDER (finite / first-stage variables and their rows, NE unknowns):
//
//        D_k = K[k, k]  (NB x NB, symmetric)      B_k = K[k, k-1]  (NB x NB)      E_k = K[k, border]  (NB x NE)
//
// The coupling is NARROW: B_k has entries only on a few rows R of block k (the derivative-approximation rows) and a few
// columns C of block k - 1 (the differentiated states) — 9 x 9 of 40 x 40 for the quadrotor — and block cyclic reduction
// PRESERVES that: eliminating a block updates its neighbours on R x R and C x C only, and the new coupling is again an
// R x C block.  So a block is D_k (NB x NB) plus Bt_k = B_k[R, C] (NC x NC), and the factorisation moves one dense
// block per support through HBM instead of four (the dense form — D, B, D^-1 B, D^-1 B' — was bound by exactly that
// traffic: 18 GB per factorisation at 1e5 quadrotor supports, profiles/r03_kkt_chain.json).
//
// (kkt_chain.py builds the grouping from the model's slab table and the Jacobian structure, finds R and C, and scatters the
// KKT values into D | Bt | E | G every iteration.)  Factorisation = block CYCLIC REDUCTION: at level l (stride s = 2^l) the
// blocks i = (2t+1) s are eliminated in parallel — kkt_eliminate inverts D_i in place (block Gauss-Jordan on the FP64
// matrix cores, no pivoting, variables first: K is quasi-definite under the interior-point regularisation, so every pivot
// order is admissible; the signs of the pivots are counted — the inertia an interior-point method asks for), forms
// Z_i = D_i^-1 E_i and the block's border term, and keeps BR_i = Bt_{i+s}; kkt_update folds the inverses into the surviving
// neighbours j = 2t s (p = j - s, q = j + s):
//        D_j[R,R] -= Bt_j Dp[C,C] Bt_j'     D_j[C,C] -= Bt_q' Dq[R,R] Bt_q     Bt_j <- -Bt_j Dp[C,R] Bt_p  (the coupling j <- j - 2s)
// ceil(log2 S) levels, two launches each, every block a workgroup; the border's Schur complement G - sum_i E_i' Z_i is
// accumulated per block (one partial per block, summed once: deterministic).  Solve: the same levels forward on the
// right-hand side (kkt_forward: z_i = D_i^-1 r_i, survivors take the R / C entries of their neighbours' z), the border
// system (tiny, dense), the levels backward (kkt_backward).  FP64 throughout; no atomics on floating-point data.
//
// Compile-time: KKT_NB (block size, multiple of 4, <= 96), KKT_NE (border size, 0 or a multiple of 4, <= 128), KKT_NC
// (coupling rows / columns, multiple of 4, <= 48).
#ifndef IEM_KKT_DEVICE_H
#define IEM_KKT_DEVICE_H

#ifndef KKT_NB
#define KKT_NB 40
#endif
#ifndef KKT_NE
#define KKT_NE 0
#endif
#ifndef KKT_NC
#define KKT_NC 12
#endif
#define KKT_LN (KKT_NC + 1)       // LDS row stride of an NC x NC coupling block
// The dense block work runs on the FP64 matrix cores: v_mfma_f64_16x16x4_f64 (77 TFLOP/s measured on MI355X against 59 for
// the vector FMA — and a quarter of the operand traffic of a 4 x 4 register tiling, whose LDS reads bound the first form of
// these kernels: tools/probes/mfma_f64_probe.hip, profiles/r03_kkt_chain.json).  An NB x NB matrix is cut into R x R tiles of
// 16 x 16 (R = ceil(NB / 16), the rim padded with zeros); wave w of the workgroup owns the tile rows w, w + W, ...
// (W = min(R, KKT_WMAX) waves; the host picks KKT_WMAX per shape — kkt_shape in csrc/iem_api.cpp: one or two waves holding
// many tiles each for blocks without a border).  Within a tile lane l holds, in register r of its accumulator, the element
//        row = (l >> 4) + 4 r ,  col = l & 15            (the instruction's C/D layout),
// and feeds A[row = l & 15][k = l >> 4] and B[k = l >> 4][col = l & 15] per step of four of the inner dimension.
#define KKT_R ((KKT_NB + 15) / 16)
#define KKT_RE ((KKT_NE + 15) / 16)
#ifndef KKT_WMAX
#define KKT_WMAX 4
#endif
#define KKT_W (KKT_R < KKT_WMAX ? KKT_R : KKT_WMAX)
#define KKT_T (64 * KKT_W)        // threads per workgroup
#define KKT_TRW ((KKT_R + KKT_W - 1) / KKT_W)   // tile rows per wave
#define KKT_LD (KKT_NB + 1)       // LDS row stride of an NB-wide matrix (odd: rows of a fragment fall into different banks)
#define KKT_LE (KKT_NE + 1)
typedef double kkt_d4 __attribute__((ext_vector_type(4)));

// acc[n][tc] += A[rows of tile row w + n W][:] * Bm[:][cols of tile column tc]      (A: NB x NB in LDS, stride lda; Bm: NB x ncols, stride ldb)
template <int RC>
__device__ __forceinline__ void kkt_mma(kkt_d4 (&acc)[KKT_TRW][RC], const double *A, int lda, const double *Bm, int ldb, int ncols) {
  const int l = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6, li = l & 15, lk = l >> 4;
#pragma unroll 2
  for (int k0 = 0; k0 < KKT_NB; k0 += 4) {
    double b[RC];
#pragma unroll
    for (int tc = 0; tc < RC; ++tc) {
      const int j = 16 * tc + li;
      b[tc] = j < ncols ? Bm[(k0 + lk) * ldb + j] : 0.0;
    }
#pragma unroll
    for (int n = 0; n < KKT_TRW; ++n) {
      const int tr = w + n * KKT_W;
      if (tr >= KKT_R) continue;
      const int i = 16 * tr + li;
      const double a = i < KKT_NB ? A[i * lda + k0 + lk] : 0.0;
#pragma unroll
      for (int tc = 0; tc < RC; ++tc) acc[n][tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[tc], acc[n][tc], 0, 0, 0);
    }
  }
}
template <int RC>
__device__ __forceinline__ void kkt_zero(kkt_d4 (&acc)[KKT_TRW][RC]) {
#pragma unroll
  for (int n = 0; n < KKT_TRW; ++n)
#pragma unroll
    for (int tc = 0; tc < RC; ++tc) acc[n][tc] = kkt_d4{0.0, 0.0, 0.0, 0.0};
}
// G (global, row-major, `ncols` wide) = sign * acc      /      G -= acc
template <int RC, bool SUBTRACT>
__device__ __forceinline__ void kkt_put(double *__restrict__ G, const kkt_d4 (&acc)[KKT_TRW][RC], double sign, int ncols) {
  const int l = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6, li = l & 15, lk = l >> 4;
#pragma unroll
  for (int n = 0; n < KKT_TRW; ++n) {
    const int tr = w + n * KKT_W;
    if (tr >= KKT_R) continue;
#pragma unroll
    for (int tc = 0; tc < RC; ++tc) {
      const int j = 16 * tc + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * tr + lk + 4 * r;
        if (i < KKT_NB && j < ncols) {
          double *g = G + (long long)i * ncols + j;
          if (SUBTRACT) *g -= acc[n][tc][r]; else *g = sign * acc[n][tc][r];
        }
      }
    }
  }
}
// the same tiles into LDS (stride ld)
template <int RC>
__device__ __forceinline__ void kkt_put_lds(double *L, int ld, const kkt_d4 (&acc)[KKT_TRW][RC], int ncols) {
  const int l = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6, li = l & 15, lk = l >> 4;
#pragma unroll
  for (int n = 0; n < KKT_TRW; ++n) {
    const int tr = w + n * KKT_W;
    if (tr >= KKT_R) continue;
#pragma unroll
    for (int tc = 0; tc < RC; ++tc) {
      const int j = 16 * tc + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * tr + lk + 4 * r;
        if (i < KKT_NB && j < ncols) L[i * ld + j] = acc[n][tc][r];
      }
    }
  }
}
// global (row-major, COLS wide) -> LDS (stride ld), optionally transposed.  All of a thread's loads are ISSUED before the
// first one is consumed (a load -> store loop waits a full HBM round trip per element).
template <int ROWS, int COLS>
__device__ __forceinline__ void kkt_load_t(double *lds, const double *__restrict__ g, int ld, bool transpose) {
  constexpr int N = (ROWS * COLS + KKT_T - 1) / KKT_T;
  double tmp[N];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    const int e = (int)threadIdx.x + n * KKT_T;
    tmp[n] = e < ROWS * COLS ? g[e] : 0.0;
  }
#pragma unroll
  for (int n = 0; n < N; ++n) {
    const int e = (int)threadIdx.x + n * KKT_T;
    if (e < ROWS * COLS) {
      const int r = e / COLS, c = e - r * COLS;
      if (transpose) lds[c * ld + r] = tmp[n]; else lds[r * ld + c] = tmp[n];
    }
  }
}
#define kkt_load(lds, g, rows, cols, ld, transpose) kkt_load_t<rows, cols>(lds, g, ld, transpose)
// 1 / x: hardware reciprocal + two Newton steps (full double precision for the normal range the pivots live in; the
// IEEE division sequence is three times as many FP64 instructions on the critical path)
__device__ __forceinline__ double kkt_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = r + r * (1.0 - x * r);
  r = r + r * (1.0 - x * r);
  return r;
}

struct KktArgs {
  double *D, *Bt, *BR, *E, *Z, *Gp;
  const int *rows, *cols;   // KKT_NC each: local rows / columns of the coupling (-1: padding)
  long long *info;          // [0] negative pivots, [1] pivots below the threshold (replaced by +-tiny: the factorisation is not to be trusted)
  long long S, s;           // blocks, stride of this level
  int final_block;          // 1: the last remaining block of every lane (index 0 of the lane), no chain neighbours
  double tiny;              // pivot threshold
  long long T;              // blocks per LANE (0: one chain of S blocks) — see kkt_eliminated / kkt_survivor
};

// The chain may be cut into LANES of T blocks each (S = lanes x T; kkt_chain.HubChainKKT: one chain per scenario of a 2-D support
// grid), every lane reduced by the same levels: block t of a lane is eliminated at the level s with t mod 2s == s, whatever T
// is — the level structure is the same in every lane, which the span bookkeeping of the hubs relies on.  (One chain with zero
// couplings at the seams, T = S, reduces to the same thing only when T is a power of two.)
struct KktIdx { long long i; bool left, right, valid; };
__device__ __forceinline__ KktIdx kkt_eliminated(long long b, long long s, long long S, long long T_) {   // b-th eliminated block of level s
  const long long T = T_ > 0 ? T_ : S, ne = (T - s + 2 * s - 1) / (2 * s);
  const long long lane = ne > 0 ? b / ne : 0, t = (2 * (b - lane * ne) + 1) * s;
  return KktIdx{lane * T + t, true, t + s < T, ne > 0 && t < T && lane * T + t < S};
}
__device__ __forceinline__ KktIdx kkt_survivor(long long b, long long s, long long S, long long T_) {     // b-th surviving block of level s
  const long long T = T_ > 0 ? T_ : S, ns = (T + 2 * s - 1) / (2 * s);
  const long long lane = b / ns, t = 2 * (b - lane * ns) * s;
  return KktIdx{lane * T + t, t > 0, t + s < T, lane * T + t < S};
}
__device__ __forceinline__ long long kkt_lane_first(long long b, long long S, long long T_) { return b * (T_ > 0 ? T_ : S); }   // final_block 1: block 0 of lane b

// Panel P (rows / columns 4P .. 4P + 3) of the in-place Gauss-Jordan inverse of the matrix held in the accumulator layout.
// The four pivot steps of the panel are REGROUPED, not reformulated: with c_i(k) the entry of row i in pivot column k at
// the time of step k and r_k the normalised pivot row at that time, the scalar algorithm does  M[i, :] -= c_i(k) r_k  for
// k = 4P .. 4P + 3 — one v_mfma_f64_16x16x4 per tile (k = 4 is the instruction's inner dimension) once the c_i(k) and r_k
// are known, and they follow from the 4 x 4 pivot block alone: every lane runs its four scalar steps (pivots in the given
// order: K is quasi-definite, their signs are the inertia), carries them over its own column of the row panel (4 values)
// and its own row of the column panel (4 values), and feeds the matrix cores.  Forming (pivot block)^-1 explicitly and
// multiplying with it costs four digits on blocks with -delta_c pivots (1e6 entries next to O(1) ones) — the regrouped form
// does the scalar algorithm's operations on the scalar algorithm's numbers.
// The row panel already sits in the B-operand layout of its owner wave (register P % 4 of tile row P / 4), the column
// panel is transposed through LDS: two buffers alternate, so that one barrier per panel is enough.  P is a template
// parameter: every register index is a compile-time constant.
#define KKT_LC 5                  // LDS row stride of the column panel (16 R rows: the rim is read, never used)
#define KKT_LP (16 * KKT_R + 1)   // ... of the 4-row row panel
#define KKT_PANEL_DOUBLES (2 * 16 * KKT_R * KKT_LC + 2 * 4 * KKT_LP)
#define KKT_GJ_DOUBLES 32          // per panel: snap[4][4] (the normalised pivot rows), then its lower triangle (zeros above the diagonal)
template <int P>
__device__ __forceinline__ void kkt_gj_panel(kkt_d4 (&m)[KKT_TRW][KKT_R], double *cpb, double *rpb, double *gjb, double tiny, int &neg_, int &bad_) {
  constexpr int TS = P / 4, RS = P % 4, C0 = 4 * RS;      // tile row / column of the panel, register slot of its rows, first local column
  constexpr int WS = TS % KKT_W, NS = TS / KKT_W;         // owner wave and slot of the panel's tile row
  double *Cp = cpb + (P & 1) * (16 * KKT_R * KKT_LC), *Rp = rpb + (P & 1) * (4 * KKT_LP), *Gj = gjb + (P & 1) * KKT_GJ_DOUBLES;
  const int l = (int)threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), li = l & 15, lk = l >> 4;
  const bool pcol = li >= C0 && li < C0 + 4;              // this lane holds panel columns (in tile column TS)
  if (pcol) {
#pragma unroll
    for (int n = 0; n < KKT_TRW; ++n) {
      const int tr = w + n * KKT_W;
      if (tr >= KKT_R) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) Cp[(16 * tr + lk + 4 * r) * KKT_LC + li - C0] = m[n][TS][r];
    }
  }
  // The OWNER wave of the panel's rows runs the four pivot steps on the 4 x 4 pivot block — once per block, not once per
  // wave: it collects the block from its own lanes (element (a, b) sits in lane 16 a + C0 + b), every lane of the wave then
  // holds the same numbers; mult[a][k] = the entry (a, k) at the time of step k, inv[k] = 1 / pivot k, snap[k][b] = the
  // normalised pivot row k at the time of step k (b < k: what the inverse holds there by then; b = k: 1 / pivot; b > k: the
  // entry the later pivot columns still see).  The other waves get `snap` through LDS, behind the panel's one barrier.
  double mult[4][4], inv[4];
  if (w == WS) {
#pragma unroll
    for (int tc = 0; tc < KKT_R; ++tc) Rp[lk * KKT_LP + 16 * tc + li] = m[NS][tc][RS];
    const double own = m[NS][TS][RS];
    const int olo = __double2loint(own), ohi = __double2hiint(own);
    double pv[4][4], snap[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        pv[a][b] = __hiloint2double(__builtin_amdgcn_readlane(ohi, 16 * a + C0 + b), __builtin_amdgcn_readlane(olo, 16 * a + C0 + b));
    int neg = 0, bad = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double piv = pv[k][k];
      if (piv < 0.0) ++neg;
      if (!(fabs(piv) >= tiny)) { ++bad; piv = piv < 0.0 ? -tiny : tiny; }   // keep going with a bounded pivot; info[1] reports it
      inv[k] = kkt_rcp(piv);
#pragma unroll
      for (int b = 0; b < 4; ++b) if (b != k) pv[k][b] *= inv[k];
      pv[k][k] = inv[k];
#pragma unroll
      for (int b = 0; b < 4; ++b) snap[k][b] = pv[k][b];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (a == k) { mult[a][k] = 0.0; continue; }
        const double f = pv[a][k];
        mult[a][k] = f;
#pragma unroll
        for (int b = 0; b < 4; ++b) if (b != k) pv[a][b] -= f * pv[k][b];
        pv[a][k] = -f * inv[k];
      }
    }
    if (l == 0) {
      neg_ += neg; bad_ += bad;      // (one wave per panel, a barrier between panels)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int b = 0; b < 4; ++b) { Gj[4 * k + b] = snap[k][b]; Gj[16 + 4 * k + b] = b <= k ? snap[k][b] : 0.0; }
    }
  }
  __syncthreads();
  // B operand = entry j of the normalised pivot row lk at the time of ITS step: a combination of the row panel's column j
  // with row lk of the lower triangle of `snap` (the steps before it, already folded).  The panel's own columns enter as
  // unit vectors (the in-place algorithm keeps the inverse where the identity of [A | I] would sit).
  // (also in a one-wave workgroup: keeping `snap` in registers across the panel instead costs more than the LDS round trip —
  // 2.52 against 2.37 ms at 1e5 quadrotor supports)
  double sl[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) sl[b] = Gj[16 + 4 * lk + b];
  double bop[KKT_R], vin[KKT_R][4];
#pragma unroll
  for (int tc = 0; tc < KKT_R; ++tc) {
#pragma unroll
    for (int a = 0; a < 4; ++a) vin[tc][a] = Rp[a * KKT_LP + 16 * tc + li];
    if (tc == TS) {
#pragma unroll
      for (int a = 0; a < 4; ++a) vin[tc][a] = pcol ? (li - C0 == a ? 1.0 : 0.0) : vin[tc][a];
    }
    bop[tc] = sl[0] * vin[tc][0] + sl[1] * vin[tc][1] + sl[2] * vin[tc][2] + sl[3] * vin[tc][3];
  }
  // the panel's own columns restart from zero; its rows (owner wave) become the finished row panel: the four steps carried
  // over each column, one after the other as the scalar algorithm does
  if (pcol) {
#pragma unroll
    for (int n = 0; n < KKT_TRW; ++n) m[n][TS] = kkt_d4{0.0, 0.0, 0.0, 0.0};
  }
  if (w == WS) {
#pragma unroll
    for (int tc = 0; tc < KKT_R; ++tc) {
      double v[4] = {vin[tc][0], vin[tc][1], vin[tc][2], vin[tc][3]};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] *= inv[k];
#pragma unroll
        for (int a = 0; a < 4; ++a) if (a != k) v[a] -= mult[a][k] * v[k];
      }
      m[NS][tc][RS] = lk == 0 ? v[0] : lk == 1 ? v[1] : lk == 2 ? v[2] : v[3];
    }
  }
  // every other row i: its entries c_i(k) in the pivot columns at the time of step k, then M[i, :] -= sum_k c_i(k) r_k
  const double u01 = Gj[1], u02 = Gj[2], u03 = Gj[3], u12 = Gj[6], u13 = Gj[7], u23 = Gj[11];
#pragma unroll
  for (int n = 0; n < KKT_TRW; ++n) {
    const int tr = w + n * KKT_W;
    if (tr >= KKT_R) continue;
    const int i = 16 * tr + li;
    double c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = Cp[i * KKT_LC + k];
    c[1] -= c[0] * u01;
    c[2] -= c[0] * u02; c[2] -= c[1] * u12;
    c[3] -= c[0] * u03; c[3] -= c[1] * u13; c[3] -= c[2] * u23;
    double a = -(lk == 0 ? c[0] : lk == 1 ? c[1] : lk == 2 ? c[2] : c[3]);
    if (i >= 4 * P && i < 4 * P + 4) a = 0.0;
#pragma unroll
    for (int tc = 0; tc < KKT_R; ++tc) m[n][tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bop[tc], m[n][tc], 0, 0, 0);
  }
}
template <int P>
__device__ __forceinline__ void kkt_gj_all(kkt_d4 (&m)[KKT_TRW][KKT_R], double *cpb, double *rpb, double *gjb, double tiny, int &neg_, int &bad_) {
  if constexpr (P < KKT_NB / 4) {
    kkt_gj_panel<P>(m, cpb, rpb, gjb, tiny, neg_, bad_);
    kkt_gj_all<P + 1>(m, cpb, rpb, gjb, tiny, neg_, bad_);
  }
}

// ---- level l, step 1: eliminate the blocks i = (2 t + 1) s --------------------------------------------------
#ifdef KKT_WPE
#define KKT_OCC __attribute__((amdgpu_waves_per_eu(KKT_WPE, KKT_WPE)))   // register budget for KKT_WPE waves per SIMD: the kernel is bound by latency x occupancy
#else
#define KKT_OCC
#endif
#if defined(KKT_ROWWISE) && KKT_NE == 0
// Blocks that fit the lanes of ONE wave (NB <= 48, no border): lane = row(s) of the block, their NB entries in registers, and
// several blocks side by side in a wave.  The in-place Gauss-Jordan inverse is then
// the scalar algorithm itself — pivot k: the pivot row goes through LDS to every lane of its block (broadcast reads), each lane
// does  m[j] -= (m[k] / p) r[j]  on its own row: NB fused multiply-adds per lane and step, no panel regrouping, no tile rim,
// no transposes.  The matrix-core form above spends 590 instructions per 4 pivots around 9 MFMAs whatever the block size (a
// 20 x 20 block costs what a 40 x 40 one does) and is bound by FP64 issue slots at these sizes; this one issues ~ 3 NB per pivot.
// Same pivots in the same order, same pivot-sign count.
// KKT_RPL rows per lane (host-chosen, KKT_ROWWISE = 1 or 2): a lane holds the rows li, li + LPB, ... of its block (LPB = NB / RPL lanes
// per block, 64 / LPB blocks per wave).  What every lane must READ per pivot — the pivot row, NB doubles through LDS — is then shared
// by RPL rows: the kernel is bound by the CU's LDS pipe (16 waves share it; profiles/r04_kkt_pmc_hub.txt), so two rows per lane
// nearly double its rate, and bring 40 x 40 blocks (20 lanes each, three per wave) within reach of this form.
#define KKT_RPL KKT_ROWWISE
#define KKT_LPB (KKT_NB / KKT_RPL)
// waves per workgroup (host-chosen): 40 lanes per block leave 24 of a wave idle, three blocks share the 128 lanes of two waves
#ifndef KKT_ROW_WPG
#define KKT_ROW_WPG 1
#endif
#define KKT_ROW_T (64 * KKT_ROW_WPG)
#define KKT_BPW (KKT_ROW_T / KKT_LPB)
// pivot K (a template parameter: every register index is a compile-time constant whatever the unroller's thresholds say)
template <int K>
__device__ __forceinline__ void kkt_row_steps(double (&m)[KKT_RPL][KKT_NB], double (*rowbuf)[KKT_BPW * KKT_NB], int bs, int li, bool mine, double tiny, int &neg, int &bad) {
  if constexpr (K < KKT_NB) {
    // The pivot row reaches the lanes as the pivot COLUMN: the block is symmetric, and the in-place elimination keeps it so up to
    // the sign of the columns already processed (M[k][j] = -M[j][k] for j < k, = M[j][k] for j > k) — every lane posts the entries
    // it holds of column k instead of the row's owner posting NB entries through masked stores.  (Row and column agree to rounding
    // only: the elimination runs on a matrix perturbed at that level.)
    constexpr int KL = K % KKT_LPB, KH = K / KKT_LPB;          // lane and register set of row K
    double *rb = rowbuf[K & 1] + bs * KKT_NB;
    if (mine) {
#pragma unroll
      for (int h = 0; h < KKT_RPL; ++h) rb[li + h * KKT_LPB] = m[h][K];
    }
    // the pivot row restarts from zero (it becomes r / p, its pivot entry 1 / p)
#if KKT_RPL == 1 && KKT_NB > 32
    {   // ... as an in-place product with 0 or 1: with 40 entries per lane the divergent assignment below costs the allocator a second
        // copy of the row while it lasts (40 x 40 at 1e5 supports: 2.24 ms against 2.35; at 20 x 20 the assignment is the faster form)
      const double keep = li == KL ? 0.0 : 1.0;
#pragma unroll
      for (int j = 0; j < KKT_NB; ++j) m[KH][j] *= keep;
    }
#else
    if (li == KL) {
#pragma unroll
      for (int j = 0; j < KKT_NB; ++j) m[KH][j] = 0.0;
    }
#endif
    __syncthreads();       // (one wave: the barrier orders the LDS traffic; two buffers alternate, so one per pivot is enough)
    double piv = rb[K];
    if (piv < 0.0) ++neg;
    if (!(fabs(piv) >= tiny)) { ++bad; piv = piv < 0.0 ? -tiny : tiny; }
    const double inv = kkt_rcp(piv);
    double g[KKT_RPL];
#pragma unroll
    for (int h = 0; h < KKT_RPL; ++h) g[h] = (h == KH && li == KL) ? -inv : m[h][K] * inv;
#pragma unroll
    for (int j0 = 0; j0 < KKT_NB; j0 += 4) {      // the pivot row four entries at a time: it never sits in registers as a whole
      const double2 v0 = *reinterpret_cast<const double2 *>(rb + j0), v1 = *reinterpret_cast<const double2 *>(rb + j0 + 2);
      const double r4[4] = {v0.x, v0.y, v1.x, v1.y};
#pragma unroll
      for (int h = 0; h < KKT_RPL; ++h)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int j = j0 + jj;
          if (j != K) m[h][j] = __builtin_fma(j < K ? g[h] : -g[h], r4[jj], m[h][j]);      // (row entry = -column entry for j < K)
        }
#if KKT_RPL > 1
      __builtin_amdgcn_sched_barrier(0);      // (keeps the scheduler from issuing every read of the row up front: NB more doubles live)
#endif
    }
#pragma unroll
    for (int h = 0; h < KKT_RPL; ++h) m[h][K] = -g[h];
    kkt_row_steps<K + 1>(m, rowbuf, bs, li, mine, tiny, neg, bad);
  }
}
// register budget: the rows (2 VGPRs per entry) + 112 for the step (with less the allocator spills: two rows of 20 under a three-wave
// budget run at HALF the rate, profiles/r04_kkt_rpl_ab.txt); waves per SIMD from that
#ifndef KKT_ROW_WAVES
#define KKT_ROW_WAVES (512 / (2 * KKT_RPL * KKT_NB + 112) < 1 ? 1 : 512 / (2 * KKT_RPL * KKT_NB + 112) > 8 ? 8 : 512 / (2 * KKT_RPL * KKT_NB + 112))
#endif
extern "C" __global__ __launch_bounds__(KKT_ROW_T) __attribute__((amdgpu_waves_per_eu(KKT_ROW_WAVES, KKT_ROW_WAVES))) void kkt_eliminate(const KktArgs A) {
  __shared__ double rowbuf[2][KKT_BPW * KKT_NB];
  __shared__ int cnt[2];
  const int lane = (int)threadIdx.x, slot = lane / KKT_LPB, li = lane - slot * KKT_LPB;
  const long long b = (long long)blockIdx.x * KKT_BPW + slot;
  const KktIdx ix = kkt_eliminated(b, A.s, A.S, A.T);
  const long long i = A.final_block == 2 ? b : A.final_block ? kkt_lane_first(b, A.S, A.T) : ix.i;
  const bool mine = slot < KKT_BPW, on = mine && i < A.S && (A.final_block || ix.valid);      // (the lanes beyond the last whole block shadow slot 0: they read, never write)
  const int bs = mine ? slot : 0;
  if (lane < 2) cnt[lane] = 0;
  double m[KKT_RPL][KKT_NB];
  double *Di = A.D + (on ? i : 0) * KKT_NB * KKT_NB;
#pragma unroll
  for (int h = 0; h < KKT_RPL; ++h) {
    const int row = li + h * KKT_LPB;
#pragma unroll
    for (int j = 0; j < KKT_NB; j += 2) {
      const double2 v = on ? *reinterpret_cast<const double2 *>(Di + row * KKT_NB + j) : double2{j == row ? 1.0 : 0.0, j + 1 == row ? 1.0 : 0.0};
      m[h][j] = v.x; m[h][j + 1] = v.y;
    }
  }
  if (on && !A.final_block && ix.right)
    for (int e = li; e < KKT_NC * KKT_NC; e += KKT_LPB) A.BR[i * KKT_NC * KKT_NC + e] = A.Bt[(i + A.s) * KKT_NC * KKT_NC + e];
  int neg = 0, bad = 0;
  kkt_row_steps<0>(m, rowbuf, bs, li, mine, A.tiny, neg, bad);
  if (on && li == 0) { if (neg) atomicAdd(&cnt[0], neg); if (bad) atomicAdd(&cnt[1], bad); }
  if (on) {
#pragma unroll
    for (int h = 0; h < KKT_RPL; ++h)
#pragma unroll
      for (int j = 0; j < KKT_NB; j += 2) *reinterpret_cast<double2 *>(Di + (li + h * KKT_LPB) * KKT_NB + j) = double2{m[h][j], m[h][j + 1]};
  }
  __syncthreads();
  if (lane < 2 && cnt[lane]) atomicAdd((unsigned long long *)(A.info + lane), (unsigned long long)cnt[lane]);
}
#else
extern "C" __global__ __launch_bounds__(KKT_T) KKT_OCC void kkt_eliminate(const KktArgs A) {
  __shared__ double panels[KKT_PANEL_DOUBLES + 2 * KKT_GJ_DOUBLES];
  double *cpb = panels, *rpb = panels + 2 * 16 * KKT_R * KKT_LC, *gjb = panels + KKT_PANEL_DOUBLES;
  __shared__ int neg_, bad_;
  // final_block: 0 = level of the chain; 1 = the last remaining block (index 0); 2 = NO chain coupling at all (scenario
  // blocks of a two-stage problem): every block is eliminated in this one launch, against the border only
  const KktIdx ix = kkt_eliminated((long long)blockIdx.x, A.s, A.S, A.T);
  const long long i = A.final_block == 2 ? (long long)blockIdx.x : A.final_block ? kkt_lane_first((long long)blockIdx.x, A.S, A.T) : ix.i;
  if (i >= A.S || (!A.final_block && !ix.valid)) return;
  const bool has_right = !A.final_block && ix.right;
  double *Di = A.D + i * KKT_NB * KKT_NB;
  const int l = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6, li = l & 15, lk = l >> 4;
  // D_i in the accumulator layout (rim tiles padded with zeros: the padding never mixes with the matrix — its operand
  // rows and columns are zero in every update)
  kkt_d4 m[KKT_TRW][KKT_R];
#pragma unroll
  for (int n = 0; n < KKT_TRW; ++n) {
    const int tr = w + n * KKT_W;
#pragma unroll
    for (int tc = 0; tc < KKT_R; ++tc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ii = 16 * tr + lk + 4 * r, jj = 16 * tc + li;
        m[n][tc][r] = (tr < KKT_R && ii < KKT_NB && jj < KKT_NB) ? Di[(long long)ii * KKT_NB + jj] : 0.0;
      }
  }
  if (threadIdx.x == 0) { neg_ = 0; bad_ = 0; }
  // the coupling of i + s to i leaves the chain with i: kept for the solves (the update overwrites Bt[i + s] with the
  // coupling of the next level)
  if (has_right)
    for (int e = (int)threadIdx.x; e < KKT_NC * KKT_NC; e += KKT_T) A.BR[i * KKT_NC * KKT_NC + e] = A.Bt[(i + A.s) * KKT_NC * KKT_NC + e];
  kkt_gj_all<0>(m, cpb, rpb, gjb, A.tiny, neg_, bad_);
  if (threadIdx.x == 0) {
    if (neg_) atomicAdd((unsigned long long *)A.info, (unsigned long long)neg_);
    if (bad_) atomicAdd((unsigned long long *)(A.info + 1), (unsigned long long)bad_);
  }
  kkt_put<KKT_R, false>(Di, m, 1.0, KKT_NB);          // D_i^-1: all the chain ever needs of block i
#if KKT_NE > 0
  {                      // Z_i = D_i^-1 E_i,  Gp_i = E_i' Z_i
    __shared__ double M[KKT_NB * KKT_LD], EE[KKT_NB * KKT_LE], ZZ[KKT_NB * KKT_LE];
    kkt_put_lds<KKT_R>(M, KKT_LD, m, KKT_NB);
    kkt_load(EE, A.E + i * KKT_NB * KKT_NE, KKT_NB, KKT_NE, KKT_LE, false);
    __syncthreads();
    kkt_d4 z[KKT_TRW][KKT_RE];
    kkt_zero(z);
    kkt_mma<KKT_RE>(z, M, KKT_LD, EE, KKT_LE, KKT_NE);
    kkt_put<KKT_RE, false>(A.Z + i * KKT_NB * KKT_NE, z, 1.0, KKT_NE);
    kkt_put_lds<KKT_RE>(ZZ, KKT_LE, z, KKT_NE);
    __syncthreads();
    for (int e = (int)threadIdx.x; e < KKT_NE * KKT_NE; e += KKT_T) {
      const int r = e / KKT_NE, c = e - r * KKT_NE;
      double acc = 0.0;
      for (int k = 0; k < KKT_NB; ++k) acc += EE[k * KKT_LE + r] * ZZ[k * KKT_LE + c];
      A.Gp[i * KKT_NE * KKT_NE + e] = acc;
    }
  }
#endif
}
#endif   // KKT_ROWWISE

// ---- level l, step 2: fold the eliminated neighbours p = j - s, q = j + s into the survivors j = 2 t s ----------------------
// B_j = K[j, p] lives on the rows R of block j and the columns C of block p (KKT_NC x KKT_NC values, Bt[j]); with Dp, Dq the
// inverses kkt_eliminate just wrote:
//        D_j[R, R] -= Bt_j Dp[C, C] Bt_j'        D_j[C, C] -= Bt_q' Dq[R, R] Bt_q        Bt_j <- -Bt_j Dp[C, R] Bt_p
//        E_j[R, :] -= Bt_j Z_p[C, :]             E_j[C, :] -= Bt_q' Z_q[R, :]
// A handful of KKT_NC^3 products on gathered sub-blocks: 64 .. 256 threads, everything through LDS.
#define KKT_TU (KKT_NC <= 8 ? 64 : KKT_NC <= 16 ? 128 : 256)      // a thread per element of an NC x NC product, up to 256
extern "C" __global__ __launch_bounds__(KKT_TU) void kkt_update(const KktArgs A) {
  __shared__ double Bj[KKT_NC * KKT_LN], Bo[KKT_NC * KKT_LN], G1[KKT_NC * KKT_LN], G2[KKT_NC * KKT_LN], T1[KKT_NC * KKT_LN], T2[KKT_NC * KKT_LN],
      BN[KKT_NC * KKT_LN];
  __shared__ int rr[KKT_NC], cc[KKT_NC];
  const KktIdx jx = kkt_survivor((long long)blockIdx.x, A.s, A.S, A.T);
  const long long j = jx.i;
  if (!jx.valid) return;
  const long long p = j - A.s, q = j + A.s;
  const bool hp = jx.left, hq = jx.right;
  if (!hp && !hq) return;
  const int t = (int)threadIdx.x;
  constexpr int NN = KKT_NC * KKT_NC;
  for (int e = t; e < KKT_NC; e += KKT_TU) { rr[e] = A.rows[e]; cc[e] = A.cols[e]; }
  __syncthreads();
  double *Dj = A.D + j * KKT_NB * KKT_NB;
  // the right neighbour's operands are fetched NOW, behind the left half's loads: the two halves are a chain of barriers, and each
  // would otherwise wait for its own HBM round trip (the kernel moves ~14 KB of 128-byte lines per block and ran at half that rate)
  constexpr int NPRE = (KKT_NC * KKT_NC + KKT_TU - 1) / KKT_TU;
  double preB[NPRE], preG[NPRE];
  if (hq) {
    const double *Dq = A.D + q * KKT_NB * KKT_NB;
#pragma unroll
    for (int n = 0; n < NPRE; ++n) {
      const int e = t + n * KKT_TU;
      if (e < KKT_NC * KKT_NC) {
        const int a = e / KKT_NC, b = e - a * KKT_NC;
        preB[n] = A.Bt[q * KKT_NC * KKT_NC + e];
        preG[n] = (rr[a] >= 0 && rr[b] >= 0) ? Dq[rr[a] * KKT_NB + rr[b]] : 0.0;
      } else { preB[n] = 0.0; preG[n] = 0.0; }
    }
  }
  if (hp) {
    const double *Dp = A.D + p * KKT_NB * KKT_NB;
    for (int e = t; e < NN; e += KKT_TU) {
      const int a = e / KKT_NC, b = e - a * KKT_NC;
      Bj[a * KKT_LN + b] = A.Bt[j * NN + e];
      Bo[a * KKT_LN + b] = A.Bt[p * NN + e];
      const bool on = cc[a] >= 0;
      G1[a * KKT_LN + b] = (on && cc[b] >= 0) ? Dp[cc[a] * KKT_NB + cc[b]] : 0.0;
      G2[a * KKT_LN + b] = (on && rr[b] >= 0) ? Dp[cc[a] * KKT_NB + rr[b]] : 0.0;
    }
    __syncthreads();
    for (int e = t; e < NN; e += KKT_TU) {
      const int a = e / KKT_NC, b = e - a * KKT_NC;
      double s1 = 0.0, s2 = 0.0;
      for (int k = 0; k < KKT_NC; ++k) { s1 += Bj[a * KKT_LN + k] * G1[k * KKT_LN + b]; s2 += Bj[a * KKT_LN + k] * G2[k * KKT_LN + b]; }
      T1[a * KKT_LN + b] = s1; T2[a * KKT_LN + b] = s2;
    }
    __syncthreads();
    for (int e = t; e < NN; e += KKT_TU) {
      const int a = e / KKT_NC, b = e - a * KKT_NC;
      double u = 0.0, bn = 0.0;
      for (int k = 0; k < KKT_NC; ++k) { u += T1[a * KKT_LN + k] * Bj[b * KKT_LN + k]; bn += T2[a * KKT_LN + k] * Bo[k * KKT_LN + b]; }
      if (rr[a] >= 0 && rr[b] >= 0) Dj[rr[a] * KKT_NB + rr[b]] -= u;
      BN[a * KKT_LN + b] = -bn;
    }
#if KKT_NE > 0
    for (int e = t; e < KKT_NC * KKT_NE; e += KKT_TU) {
      const int a = e / KKT_NE, c = e - a * KKT_NE;
      if (rr[a] < 0) continue;
      double u = 0.0;
      for (int k = 0; k < KKT_NC; ++k) if (cc[k] >= 0) u += Bj[a * KKT_LN + k] * A.Z[(p * KKT_NB + cc[k]) * KKT_NE + c];
      A.E[(j * KKT_NB + rr[a]) * KKT_NE + c] -= u;
    }
#endif
    __threadfence_block();
    __syncthreads();       // (R and C may share an entry of D_j: the two halves touch it one after the other)
  }
  if (hq) {
#pragma unroll
    for (int n = 0; n < NPRE; ++n) {
      const int e = t + n * KKT_TU;
      if (e < NN) { const int a = e / KKT_NC, b = e - a * KKT_NC; Bo[a * KKT_LN + b] = preB[n]; G1[a * KKT_LN + b] = preG[n]; }
    }
    __syncthreads();
    for (int e = t; e < NN; e += KKT_TU) {
      const int a = e / KKT_NC, b = e - a * KKT_NC;
      double s1 = 0.0;
      for (int k = 0; k < KKT_NC; ++k) s1 += Bo[k * KKT_LN + a] * G1[k * KKT_LN + b];
      T1[a * KKT_LN + b] = s1;
    }
    __syncthreads();
    for (int e = t; e < NN; e += KKT_TU) {
      const int a = e / KKT_NC, b = e - a * KKT_NC;
      double u = 0.0;
      for (int k = 0; k < KKT_NC; ++k) u += T1[a * KKT_LN + k] * Bo[k * KKT_LN + b];
      if (cc[a] >= 0 && cc[b] >= 0) Dj[cc[a] * KKT_NB + cc[b]] -= u;
    }
#if KKT_NE > 0
    for (int e = t; e < KKT_NC * KKT_NE; e += KKT_TU) {
      const int a = e / KKT_NE, c = e - a * KKT_NE;
      if (cc[a] < 0) continue;
      double u = 0.0;
      for (int k = 0; k < KKT_NC; ++k) if (rr[k] >= 0) u += Bo[k * KKT_LN + a] * A.Z[(q * KKT_NB + rr[k]) * KKT_NE + c];
      A.E[(j * KKT_NB + cc[a]) * KKT_NE + c] -= u;
    }
#endif
  }
  if (hp)
    for (int e = t; e < NN; e += KKT_TU) { const int a = e / KKT_NC, b = e - a * KKT_NC; A.Bt[j * NN + e] = BN[a * KKT_LN + b]; }
}

// ---- solves --------------------------------------------------------------------------------------------------------
struct KktSolveArgs {
  const double *D, *Bt, *BR, *Z;   // D holds the inverses; Bt[i] / BR[i]: the couplings of an eliminated block i to i - s / of i + s to i
  const int *rows, *cols;
  double *r;                     // S x NB: right-hand side in, solution out
  double *z;                     // S x NB: D_i^-1 r_i of the eliminated blocks (forward -> backward)
  double *rBp;                   // S x NE: per block  Z_i' r_i  (forward);  unused backward
  const double *xB;              // NE: the border's solution (backward)
  long long S, s;
  int final_block;
  long long T;                   // blocks per lane (0: one chain), as in KktArgs
};
// y[c] = sum_k M[k][c] * v[k]   (M row-major NB x cols: coalesced down the rows)
__device__ __forceinline__ double kkt_tdot(const double *__restrict__ M, const double *v, int cols, int c) {
  double acc = 0.0;
  for (int k = 0; k < KKT_NB; ++k) acc += M[(long long)k * cols + c] * v[k];
  return acc;
}
// forward, level l:   eliminated  z_i = D_i^-1 r_i,  rBp_i = Z_i' r_i ;
//                     survivors   r_j[R] -= BR_p (D_p^-1 r_p)[C] ,  r_j[C] -= Bt_q' (D_q^-1 r_q)[R]      (64 threads per block)
extern "C" __global__ __launch_bounds__(64) void kkt_forward(const KktSolveArgs A) {
  __shared__ double v[KKT_NB], zc[KKT_NC], zr[KKT_NC];
  __shared__ int rr[KKT_NC], cc[KKT_NC];
  const long long T_ = A.T > 0 ? A.T : A.S;
  const long long n_surv = A.final_block ? 0 : (A.S / T_) * ((T_ + 2 * A.s - 1) / (2 * A.s));   // (final_block 1 / 2: only the border terms below)
  const int t = (int)threadIdx.x;
  constexpr int NN = KKT_NC * KKT_NC;
  if ((long long)blockIdx.x < n_surv) {
    const KktIdx jx = kkt_survivor((long long)blockIdx.x, A.s, A.S, A.T);
    const long long j = jx.i, p = j - A.s, q = j + A.s;
    const bool hp = jx.left, hq = jx.right;
    for (int e = t; e < KKT_NC; e += 64) { rr[e] = A.rows[e]; cc[e] = A.cols[e]; zc[e] = 0.0; zr[e] = 0.0; }
    if (hp) {
      for (int e = t; e < KKT_NB; e += 64) v[e] = A.r[p * KKT_NB + e];
      __syncthreads();
      // row cc[a] of the (symmetric) inverse: four lanes per row, 16 rows per pass
      for (int a0 = 0; a0 < KKT_NC; a0 += 16) {
        const int a = a0 + (t >> 2);
        double acc = 0.0;
        if (a < KKT_NC && cc[a] >= 0) { const double *row = A.D + (p * KKT_NB + cc[a]) * KKT_NB; for (int k = t & 3; k < KKT_NB; k += 4) acc += row[k] * v[k]; }
        acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
        if (a < KKT_NC && (t & 3) == 0) zc[a] = acc;
      }
      __syncthreads();
    }
    if (hq) {
      for (int e = t; e < KKT_NB; e += 64) v[e] = A.r[q * KKT_NB + e];
      __syncthreads();
      for (int a0 = 0; a0 < KKT_NC; a0 += 16) {
        const int a = a0 + (t >> 2);
        double acc = 0.0;
        if (a < KKT_NC && rr[a] >= 0) { const double *row = A.D + (q * KKT_NB + rr[a]) * KKT_NB; for (int k = t & 3; k < KKT_NB; k += 4) acc += row[k] * v[k]; }
        acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
        if (a < KKT_NC && (t & 3) == 0) zr[a] = acc;
      }
    }
    __syncthreads();
    if (hp)
      for (int a = t; a < KKT_NC; a += 64) {
        if (rr[a] < 0) continue;
        double acc = 0.0;
        for (int c = 0; c < KKT_NC; ++c) acc += A.BR[p * NN + a * KKT_NC + c] * zc[c];
        A.r[j * KKT_NB + rr[a]] -= acc;
      }
    __threadfence_block();
    __syncthreads();
    if (hq)
      for (int a = t; a < KKT_NC; a += 64) {
        if (cc[a] < 0) continue;
        double acc = 0.0;
        for (int k = 0; k < KKT_NC; ++k) acc += A.Bt[q * NN + k * KKT_NC + a] * zr[k];
        A.r[j * KKT_NB + cc[a]] -= acc;
      }
    return;
  }
  const long long e_idx = (long long)blockIdx.x - n_surv;
  const KktIdx ix = kkt_eliminated(e_idx, A.s, A.S, A.T);
  const long long i = A.final_block == 2 ? e_idx : A.final_block ? kkt_lane_first(e_idx, A.S, A.T) : ix.i;
  if (i >= A.S || (!A.final_block && !ix.valid)) return;
  for (int e = t; e < KKT_NB; e += 64) v[e] = A.r[i * KKT_NB + e];
  __syncthreads();
  if (!A.final_block)
    for (int c = t; c < KKT_NB; c += 64) A.z[i * KKT_NB + c] = kkt_tdot(A.D + i * KKT_NB * KKT_NB, v, KKT_NB, c);
#if KKT_NE > 0
  for (int c = t; c < KKT_NE; c += 64) A.rBp[i * KKT_NE + c] = kkt_tdot(A.Z + i * KKT_NB * KKT_NE, v, KKT_NE, c);
#endif
}
// backward, level l: eliminated  x_i = z_i - D_i^-1[:, R] (Bt_i x_p[C]) - D_i^-1[:, C] (BR_i' x_q[R]) - Z_i x_B
extern "C" __global__ __launch_bounds__(64) void kkt_backward(const KktSolveArgs A) {
  __shared__ double ri[KKT_NB], xc[KKT_NC], xr[KKT_NC], t1[KKT_NC], t2[KKT_NC], xb[KKT_NE > 0 ? KKT_NE : 1];
  __shared__ int rr[KKT_NC], cc[KKT_NC];
  const KktIdx ix = kkt_eliminated((long long)blockIdx.x, A.s, A.S, A.T);
  const long long i = A.final_block == 2 ? (long long)blockIdx.x : A.final_block ? kkt_lane_first((long long)blockIdx.x, A.S, A.T) : ix.i;
  if (i >= A.S || (!A.final_block && !ix.valid)) return;
  const int t = (int)threadIdx.x;
  constexpr int NN = KKT_NC * KKT_NC;
  const bool hp = !A.final_block, hq = !A.final_block && ix.right;
  for (int e = t; e < KKT_NC; e += 64) {      // (no coupling tables without a chain: final_block 1 / 2 never read them)
    const int re = hp ? A.rows[e] : -1, ce = hp ? A.cols[e] : -1;
    rr[e] = re; cc[e] = ce;
    xc[e] = (hp && ce >= 0) ? A.r[(i - A.s) * KKT_NB + ce] : 0.0;
    xr[e] = (hq && re >= 0) ? A.r[(i + A.s) * KKT_NB + re] : 0.0;
  }
  if (A.final_block) for (int e = t; e < KKT_NB; e += 64) ri[e] = A.r[i * KKT_NB + e];
#if KKT_NE > 0
  for (int e = t; e < KKT_NE; e += 64) xb[e] = A.xB[e];
#endif
  __syncthreads();
  for (int a = t; a < KKT_NC; a += 64) {
    double s1 = 0.0, s2 = 0.0;
    if (hp) for (int c = 0; c < KKT_NC; ++c) s1 += A.Bt[i * NN + a * KKT_NC + c] * xc[c];
    if (hq) for (int k = 0; k < KKT_NC; ++k) s2 += A.BR[i * NN + k * KKT_NC + a] * xr[k];
    t1[a] = s1; t2[a] = s2;
  }
  __syncthreads();
  const double *Di = A.D + i * KKT_NB * KKT_NB;
  constexpr int NOUT = (KKT_NB + 63) / 64;
  double out[NOUT];
#pragma unroll
  for (int n = 0; n < NOUT; ++n) {
    const int c = t + 64 * n;
    out[n] = 0.0;
    if (c >= KKT_NB) continue;
    // D^-1 is symmetric: its rows rr[a] / cc[a] are the columns the formula asks for (coalesced over c)
    double acc = A.final_block ? kkt_tdot(Di, ri, KKT_NB, c) : A.z[i * KKT_NB + c];
    if (hp) for (int a = 0; a < KKT_NC; ++a) if (rr[a] >= 0) acc -= Di[rr[a] * KKT_NB + c] * t1[a];
    if (hq) for (int a = 0; a < KKT_NC; ++a) if (cc[a] >= 0) acc -= Di[cc[a] * KKT_NB + c] * t2[a];
#if KKT_NE > 0
    { const double *Zr = A.Z + (i * KKT_NB + c) * KKT_NE; for (int e = 0; e < KKT_NE; ++e) acc -= Zr[e] * xb[e]; }
#endif
    out[n] = acc;
  }
#pragma unroll
  for (int n = 0; n < NOUT; ++n) {
    const int c = t + 64 * n;
    if (c < KKT_NB) A.r[i * KKT_NB + c] = out[n];   // block i's own entries: nobody else reads or writes them at this level
  }
}

#if KKT_NE == 0 && KKT_NB <= 64
// ---- the same solves with a LANE PER ROW (no border, blocks that fit a wave: 64 / NB blocks side by side) --------------------------
// kkt_forward / kkt_backward give a block 64 threads and a handful of barriers whatever its size, and their survivors recompute
// rows of D_p^-1 r_p that the eliminated blocks' part of the same launch computes in full.  Split by what a thread owns:
//   kkt_fz   z_i = D_i^-1 r_i of the level's eliminated blocks — lane c sums column c of the inverse (a row per load, coalesced), r_i through LDS
//   kkt_fs   the survivors: r_j[R] -= BR_p z_p[C],  r_j[C] -= Bt_q' z_q[R]  from the stored z — a thread per survivor, NC x NC each
//   kkt_bw   x_i = z_i - D_i^-1[:, R] (Bt_i x_p[C]) - D_i^-1[:, C] (BR_i' x_q[R]): the two NC-vectors through LDS, then a lane per entry
// Every block inverse is read once per solve on the way up (3.2 KB for 20 x 20) and on 2 NC of its columns on the way down.
#define KKT_SBPW (64 / KKT_NB)
extern "C" __global__ __launch_bounds__(64) void kkt_fz(const KktSolveArgs A) {
  __shared__ double rv[KKT_SBPW * KKT_NB];
  const int lane = (int)threadIdx.x, slot = lane / KKT_NB, li = lane - slot * KKT_NB;
  const long long b = (long long)blockIdx.x * KKT_SBPW + slot;
  const KktIdx ix = kkt_eliminated(b, A.s, A.S, A.T);
  const long long i = A.final_block == 2 ? b : A.final_block ? kkt_lane_first(b, A.S, A.T) : ix.i;
  const bool on = slot < KKT_SBPW && i < A.S && (A.final_block || ix.valid);
  if (slot < KKT_SBPW) rv[slot * KKT_NB + li] = on ? A.r[i * KKT_NB + li] : 0.0;
  __syncthreads();
  if (!on) return;
  // y[c] = sum_k M[k][c] v[k] with c = this lane: every load takes one ROW of the inverse across the block's lanes (coalesced), and the
  // orientation is the one kkt_backward uses (the inverse is symmetric up to rounding only: the two passes must read it the same way)
  const double *col = A.D + i * KKT_NB * KKT_NB + li, *v = rv + slot * KKT_NB;
  double acc = 0.0;
#pragma unroll 8
  for (int kk = 0; kk < KKT_NB; ++kk) acc += col[kk * KKT_NB] * v[kk];
  if (A.final_block) A.r[i * KKT_NB + li] = acc; else A.z[i * KKT_NB + li] = acc;
}
extern "C" __global__ __launch_bounds__(64) void kkt_fs(const KktSolveArgs A) {
  const long long T_ = A.T > 0 ? A.T : A.S, n_surv = (A.S / T_) * ((T_ + 2 * A.s - 1) / (2 * A.s));
  const long long g = (long long)blockIdx.x * 64 + threadIdx.x;
  if (g >= n_surv) return;
  const KktIdx jx = kkt_survivor(g, A.s, A.S, A.T);
  if (!jx.valid) return;
  const long long j = jx.i, p = j - A.s, q = j + A.s;
  constexpr int NN = KKT_NC * KKT_NC;
  double zv[KKT_NC];
  if (jx.left) {
#pragma unroll
    for (int c = 0; c < KKT_NC; ++c) { const int cc = A.cols[c]; zv[c] = cc >= 0 ? A.z[p * KKT_NB + cc] : 0.0; }
    for (int a = 0; a < KKT_NC; ++a) {
      const int ra = A.rows[a];
      if (ra < 0) continue;
      double acc = 0.0;
#pragma unroll
      for (int c = 0; c < KKT_NC; ++c) acc += A.BR[p * NN + a * KKT_NC + c] * zv[c];
      A.r[j * KKT_NB + ra] -= acc;
    }
  }
  if (jx.right) {      // (after the left half: R and C may share an entry of r_j)
#pragma unroll
    for (int c = 0; c < KKT_NC; ++c) { const int rr = A.rows[c]; zv[c] = rr >= 0 ? A.z[q * KKT_NB + rr] : 0.0; }
    for (int a = 0; a < KKT_NC; ++a) {
      const int ca = A.cols[a];
      if (ca < 0) continue;
      double acc = 0.0;
#pragma unroll
      for (int kk = 0; kk < KKT_NC; ++kk) acc += A.Bt[q * NN + kk * KKT_NC + a] * zv[kk];
      A.r[j * KKT_NB + ca] -= acc;
    }
  }
}
extern "C" __global__ __launch_bounds__(64) void kkt_bw(const KktSolveArgs A) {
  __shared__ double t1s[KKT_SBPW * KKT_NC], t2s[KKT_SBPW * KKT_NC];
  const int lane = (int)threadIdx.x, slot = lane / KKT_NB, li = lane - slot * KKT_NB;
  const long long b = (long long)blockIdx.x * KKT_SBPW + slot;
  const KktIdx ix = kkt_eliminated(b, A.s, A.S, A.T);
  const long long i = ix.i;
  const bool on = slot < KKT_SBPW && ix.valid && i < A.S;
  constexpr int NN = KKT_NC * KKT_NC;
  if (on && li < KKT_NC) {
    double s1 = 0.0, s2 = 0.0;
    for (int c = 0; c < KKT_NC; ++c) {
      const int cc = A.cols[c], rr = A.rows[c];
      if (cc >= 0) s1 += A.Bt[i * NN + li * KKT_NC + c] * A.r[(i - A.s) * KKT_NB + cc];
      if (ix.right && rr >= 0) s2 += A.BR[i * NN + c * KKT_NC + li] * A.r[(i + A.s) * KKT_NB + rr];
    }
    t1s[slot * KKT_NC + li] = s1; t2s[slot * KKT_NC + li] = s2;
  }
  __syncthreads();
  if (!on) return;
  const double *col = A.D + i * KKT_NB * KKT_NB + li;      // D^-1 is symmetric: its rows R / C are the columns the formula asks for (coalesced over the lanes)
  double acc = A.z[i * KKT_NB + li];
#pragma unroll
  for (int a = 0; a < KKT_NC; ++a) {
    const int ra = A.rows[a], ca = A.cols[a];
    if (ra >= 0) acc -= col[ra * KKT_NB] * t1s[slot * KKT_NC + a];
    if (ca >= 0) acc -= col[ca * KKT_NB] * t2s[slot * KKT_NC + a];
  }
  A.r[i * KKT_NB + li] = acc;
}
#endif

// ---- the solver-facing entry points (iem_kkt_assemble / _solve): blocks from the COO values, right-hand sides in and out ----
// flat[dest[i]] = sum over k in [seg[i], seg[i + 1]) of the source perm[k] — an index into the virtual array
//   hess values | jac values | (sigma + delta_w) per variable | -delta_c per row | 1.0 (the padding's unit diagonal)
struct KktGatherArgs {
  double *flat;
  const long long *dest;
  const unsigned *seg, *perm;
  const double *hess, *jac, *sigma;
  double dw, dc;
  long long n_dest, n_h, n_j, n_var, n_con;
};
extern "C" __global__ __launch_bounds__(256) void kkt_gather(const KktGatherArgs A) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= A.n_dest) return;
  double acc = 0.0;
  for (unsigned k = A.seg[i]; k < A.seg[i + 1]; ++k) {
    long long s = A.perm[k];
    double v;
    if (s < A.n_h) v = A.hess[s];
    else if ((s -= A.n_h) < A.n_j) v = A.jac[s];
    else if ((s -= A.n_j) < A.n_var) v = (A.sigma ? A.sigma[s] : 0.0) + A.dw;
    else if ((s -= A.n_var) < A.n_con) v = -A.dc;
    else v = 1.0;
    acc += v;
  }
  A.flat[A.dest[i]] = acc;
}
// r[pos[i]] = rhs[idx[i]]  (into the chain / the border)   and back:  sol[idx[i]] = r[pos[i]]
struct KktMoveArgs { double *dst; const double *src; const long long *di, *si; long long n; };
extern "C" __global__ __launch_bounds__(256) void kkt_move(const KktMoveArgs A) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < A.n) A.dst[A.di[i]] = A.src[A.si[i]];
}
// column sums of a rows x w matrix (per-block border terms): workgroup b sums the rows of chunk b / ncc for the 256 columns of
// chunk b % ncc (ncc = ceil(w / 256)) into out[chunk][column]; a second launch over the partials leaves one row
struct KktSumArgs { const double *in; double *out; long long rows, w, rows_per_wg; };
extern "C" __global__ __launch_bounds__(256) void kkt_colsum(const KktSumArgs A) {
  const long long ncc = (A.w + 255) / 256, rc = (long long)blockIdx.x / ncc, cc = (long long)blockIdx.x % ncc;
  const long long c = cc * 256 + threadIdx.x;
  if (c >= A.w) return;
  const long long r0 = rc * A.rows_per_wg, r1 = r0 + A.rows_per_wg < A.rows ? r0 + A.rows_per_wg : A.rows;
  double acc = 0.0;
  for (long long r = r0; r < r1; ++r) acc += A.in[r * A.w + c];
  A.out[rc * A.w + c] = acc;
}

// ---- span-sparse border columns of a laned chain (kkt_chain.HubChainKKT; pandemic 5 000 x 100: u(t) as 5 000 hubs) --------------
// A block alive at level s (time t = a s, a = its index among the blocks alive) carries border columns for the hubs
// t - (s - 1) .. t + (s - 1) only:  E[a][lane] is NQ x W, W = (2 s - 1) hw, on the few local rows Q that ever hold a border entry.
//   kkt_hub_z      Z = D_i^-1[Q, Q] E_i  for the blocks the level has just eliminated (odd a), or for block 0 of every lane (last)
//   kkt_hub_widen  the survivors' columns for level 2 s (W2 = (4 s - 1) hw):  their own in the middle,  - Bt_j Z_p[C] on the rows R
//                  from the left neighbour p = j - s,  - Bt_q' Z_q[R] on the rows C from the right neighbour q = j + s
// (the couplings Bt are those of level s: before kkt_update rewrites them).  One thread per (block, column): pure streaming — the
// same arithmetic took 6.5 ms per factorisation as batched library products over gathered operands.
#define KKT_HUB_MAXQ 24
struct KktHubArgs {
  const double *D, *Bt, *E, *Z;      // Z: input of kkt_hub_widen
  double *out;                       // Z (kkt_hub_z) or the widened columns (kkt_hub_widen)
  const int *q, *qr, *qc;            // local rows Q (nq), positions of the coupling rows R / columns C inside Q (nr / ncq)
  long long T, lanes, s, n_e, n_s;   // blocks per lane, lanes, stride of the level, eliminated / surviving blocks per lane
  int nb, nc, nq, nr, ncq, hw, last; // last: kkt_hub_z on alive block 0 of every lane
};
extern "C" __global__ __launch_bounds__(256) void kkt_hub_z(const KktHubArgs A) {
  const long long W = (2 * A.s - 1) * A.hw, nblk = (A.last ? 1 : A.n_e) * A.lanes;
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= nblk * W) return;
  const long long b = g / W, c = g - b * W, m = b / A.lanes, lane = b - m * A.lanes;
  const long long a = A.last ? 0 : 2 * m + 1;                                    // index among the blocks alive
  const double *Di = A.D + (lane * A.T + a * A.s) * A.nb * A.nb;
  const double *Ei = A.E + ((a * A.lanes + lane) * A.nq) * W + c;
  double e[KKT_HUB_MAXQ];
#pragma unroll
  for (int k = 0; k < KKT_HUB_MAXQ; ++k) e[k] = k < A.nq ? Ei[k * W] : 0.0;
  double *Zo = A.out + (b * A.nq) * W + c;
  for (int r = 0; r < A.nq; ++r) {
    const double *row = Di + (long long)A.q[r] * A.nb;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < KKT_HUB_MAXQ; ++k) if (k < A.nq) acc += row[A.q[k]] * e[k];
    Zo[r * W] = acc;
  }
}
extern "C" __global__ __launch_bounds__(256) void kkt_hub_widen(const KktHubArgs A) {
  const long long W = (2 * A.s - 1) * A.hw, W2 = (4 * A.s - 1) * A.hw;
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= A.n_s * A.lanes * W2) return;
  const long long b = g / W2, c = g - b * W2, m = b / A.lanes, lane = b - m * A.lanes;
  const int NN = A.nc * A.nc;
  double v[KKT_HUB_MAXQ];
  const long long cm = c - A.s * A.hw;                                           // the survivor's own columns sit in the middle
  const double *Es = A.E + ((2 * m * A.lanes + lane) * A.nq) * W + cm;
#pragma unroll
  for (int k = 0; k < KKT_HUB_MAXQ; ++k) v[k] = (k < A.nq && cm >= 0 && cm < W) ? Es[k * W] : 0.0;
  if (m > 0 && c < W) {                                                          // left neighbour: the block eliminated at index m - 1
    const double *Bj = A.Bt + (lane * A.T + 2 * m * A.s) * NN;
    const double *Zp = A.Z + (((m - 1) * A.lanes + lane) * A.nq) * W + c;
    for (int i = 0; i < A.nr; ++i) {
      double acc = 0.0;
      for (int k = 0; k < A.ncq; ++k) acc += Bj[i * A.nc + k] * Zp[A.qc[k] * W];
      const int r = A.qr[i];
#pragma unroll
      for (int k = 0; k < KKT_HUB_MAXQ; ++k) if (k == r) v[k] -= acc;
    }
  }
  const long long cr = c - 2 * A.s * A.hw;
  if (m < A.n_e && cr >= 0 && cr < W) {                                          // right neighbour: the block eliminated at index m
    const double *Bq = A.Bt + (lane * A.T + (2 * m + 1) * A.s) * NN;
    const double *Zq = A.Z + ((m * A.lanes + lane) * A.nq) * W + cr;
    for (int k = 0; k < A.ncq; ++k) {
      double acc = 0.0;
      for (int i = 0; i < A.nr; ++i) acc += Bq[i * A.nc + k] * Zq[A.qr[i] * W];
      const int r = A.qc[k];
#pragma unroll
      for (int j = 0; j < KKT_HUB_MAXQ; ++j) if (j == r) v[j] -= acc;
    }
  }
  double *o = A.out + (b * A.nq) * W2 + c;
#pragma unroll
  for (int k = 0; k < KKT_HUB_MAXQ; ++k) if (k < A.nq) o[k * W2] = v[k];
}

// ---- the hubs' side of iem_kkt_factor / iem_kkt_solve in hub mode (csrc/iem_api.cpp: the GEMMs in between are rocBLAS's) ----------
// N = the part of a panel of the hubs' LDL' factor below its pivot blocks' own squares (those hold D_k, not L), X = I - N
struct KktHubMaskArgs { const double *src; double *N, *X; long long ld; int pw, ldp, leaf; };
extern "C" __global__ __launch_bounds__(256) void kkt_hub_mask(const KktHubMaskArgs A) {
  const int g = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (g >= A.pw * A.pw) return;
  const int r = g / A.pw, c = g - r * A.pw;
  const double v = (r / A.leaf > c / A.leaf) ? A.src[(long long)r * A.ld + c] : 0.0;
  A.N[r * A.ldp + c] = v;
  A.X[r * A.ldp + c] = (r == c ? 1.0 : 0.0) - v;
}
// max |S_ii| of the hubs' Schur complement (one workgroup): the scale of its pivot threshold
struct KktHubDiagArgs { const double *S; double *out; long long n, ld; };
extern "C" __global__ __launch_bounds__(256) void kkt_hub_diagmax(const KktHubDiagArgs A) {
  __shared__ double red[256];
  double v = 0.0;
  for (long long i = threadIdx.x; i < A.n; i += 256) v = fmax(v, fabs(A.S[i * A.ld + i]));
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
  if (threadIdx.x == 0) A.out[0] = red[0];
}
// a pivot block of the hubs' LDL' into its dense 96 x 96 buffer: the w x w square, a unit diagonal beyond it (a short last block)
struct KktHubLeafArgs { const double *src; double *dst; long long ld; int w, nb; };
extern "C" __global__ __launch_bounds__(256) void kkt_hub_leaf(const KktHubLeafArgs A) {
  const int g = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (g >= A.nb * A.nb) return;
  const int r = g / A.nb, c = g - r * A.nb;
  A.dst[g] = (r < A.w && c < A.w) ? A.src[(long long)r * A.ld + c] : (r == c ? 1.0 : 0.0);
}
// out[t hw + k] -= sum over lanes and rows of Q of E0[t][lane][q][k] y[block (lane, t)][Q[q]]      (one wave per time block)
struct KktHubVecArgs { const double *E0, *v; double *out; const int *q; long long T, lanes; int nb, nq, hw; };
extern "C" __global__ __launch_bounds__(64) void kkt_hub_ety(const KktHubVecArgs A) {
  const long long t = blockIdx.x;
  const long long n = A.lanes * A.nq;
  for (int k = 0; k < A.hw; ++k) {
    double acc = 0.0;
    for (long long e = threadIdx.x; e < n; e += 64) {
      const long long lane = e / A.nq; const int qi = (int)(e - lane * A.nq);
      acc += A.E0[((t * A.lanes + lane) * A.nq + qi) * A.hw + k] * A.v[(lane * A.T + t) * A.nb + A.q[qi]];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (threadIdx.x == 0) A.out[t * A.hw + k] -= acc;
  }
}
// r[block (lane, t)][Q[q]] -= sum_k E0[t][lane][q][k] x_B[t hw + k]
extern "C" __global__ __launch_bounds__(256) void kkt_hub_ex(const KktHubVecArgs A) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= A.T * A.lanes * A.nq) return;
  const long long tl = g / A.nq; const int qi = (int)(g - tl * A.nq);
  const long long t = tl / A.lanes, lane = tl - t * A.lanes;
  double acc = 0.0;
  for (int k = 0; k < A.hw; ++k) acc += A.E0[g * A.hw + k] * A.v[t * A.hw + k];
  A.out[(lane * A.T + t) * A.nb + A.q[qi]] -= acc;
}

#endif  // IEM_KKT_DEVICE_H

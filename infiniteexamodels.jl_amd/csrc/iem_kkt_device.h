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
// (kkt_chain.py builds the grouping from the model's slab table and the Jacobian structure, and scatters the KKT values
// into D | B | E | G every iteration.)  Factorisation = block CYCLIC REDUCTION: at level l (stride s = 2^l) the blocks
// i = (2t+1) s are eliminated in parallel — kkt_eliminate inverts D_i in LDS (Gauss-Jordan without pivoting, variables
// first: K is quasi-definite under the interior-point regularisation, so every pivot order is admissible; the signs of
// the pivots are counted — the inertia an interior-point method asks for) and forms X_i = D_i^-1 B_i,
// Y_i = D_i^-1 B_{i+s}', Z_i = D_i^-1 E_i; kkt_update folds them into the surviving neighbours j = 2t s:
//        D_j -= B_j Y_p + B_q' X_q      B_j <- -B_j X_p  (the new coupling j <- j - 2s)      E_j -= B_j Z_p + B_q' Z_q
// (p = j - s, q = j + s).  ceil(log2 S) levels, two launches each, every block a workgroup; the border's Schur
// complement G - sum_i E_i' Z_i is accumulated per block (one partial per block, summed once: deterministic).
// Solve: the same levels forward on the right-hand side (kkt_forward), the border system (tiny, dense), the levels
// backward (kkt_backward).  FP64 throughout; no atomics on floating-point data.
//
// Compile-time: KKT_NB (block size, multiple of 4, <= 96), KKT_NE (border size, 0 or a multiple of 4, <= 64).
#ifndef IEM_KKT_DEVICE_H
#define IEM_KKT_DEVICE_H

#ifndef KKT_NB
#define KKT_NB 40
#endif
#ifndef KKT_NE
#define KKT_NE 0
#endif
#if (KKT_NB / 4) * (KKT_NB / 4) <= 128
#define KKT_T 128                 // threads per workgroup: one 4 x 4 tile per thread up to NB = 44 ...
#else
#define KKT_T 256                 // ... then 256 threads, one to three tiles each
#endif
                                  // (one wave per block — no barriers at all — was measured SLOWER, 73 vs 42 us per block at NB = 40:
                                  //  the sweep is bound by its FP64 instruction stream, and a lane then owns two tiles)
#define KKT_LD (KKT_NB + 1)       // LDS row stride of an NB-wide matrix (odd: no bank conflicts down a column)
#define KKT_LE (KKT_NE + 1)

// ---- register-tiled products ---------------------------------------------------------------------------------------
// Every NB x NB (NB x NE) result is cut into 4 x 4 tiles INTERLEAVED at stride Q = NB / 4 (QE = NE / 4): tile (tr, tc)
// holds rows tr + i Q and columns tc + j Q.  A thread owns the tiles tau = tid + n KKT_T (n < NT; one tile for NB <= 64),
// so that per step of the inner product it reads 4 + 4 values from LDS for 16 fused multiply-adds (the plain
// one-element-per-thread form reads 2 per multiply-add and is bound by LDS bandwidth: 13.9 ms per factorisation at 1e5
// quadrotor supports), neighbouring threads read neighbouring columns (no bank conflict) and share the row (broadcast).
#define KKT_Q (KKT_NB / 4)
#define KKT_NT ((KKT_Q * KKT_Q + KKT_T - 1) / KKT_T)
#define KKT_QE (KKT_NE / 4)
#define KKT_NTE ((KKT_Q * KKT_QE + KKT_T - 1) / KKT_T)

// acc[n][i][j] += sum_k A[(tr + i Q) LD + k] * Bm[k ldb + tc + j QC]      (A: NB x NB in LDS; Bm: NB x (4 QC) in LDS)
template <int NT, int QC>
__device__ __forceinline__ void kkt_mm(double (&acc)[NT][4][4], const double *A, const double *Bm, int ldb) {
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int tau = (int)threadIdx.x + n * KKT_T;
    if (tau >= KKT_Q * QC) continue;
    const int tr = tau / QC, tc = tau - tr * QC;
    const double *a0 = A + tr * KKT_LD, *b0 = Bm + tc;
#pragma unroll 4
    for (int k = 0; k < KKT_NB; ++k) {
      double av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = a0[i * KKT_Q * KKT_LD + k];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = b0[k * ldb + j * QC];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[n][i][j] += av[i] * bv[j];
    }
  }
}
template <int NT>
__device__ __forceinline__ void kkt_zero(double (&acc)[NT][4][4]) {
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[n][i][j] = 0.0;
}
// G (global, row-major, `cols` = 4 QC wide) = sign * acc      /      G -= acc
template <int NT, int QC, bool SUBTRACT>
__device__ __forceinline__ void kkt_put(double *__restrict__ G, const double (&acc)[NT][4][4], double sign) {
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int tau = (int)threadIdx.x + n * KKT_T;
    if (tau >= KKT_Q * QC) continue;
    const int tr = tau / QC, tc = tau - tr * QC;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double *g = G + (long long)(tr + i * KKT_Q) * (4 * QC) + tc + j * QC;
        if (SUBTRACT) *g -= acc[n][i][j]; else *g = sign * acc[n][i][j];
      }
  }
}
// the same tiles into LDS (stride ld)
template <int NT, int QC>
__device__ __forceinline__ void kkt_put_lds(double *L, int ld, const double (&acc)[NT][4][4]) {
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int tau = (int)threadIdx.x + n * KKT_T;
    if (tau >= KKT_Q * QC) continue;
    const int tr = tau / QC, tc = tau - tr * QC;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) L[(tr + i * KKT_Q) * ld + tc + j * QC] = acc[n][i][j];
  }
}
// global (row-major, COLS wide) -> LDS (stride ld), optionally transposed.  All of a thread's loads are ISSUED before the
// first one is consumed (a load -> store loop waits a full HBM round trip per element: 12 round trips per NB = 40 tile).
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
// the two halves separately, so that a tile needed LATER is already on its way while the current product runs
#define KKT_NREG ((KKT_NB * KKT_NB + KKT_T - 1) / KKT_T)
__device__ __forceinline__ void kkt_fetch(double (&tmp)[KKT_NREG], const double *__restrict__ g, bool on) {
#pragma unroll
  for (int n = 0; n < KKT_NREG; ++n) {
    const int e = (int)threadIdx.x + n * KKT_T;
    tmp[n] = (on && e < KKT_NB * KKT_NB) ? g[e] : 0.0;
  }
}
__device__ __forceinline__ void kkt_stash(double *lds, const double (&tmp)[KKT_NREG], bool transpose) {
#pragma unroll
  for (int n = 0; n < KKT_NREG; ++n) {
    const int e = (int)threadIdx.x + n * KKT_T;
    if (e < KKT_NB * KKT_NB) {
      const int r = e / KKT_NB, c = e - r * KKT_NB;
      if (transpose) lds[c * KKT_LD + r] = tmp[n]; else lds[r * KKT_LD + c] = tmp[n];
    }
  }
}
// 1 / x: hardware reciprocal + two Newton steps (full double precision for the normal range the pivots live in; the
// IEEE division sequence is three times as many FP64 instructions on the sweep's critical path)
__device__ __forceinline__ double kkt_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = r + r * (1.0 - x * r);
  r = r + r * (1.0 - x * r);
  return r;
}

struct KktArgs {
  double *D, *B, *X, *Y, *E, *Z, *Gp;
  long long *info;          // [0] negative pivots, [1] pivots below the threshold (replaced by +-tiny: the factorisation is not to be trusted)
  long long S, s;           // blocks, stride of this level
  int final_block;          // 1: the last remaining block (index 0), no chain neighbours
  double tiny;              // pivot threshold
};

// Gauss-Jordan steps k = KA Q .. KA Q + Q - 1 of the in-place inverse of the matrix held tile-wise in registers: pivot row /
// column k = kq + KA Q sit in slot KA of the tiles with tr == kq / tc == kq — KA is a template parameter, so every register
// index below is a compile-time constant (with a run-time slot the compiler emits a chain of 64-bit conditional moves per
// element and the sweep is three times slower).  Per step only the pivot row and column travel through LDS.
template <int KA>
__device__ __forceinline__ void kkt_gj_sweep(double (&m)[KKT_NT][4][4], double *rk, double *ck, double tiny, int &neg_, int &bad_) {
  for (int kq = 0; kq < KKT_Q; ++kq) {
    const int k = kq + KA * KKT_Q;
#pragma unroll
    for (int n = 0; n < KKT_NT; ++n) {
      const int tau = (int)threadIdx.x + n * KKT_T;
      if (tau >= KKT_Q * KKT_Q) continue;
      const int tr = tau / KKT_Q, tc = tau - tr * KKT_Q;
      if (tr == kq) {
#pragma unroll
        for (int b = 0; b < 4; ++b) rk[tc + b * KKT_Q] = m[n][KA][b];
      }
      if (tc == kq) {
#pragma unroll
        for (int a = 0; a < 4; ++a) ck[tr + a * KKT_Q] = m[n][a][KA];
      }
    }
    __syncthreads();
    double piv = rk[k];
    if (threadIdx.x == 0) {
      if (piv < 0.0) ++neg_;
      if (!(fabs(piv) >= tiny)) ++bad_;
    }
    if (!(fabs(piv) >= tiny)) piv = piv < 0.0 ? -tiny : tiny;   // keep going with a bounded pivot; info[1] reports it
    const double inv = kkt_rcp(piv);
#pragma unroll
    for (int n = 0; n < KKT_NT; ++n) {
      const int tau = (int)threadIdx.x + n * KKT_T;
      if (tau >= KKT_Q * KKT_Q) continue;
      const int tr = tau / KKT_Q, tc = tau - tr * KKT_Q;
      double rc[4], ci[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) ci[a] = ck[tr + a * KKT_Q] * inv;
#pragma unroll
      for (int b = 0; b < 4; ++b) rc[b] = rk[tc + b * KKT_Q];
      // every element: the rank-1 step m -= c_r (1/piv) r_c ...
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) m[n][a][b] -= ci[a] * rc[b];
      // ... then the few threads that own a piece of the pivot row / column put the inverse's entries there
      if (tr == kq) {
#pragma unroll
        for (int b = 0; b < 4; ++b) m[n][KA][b] = rc[b] * inv;
      }
      if (tc == kq) {
#pragma unroll
        for (int a = 0; a < 4; ++a) m[n][a][KA] = -ci[a];
        if (tr == kq) m[n][KA][KA] = inv;
      }
    }
    __syncthreads();
  }
}

// ---- level l, step 1: eliminate the blocks i = (2 t + 1) s --------------------------------------------------
extern "C" __global__ __launch_bounds__(KKT_T) void kkt_eliminate(const KktArgs A) {
  __shared__ double M[KKT_NB * KKT_LD], T[KKT_NB * KKT_LD], rk[KKT_NB], ck[KKT_NB];
  __shared__ int neg_, bad_;
  // final_block: 0 = level of the chain; 1 = the last remaining block (index 0); 2 = NO chain coupling at all (scenario
  // blocks of a two-stage problem): every block is eliminated in this one launch, against the border only
  const long long i = A.final_block == 2 ? (long long)blockIdx.x : A.final_block ? 0 : (2 * (long long)blockIdx.x + 1) * A.s;
  if (i >= A.S) return;
  const bool has_left = !A.final_block, has_right = !A.final_block && i + A.s < A.S;
  double *Di = A.D + i * KKT_NB * KKT_NB;
  // D_i in REGISTERS, tile-wise; in-place inverse by Gauss-Jordan with the pivots on the diagonal in the given order
  // (variables first).  Per step only the pivot row and column travel through LDS (2 NB values), every thread updates
  // the 16 elements it owns.
  double m[KKT_NT][4][4];
#pragma unroll
  for (int n = 0; n < KKT_NT; ++n) {
    const int tau = (int)threadIdx.x + n * KKT_T;
    const int tr = tau / KKT_Q, tc = tau - tr * KKT_Q;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) m[n][a][b] = tau < KKT_Q * KKT_Q ? Di[(long long)(tr + a * KKT_Q) * KKT_NB + tc + b * KKT_Q] : 0.0;
  }
  if (threadIdx.x == 0) { neg_ = 0; bad_ = 0; }
  double pl[KKT_NREG], pr[KKT_NREG];       // B_i and B_{i+s}: on their way while the sweep runs
  kkt_fetch(pl, A.B + i * KKT_NB * KKT_NB, has_left);
  kkt_fetch(pr, A.B + (i + A.s) * KKT_NB * KKT_NB, has_right);
  kkt_gj_sweep<0>(m, rk, ck, A.tiny, neg_, bad_);
  kkt_gj_sweep<1>(m, rk, ck, A.tiny, neg_, bad_);
  kkt_gj_sweep<2>(m, rk, ck, A.tiny, neg_, bad_);
  kkt_gj_sweep<3>(m, rk, ck, A.tiny, neg_, bad_);
  if (threadIdx.x == 0) {
    if (neg_) atomicAdd((unsigned long long *)A.info, (unsigned long long)neg_);
    if (bad_) atomicAdd((unsigned long long *)(A.info + 1), (unsigned long long)bad_);
  }
  kkt_put<KKT_NT, KKT_Q, false>(Di, m, 1.0);          // D_i^-1 stays for the solves
  kkt_put_lds<KKT_NT, KKT_Q>(M, KKT_LD, m);           // ... and is the left operand of the products below
  if (has_left) {        // X_i = D_i^-1 B_i
    kkt_stash(T, pl, false);
    __syncthreads();
    kkt_zero(m);
    kkt_mm<KKT_NT, KKT_Q>(m, M, T, KKT_LD);
    kkt_put<KKT_NT, KKT_Q, false>(A.X + i * KKT_NB * KKT_NB, m, 1.0);
    __syncthreads();
  }
  if (has_right) {       // Y_i = D_i^-1 B_{i+s}'
    kkt_stash(T, pr, true);
    __syncthreads();
    kkt_zero(m);
    kkt_mm<KKT_NT, KKT_Q>(m, M, T, KKT_LD);
    kkt_put<KKT_NT, KKT_Q, false>(A.Y + i * KKT_NB * KKT_NB, m, 1.0);
    __syncthreads();
  }
#if KKT_NE > 0
  {                      // Z_i = D_i^-1 E_i,  Gp_i = E_i' Z_i
    __shared__ double EE[KKT_NB * KKT_LE], ZZ[KKT_NB * KKT_LE];
    if (!has_left && !has_right) __syncthreads();     // M complete before it is read
    kkt_load(EE, A.E + i * KKT_NB * KKT_NE, KKT_NB, KKT_NE, KKT_LE, false);
    __syncthreads();
    double z[KKT_NTE][4][4];
    kkt_zero(z);
    kkt_mm<KKT_NTE, KKT_QE>(z, M, EE, KKT_LE);
    kkt_put<KKT_NTE, KKT_QE, false>(A.Z + i * KKT_NB * KKT_NE, z, 1.0);
    kkt_put_lds<KKT_NTE, KKT_QE>(ZZ, KKT_LE, z);
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

// ---- level l, step 2: fold the eliminated neighbours into the survivors j = 2 t s ---------------------------------
extern "C" __global__ __launch_bounds__(KKT_T) void kkt_update(const KktArgs A) {
  __shared__ double T1[KKT_NB * KKT_LD], T2[KKT_NB * KKT_LD];
  const long long j = 2 * (long long)blockIdx.x * A.s;
  if (j >= A.S) return;
  const long long p = j - A.s, q = j + A.s;
  const bool hp = j > 0, hq = q < A.S;
  if (!hp && !hq) return;
  double u[KKT_NT][4][4];
  kkt_zero(u);
#if KKT_NE > 0
  __shared__ double ZZ[KKT_NB * KKT_LE];
  double ue[KKT_NTE][4][4];
  kkt_zero(ue);
#endif
  double ra[KKT_NREG], rb[KKT_NREG], rc[KKT_NREG];
  kkt_fetch(ra, A.B + j * KKT_NB * KKT_NB, hp);        // B_j = K[j, p]
  kkt_fetch(rb, A.Y + p * KKT_NB * KKT_NB, hp);        // Y_p = D_p^-1 B_j'
  kkt_fetch(rc, A.X + p * KKT_NB * KKT_NB, hp);        // X_p = D_p^-1 B_p
  if (hp) {
    kkt_stash(T1, ra, false);
    kkt_stash(T2, rb, false);
#if KKT_NE > 0
    kkt_load(ZZ, A.Z + p * KKT_NB * KKT_NE, KKT_NB, KKT_NE, KKT_LE, false);
#endif
    __syncthreads();
  }
  kkt_fetch(ra, A.B + q * KKT_NB * KKT_NB, hq);        // B_q' = K[j, q]   — in flight during the products below
  kkt_fetch(rb, A.X + q * KKT_NB * KKT_NB, hq);        // X_q = D_q^-1 B_q
  if (hp) {
    kkt_mm<KKT_NT, KKT_Q>(u, T1, T2, KKT_LD);
#if KKT_NE > 0
    kkt_mm<KKT_NTE, KKT_QE>(ue, T1, ZZ, KKT_LE);
#endif
    __syncthreads();
    kkt_stash(T2, rc, false);
    __syncthreads();
    {   // the new coupling of j to j - 2s (p's other neighbour; p - s >= 0 always)
      double w[KKT_NT][4][4];
      kkt_zero(w);
      kkt_mm<KKT_NT, KKT_Q>(w, T1, T2, KKT_LD);
      kkt_put<KKT_NT, KKT_Q, false>(A.B + j * KKT_NB * KKT_NB, w, -1.0);
    }
    __syncthreads();
  }
  if (hq) {
    kkt_stash(T1, ra, true);
    kkt_stash(T2, rb, false);
#if KKT_NE > 0
    kkt_load(ZZ, A.Z + q * KKT_NB * KKT_NE, KKT_NB, KKT_NE, KKT_LE, false);
#endif
    __syncthreads();
    kkt_mm<KKT_NT, KKT_Q>(u, T1, T2, KKT_LD);
#if KKT_NE > 0
    kkt_mm<KKT_NTE, KKT_QE>(ue, T1, ZZ, KKT_LE);
#endif
  }
  kkt_put<KKT_NT, KKT_Q, true>(A.D + j * KKT_NB * KKT_NB, u, 1.0);
#if KKT_NE > 0
  kkt_put<KKT_NTE, KKT_QE, true>(A.E + j * KKT_NB * KKT_NE, ue, 1.0);
#endif
}

// ---- solves --------------------------------------------------------------------------------------------------------
struct KktSolveArgs {
  const double *D, *X, *Y, *Z;   // D holds the inverses
  double *r;                     // S x NB: right-hand side in, solution out
  double *rBp;                   // S x NE: per block  Z_i' r_i  (forward);  unused backward
  const double *xB;              // NE: the border's solution (backward)
  long long S, s;
  int final_block;
};
// y[c] (+)= sum_k M[k][c] * v[k]   (M row-major NB x cols: coalesced down the rows)
__device__ __forceinline__ double kkt_tdot(const double *__restrict__ M, const double *v, int cols, int c) {
  double acc = 0.0;
  for (int k = 0; k < KKT_NB; ++k) acc += M[(long long)k * cols + c] * v[k];
  return acc;
}
// forward, level l: survivors  r_j -= Y_p' r_p + X_q' r_q ;  eliminated  rBp_i = Z_i' r_i   (64 threads per block of unknowns)
extern "C" __global__ __launch_bounds__(64) void kkt_forward(const KktSolveArgs A) {
  __shared__ double v[KKT_NB];
  const long long n_surv = A.final_block ? 0 : (A.S + 2 * A.s - 1) / (2 * A.s);   // (final_block 1 / 2: only the border terms below)
  const int t = (int)threadIdx.x;
  if ((long long)blockIdx.x < n_surv) {
    const long long j = 2 * (long long)blockIdx.x * A.s, p = j - A.s, q = j + A.s;
    double acc = 0.0;
    if (j > 0) {
      for (int e = t; e < KKT_NB; e += 64) v[e] = A.r[p * KKT_NB + e];
      __syncthreads();
      for (int c = t; c < KKT_NB; c += 64) acc += kkt_tdot(A.Y + p * KKT_NB * KKT_NB, v, KKT_NB, c);
      __syncthreads();
    }
    if (q < A.S) {
      for (int e = t; e < KKT_NB; e += 64) v[e] = A.r[q * KKT_NB + e];
      __syncthreads();
      for (int c = t; c < KKT_NB; c += 64) acc += kkt_tdot(A.X + q * KKT_NB * KKT_NB, v, KKT_NB, c);
    }
    // (NB <= 64 is NOT assumed: each thread owns the columns c = t, t + 64, ... — but acc sums them; handle > 64 separately)
#if KKT_NB <= 64
    if (t < KKT_NB) A.r[j * KKT_NB + t] -= acc;
#else
    // one column per pass
    (void)acc;
    for (int c = t; c < KKT_NB; c += 64) {
      double a2 = 0.0;
      if (j > 0) { for (int k = 0; k < KKT_NB; ++k) a2 += A.Y[(p * KKT_NB + k) * KKT_NB + c] * A.r[p * KKT_NB + k]; }
      if (q < A.S) { for (int k = 0; k < KKT_NB; ++k) a2 += A.X[(q * KKT_NB + k) * KKT_NB + c] * A.r[q * KKT_NB + k]; }
      A.r[j * KKT_NB + c] -= a2;
    }
#endif
    return;
  }
#if KKT_NE > 0
  const long long e_idx = (long long)blockIdx.x - n_surv;
  const long long i = A.final_block == 2 ? e_idx : A.final_block ? 0 : (2 * e_idx + 1) * A.s;
  if (i >= A.S) return;
  for (int e = t; e < KKT_NB; e += 64) v[e] = A.r[i * KKT_NB + e];
  __syncthreads();
  for (int c = t; c < KKT_NE; c += 64) A.rBp[i * KKT_NE + c] = kkt_tdot(A.Z + i * KKT_NB * KKT_NE, v, KKT_NE, c);
#endif
}
// backward, level l: eliminated  x_i = D_i^-1 r_i - X_i x_p - Y_i x_q - Z_i x_B
extern "C" __global__ __launch_bounds__(64) void kkt_backward(const KktSolveArgs A) {
  __shared__ double ri[KKT_NB], xp[KKT_NB], xq[KKT_NB], xb[KKT_NE > 0 ? KKT_NE : 1];
  const long long i = A.final_block == 2 ? (long long)blockIdx.x : A.final_block ? 0 : (2 * (long long)blockIdx.x + 1) * A.s;
  if (i >= A.S) return;
  const int t = (int)threadIdx.x;
  const bool hp = !A.final_block, hq = !A.final_block && i + A.s < A.S;
  for (int e = t; e < KKT_NB; e += 64) {
    ri[e] = A.r[i * KKT_NB + e];
    xp[e] = hp ? A.r[(i - A.s) * KKT_NB + e] : 0.0;
    xq[e] = hq ? A.r[(i + A.s) * KKT_NB + e] : 0.0;
  }
#if KKT_NE > 0
  for (int e = t; e < KKT_NE; e += 64) xb[e] = A.xB[e];
#endif
  __syncthreads();
  constexpr int NOUT = (KKT_NB + 63) / 64;
  double out[NOUT];
#pragma unroll
  for (int n = 0; n < NOUT; ++n) {
    const int r = t + 64 * n;
    out[n] = 0.0;
    if (r >= KKT_NB) continue;
    // D^-1 is symmetric: walk its column r (coalesced); X, Y, Z rows are read strided (NB x NB is small, L2-resident)
    double acc = kkt_tdot(A.D + i * KKT_NB * KKT_NB, ri, KKT_NB, r);
    if (hp) { const double *Xr = A.X + (i * KKT_NB + r) * KKT_NB; for (int c = 0; c < KKT_NB; ++c) acc -= Xr[c] * xp[c]; }
    if (hq) { const double *Yr = A.Y + (i * KKT_NB + r) * KKT_NB; for (int c = 0; c < KKT_NB; ++c) acc -= Yr[c] * xq[c]; }
#if KKT_NE > 0
    { const double *Zr = A.Z + (i * KKT_NB + r) * KKT_NE; for (int c = 0; c < KKT_NE; ++c) acc -= Zr[c] * xb[c]; }
#endif
    out[n] = acc;
  }
#pragma unroll
  for (int n = 0; n < NOUT; ++n) {
    const int r = t + 64 * n;
    if (r < KKT_NB) A.r[i * KKT_NB + r] = out[n];   // block i's own entries: nobody else reads or writes them at this level
  }
}

#endif  // IEM_KKT_DEVICE_H

// iem_device.h — hand-written gfx950 device primitives of the evaluation kernels.
//
// The per-model arithmetic (forward values, adjoints, second-order sweeps of every
// template sharing a support grid, fused into one straight-line lane program) is
// emitted by iem_codegen.cpp; everything that touches memory in a non-trivial way
// lives here:
//
//   * one wavefront LANE per discretisation support: lane q of a 64-wide wave owns
//     grid point q, so x-slab reads  x[off_k + q]  are 512-byte coalesced loads;
//   * COO blocks are item-major (position o + nslots*k + s — the layout ExaModels'
//     jac_coord!/hess_coord! define), so a workgroup's NS×TILE values are CONTIGUOUS in HBM.
//     Default store path (store_mode 2): iem_stage<NS> parks the values of several templates
//     in LDS, one barrier pair later iem_flush<NS> writes each block with 128-byte-ALIGNED,
//     fully coalesced, non-temporal stores (COO offsets such as 9 would otherwise misalign
//     every line: 4.5 vs 5.6 TB/s, profiles/r01_store_pattern_microbench.txt).  store_mode 1
//     (iem_store_rows<NS>, wave-private LDS transpose, no barrier) and store_mode 0 (direct
//     strided stores) remain for A/B runs.  No atomics, no zero-fill pass;
//   * objective: a bounded number of workgroups walks the tiles; wave shuffle reduction → LDS → one
//     partial per workgroup; the last workgroup to finish sums the partials in a fixed order
//     (bitwise reproducible, no second launch);
//   * gradient / J'v / Hv entries shared by many items (finite / first-stage variables): the same
//     scheme per entry (iem_shared_park / iem_shared_totals) — deterministic, no atomics; sums over a
//     non-lane axis are parked row by row and reduced by iem_axis_sum_kernel, what would still need more
//     than two atomics per entry is parked per item and summed by iem_gather_sum_kernel in the order of a
//     host-built plan; f64 atomics remain only where at most two addends can meet (they commute).
//     Entries nothing writes are zeroed by the kernel itself (iem_zero_fill) when no slot accumulates.
//
// wave = 64 lanes on CDNA4; the generated kernels run IEM_TILE (default 512) lanes per workgroup.
#ifndef IEM_DEVICE_H
#define IEM_DEVICE_H

#define IEM_BLOCK 256   // helper kernels (reduction, structure)
#ifndef IEM_TILE
#define IEM_TILE 256    // workgroup size of the generated fused kernels (set by the generator)
#endif
#define IEM_WAVE 64

__device__ __forceinline__ int iem_lane() { return (int)(threadIdx.x & (IEM_WAVE - 1)); }
__device__ __forceinline__ int iem_wave() { return (int)(threadIdx.x >> 6); }

// Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one; observed, not promised).
// Optional map (generator knob xcd_remap, default off) from hardware workgroup b to a logical
// workgroup L such that CONSECUTIVE logical workgroups run on the same XCD.  Tried so that the two
// partial writes of the 128-byte line at a block seam would merge in one L2: they do not
// (profiles/r01_ab_overlap_xcd.txt: -1.5 %); the overlapped tiles of iem_flush remove the partial
// lines instead.  Bijective for any nb.
__device__ __forceinline__ long long iem_xcd_remap(long long b, long long nb) {
  const long long p = nb >> 3, r = nb & 7, x = b & 7, i = b >> 3;
  return x * p + (x < r ? x : r) + i;
}

// order LDS traffic of one wave: LDS executes a wave's instructions in issue order, so a
// compiler-level fence plus lgkmcnt(0) is enough; no workgroup barrier.
__device__ __forceinline__ void iem_wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// streaming store of an output element: the COO buffers are written once per call and
// never re-read by the evaluator, so IEM_NT=1 marks them non-temporal.
#ifndef IEM_NT
#define IEM_NT 0
#endif
#ifndef IEM_WIDE_STORES
#define IEM_WIDE_STORES 0
#endif
__device__ __forceinline__ void iem_stg(double *p, double v) {
#if IEM_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

typedef double iem_dbl2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void iem_stg2(double *p, double a, double b) {   // p 16-byte aligned
  iem_dbl2 v = {a, b};
#if IEM_NT
  __builtin_nontemporal_store(v, reinterpret_cast<iem_dbl2 *>(p));
#else
  *reinterpret_cast<iem_dbl2 *>(p) = v;
#endif
}

// Direct form: lane writes its NS slots at stride NS (kept for A/B measurement).
template <int NS>
__device__ __forceinline__ void iem_store_rows_direct(double *__restrict__ out, long long pos0, bool valid,
                                                      const double (&v)[NS]) {
  if (valid) {
#pragma unroll
    for (int s = 0; s < NS; ++s) iem_stg(out + pos0 + s, v[s]);
  }
}

// LDS-transposed form.  Precondition (guaranteed by the generator): within a wave the
// valid lanes hold consecutive items, i.e. pos0(lane+1) == pos0(lane) + NS.
template <int NS>
__device__ __forceinline__ void iem_store_rows(double *__restrict__ out, long long pos0, bool valid,
                                               const double (&v)[NS], double *__restrict__ lds_wave) {
  const int lane = iem_lane();
  const unsigned long long m = __ballot(valid);
  if (m == 0ull) return;  // wave-uniform
#if defined(IEM_ABLATE) && IEM_ABLATE == 1
  {  // timing experiment only (WRONG layout): coalesced stores without the LDS transpose
    const long long b0 = __shfl(pos0, __ffsll((long long)m) - 1, IEM_WAVE);
#pragma unroll
    for (int s = 0; s < NS; ++s) if (valid) iem_stg(out + b0 + s * IEM_WAVE + lane, v[s]);
    return;
  }
#endif
#pragma unroll
  for (int s = 0; s < NS; ++s) lds_wave[lane * NS + s] = v[s];
  const int first = __ffsll((long long)m) - 1;
  // position of (virtual) lane 0, slot 0 — wave-uniform
  const long long base = __shfl(pos0, first, IEM_WAVE) - (long long)first * NS;
  iem_wave_lds_sync();
  double *__restrict__ dst = out + base + lane;
  if (m == ~0ull) {  // full wave (the common case): NS unpredicated 512-byte stores
#pragma unroll
    for (int j = 0; j < NS; ++j) iem_stg(dst + j * IEM_WAVE, lds_wave[j * IEM_WAVE + lane]);
  } else {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int e = j * IEM_WAVE + lane;  // element of the wave's contiguous block
      const int src = e / NS;             // lane that produced it (NS is a compile-time constant)
      if ((m >> src) & 1ull) iem_stg(dst + j * IEM_WAVE, lds_wave[e]);
    }
  }
  iem_wave_lds_sync();
}

// IEM-TILE-REGION-BEGIN — everything from here to the matching END marker depends on IEM_TILE.  A program whose kernels
// use more than one workgroup size (jac_coord! / hess_coord! of a LARGE grid run 1 024-lane tiles, everything else 512) gets
// this region once per size, each copy in a namespace iem_t<size> of its own together with the kernels of that size
// (csrc/iem_api.cpp: full_source).
// Block-cooperative, 128-byte-ALIGNED form (store_mode 2).
//
// The COO offsets are ExaModels' running counters (o1 = 9, 209, ...), so a wave's own
// block  [o + NS*64*w, o + NS*64*(w+1))  starts at an arbitrary 8-byte phase: every 512-byte
// wave store would straddle an extra cache line (measured: 4.5 vs 5.6 TB/s for the same
// bytes, tools/membw2.hip).  Here the 4 waves of a workgroup stage their 256 items and
// re-cut the workgroup's contiguous block at 128-byte boundaries of the OUTPUT address:
//   head  (< 16 elements up to the first boundary)   one partial store
//   body  every wave store is 512 bytes, 128-byte aligned
// v0/v1 = first / one-past-last valid lane of the workgroup (template guards are ranges
// in q0, so validity is an interval); P0 = position of lane 0's slot 0 (block-uniform).
__device__ __forceinline__ int iem_clamp256(long long v) { return v < 0 ? 0 : (v > IEM_TILE ? IEM_TILE : (int)v); }

// The two halves of the block store, so that several templates can share one barrier pair:
//   iem_stage<NS>  lane writes its NS values item-major into its template's LDS region
//   __syncthreads()
//   iem_flush<NS>  the workgroup writes the region out, re-cut at 128-byte boundaries
//   __syncthreads()   (before the regions are reused)
template <int NS>
__device__ __forceinline__ void iem_stage(const double (&v)[NS], double *__restrict__ lds_reg) {
  const int t = (int)threadIdx.x;
#pragma unroll
  for (int s = 0; s < NS; ++s) lds_reg[t * NS + s] = v[s];
}

// P0 = position of lane 0 / slot 0 (virtual when the template's row starts inside the tile);
// [v0, v1) = lanes of the tile that hold items of the template; `first` = the row's first item
// is at or behind lane 0.  STRIDE = grid points per workgroup: with STRIDE = IEM_TILE - 16 the
// tiles overlap by 16 lanes (the halo items are computed by both neighbours) and a workgroup
// writes exactly the 128-byte lines that START inside its own STRIDE lanes — whole lines only,
// except at the two ends of the row.  STRIDE = IEM_TILE: disjoint tiles, a partial line at
// every seam (two workgroups write the two parts; costs ~10 % of the store rate).
// The range logic runs in 32-bit offsets relative to `lo` (the block's first element).  Measured
// (profiles/r02_ab_flush_flat.txt, in-process A/B): deciding whole rounds with scalar compares and a
// branch (IEM_FLUSH32 == 1) is 1-5 % SLOWER than a per-lane predicate on every round — the kernels
// wait on the store queue, not on instruction issue — so the default (2) predicates every round.
//   ph   = lo & 15: phase of the block's first element inside its 128-byte line
//   n_u  = elements of the lanes only this workgroup computes, n_all = of all its lanes (halo included)
//   prev = elements the PREVIOUS workgroup's lanes reach past `lo` (its halo), -1: no previous workgroup
template <int NS, int STRIDE>
__device__ __forceinline__ void iem_flush_rel(double *__restrict__ out_lo, const double *__restrict__ lds_lo, int ph, int n_u,
                                              int n_all, int prev) {
  const int t = (int)threadIdx.x;
  int own_lo = 0, own_hi = n_all;
  if (STRIDE < IEM_TILE) {
    if (prev >= 0) {                       // not the first workgroup of the row: start at the next line boundary ...
      own_lo = (16 - ph) & 15;
      if (own_lo > prev) own_lo = prev;    // ... but never leave a gap behind what the previous workgroup could write
    }
    own_hi = ((ph + n_u + 15) & ~15) - ph;
    if (own_hi > n_all) own_hi = n_all;
  }
  const int e_al = ((ph + own_lo) & ~15) - ph;   // in (-16, 16): 128-byte aligned in the output
#if IEM_WIDE_STORES
#pragma unroll
  for (int j = 0; j < (NS + 3) / 2; ++j) {
    const int e = e_al + 2 * t + j * (2 * IEM_TILE);
    const bool a = e >= own_lo && e < own_hi, b = e + 1 >= own_lo && e + 1 < own_hi;
    if (a && b) iem_stg2(out_lo + e, lds_lo[e], lds_lo[e + 1]);
    else if (a) iem_stg(out_lo + e, lds_lo[e]);
    else if (b) iem_stg(out_lo + e + 1, lds_lo[e + 1]);
  }
#else
#if IEM_FLUSH32 == 2
#pragma unroll
  for (int j = 0; j <= NS; ++j) {          // 32-bit per-lane predicate on every round, no branch
    const int e = e_al + t + j * IEM_TILE;
    if (e >= own_lo && e < own_hi) iem_stg(out_lo + e, lds_lo[e]);
  }
#else
#pragma unroll
  for (int j = 0; j <= NS; ++j) {
    const int eb = e_al + j * IEM_TILE;    // block-uniform
    if (eb >= own_lo && eb + IEM_TILE <= own_hi) {
      iem_stg(out_lo + eb + t, lds_lo[eb + t]);
    } else if (eb < own_hi && eb + IEM_TILE > own_lo) {
      const int e = eb + t;
      if (e >= own_lo && e < own_hi) iem_stg(out_lo + e, lds_lo[e]);
    }
  }
#endif
#endif
}

#ifndef IEM_FLUSH32
#define IEM_FLUSH32 2
#endif
#if !IEM_FLUSH32
// round-1 form (knob flush32 = 0, A/B runs only): 64-bit range checks as per-lane predicates on every round
template <int NS, int STRIDE>
__device__ __forceinline__ void iem_flush(double *__restrict__ out, long long P0, int v0, int v1, bool first,
                                          const double *__restrict__ lds_reg) {
  const int t = (int)threadIdx.x;
  const long long lo = P0 + (long long)v0 * NS, hi_all = P0 + (long long)v1 * NS;
  long long own_lo = lo, own_hi = hi_all;
  if (STRIDE < IEM_TILE) {
    if (v0 >= STRIDE) return;
    const int vu = v1 < STRIDE ? v1 : STRIDE;
    const long long hi_u = P0 + (long long)vu * NS;
    if (!first) own_lo = (lo + 15) & ~15LL;
    own_hi = (hi_u + 15) & ~15LL;
    if (own_hi > hi_all) own_hi = hi_all;
  }
  const long long e_al = own_lo & ~15LL;
#pragma unroll
  for (int j = 0; j <= NS; ++j) {
    const long long e = e_al + t + (long long)j * IEM_TILE;
    if (e >= own_lo && e < own_hi) iem_stg(out + e, lds_reg[e - P0]);
  }
}
#else
template <int NS, int STRIDE>
__device__ __forceinline__ void iem_flush(double *__restrict__ out, long long P0, int v0, int v1, bool first,
                                          const double *__restrict__ lds_reg) {
  if (STRIDE < IEM_TILE && v0 >= STRIDE) return;   // the row starts in the halo: the next workgroup owns all of it
  if (v1 <= v0) return;
  const long long lo = P0 + (long long)v0 * NS;
  const int vu = v1 < STRIDE ? v1 : STRIDE;
  // previous workgroup's reach past lo: its lanes end 16 lanes into this tile
  const int prev = first ? -1 : ((v1 < 16 ? v1 : 16) - v0) * NS;
  iem_flush_rel<NS, STRIDE>(out + lo, lds_reg + v0 * NS, (int)(lo & 15), (vu - v0) * NS, (v1 - v0) * NS, prev);
}
#endif

// ---- flat 2-D grids: block store by ITEM ORDINAL -----------------------------------------------
// A 2-D support grid (E0 x E1, first coordinate fastest) walked by ONE linear lane index has no
// partly filled workgroup at the end of every row.  A template whose item box is a sub-box of the
// grid (the difference rows t = 2..Nt) then skips lanes, so its items are not lane-linear any more —
// but they are still consecutive in ORDINAL: ord_lt(q) = number of the template's items at flat
// indices < q.  Lanes stage at (ord - ord_lt(first lane)) and the workgroup writes the contiguous
// positions of the ordinals it owns; ownership is the whole-line rule of iem_flush, stated in
// ordinals: workgroup b owns the lines that start in [lo_b, lo_{b+1}) rounded up, never past what
// its own lanes (halo included) computed.
// (row, column) of a workgroup's first flat index: one division per workgroup, through a double
// reciprocal (exact after one correction step for q < 2^52) instead of the 64-bit integer sequence
__device__ __forceinline__ void iem_flat_base(long long q, long long E0, long long &r1, long long &r0) {
  long long d = (long long)((double)q / (double)E0);
  long long r = q - d * E0;
  if (r < 0) { --d; r += E0; }
  if (r >= E0) { ++d; r -= E0; }
  r1 = d; r0 = r;
}
// (r1, r0) = (row, column) of flat index  fr1*E0 + fr0 + d,  0 <= fr0 < E0, 0 <= d <= IEM_TILE
__device__ __forceinline__ void iem_flat_split(long long fr1, long long fr0, int d, long long E0, long long &r1, long long &r0) {
  const long long t = fr0 + d;
  const long long adv = E0 >= 2 * IEM_TILE ? (t >= E0 ? 1LL : 0LL) : (long long)((unsigned)t / (unsigned)E0);   // t < E0 + IEM_TILE
  r1 = fr1 + adv;
  r0 = t - adv * E0;
}
// items of the box [lo0, lo0 + w0) x [lo1, lo1 + h1) at flat positions before (r1, r0)
__device__ __forceinline__ long long iem_ord_lt(long long r1, long long r0, long long lo0, long long w0, long long lo1, long long h1) {
  long long rows = r1 - lo1;
  rows = rows < 0 ? 0 : (rows > h1 ? h1 : rows);
  long long in = 0;
  if (r1 >= lo1 && r1 < lo1 + h1) { in = r0 - lo0; in = in < 0 ? 0 : (in > w0 ? w0 : in); }
  return rows * w0 + in;
}
template <int NS>
__device__ __forceinline__ void iem_stage_ord(const double (&v)[NS], double *__restrict__ lds_reg, long long slot, bool valid) {
  if (valid) {
#pragma unroll
    for (int s = 0; s < NS; ++s) lds_reg[slot * NS + s] = v[s];
  }
}
// o = position of ordinal 0; ob / o16 / ou / oa = ord_lt at this workgroup's lane 0, lane 16 (= where the
// PREVIOUS workgroup's lanes end), lane STRIDE (where the next workgroup starts) and lane IEM_TILE
template <int NS, int STRIDE>
__device__ __forceinline__ void iem_flush_ord(double *__restrict__ out, long long o, long long ob, long long o16, long long ou,
                                              long long oa, const double *__restrict__ lds_reg, long long /*slot*/, bool /*valid*/) {
  if (oa <= ob) return;
  const long long lo = o + NS * ob;
  iem_flush_rel<NS, STRIDE>(out + lo, lds_reg, (int)(lo & 15), (int)(ou - ob) * NS, (int)(oa - ob) * NS, ob > 0 ? (int)(o16 - ob) * NS : -1);
}

template <int NS>
__device__ __forceinline__ void iem_store_block(double *__restrict__ out, long long P0, int v0, int v1,
                                                const double (&v)[NS], double *__restrict__ lds_blk) {
  iem_stage<NS>(v, lds_blk);
  __syncthreads();
  iem_flush<NS, IEM_TILE>(out, P0, v0, v1, true, lds_blk);
  __syncthreads();
}

// ---- reductions -------------------------------------------------------------
__device__ __forceinline__ double iem_wave_sum(double v) {
#pragma unroll
  for (int off = IEM_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, IEM_WAVE);
  return v;  // lane 0 holds the sum
}

// "Which workgroup of this call finishes last?" — two levels of ticket counters (T[0] = top, T[1+g] =
// group g of IEM_TICKET_GROUP workgroups; a single counter serialises ~2000 same-address atomics).
// Called by ONE thread of the workgroup, after the workgroup's hand-off stores (agent-scope
// write-through stores, each storing thread having waited `vmcnt(0)`, and a workgroup barrier behind
// them — MI355X_MICROARCH.md "Valid forms", R1) have completed.  The counters reset themselves.
#define IEM_TICKET_GROUP 32
__device__ __forceinline__ bool iem_last_arrival(unsigned long long *__restrict__ T, long long slot, long long n) {
  if (n <= IEM_TICKET_GROUP) {   // one level
    const bool last = __hip_atomic_fetch_add(T, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)(n - 1);
    if (last) __hip_atomic_store(T, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
    return last;
  }
  const long long g = slot / IEM_TICKET_GROUP, ng = (n + IEM_TICKET_GROUP - 1) / IEM_TICKET_GROUP;
  const long long gsize = g + 1 < ng ? IEM_TICKET_GROUP : n - g * IEM_TICKET_GROUP;
  bool last = false;
  if (__hip_atomic_fetch_add(T + 1 + g, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)(gsize - 1)) {
    __hip_atomic_store(T + 1 + g, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = __hip_atomic_fetch_add(T, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)(ng - 1);
    if (last) __hip_atomic_store(T, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return last;
}
// words behind `n` partials that the tickets need
#define IEM_TICKET_WORDS(n) (1 + ((n) + IEM_TICKET_GROUP - 1) / IEM_TICKET_GROUP)

// Objective.  The launch has AT MOST a fixed number of workgroups (generator knob obj_wgs): a workgroup
// walks the tiles b, b + gridDim.x, ... and every lane adds its tiles' terms in that order, so there is
// ONE partial per workgroup and one block reduction per workgroup however large the model is.  The
// workgroup that finishes LAST sums the n partials in a fixed order — thread t takes t, t+TILE, ...,
// then the wave shuffle tree, then the waves in order — and writes the scalar.  Which workgroup is
// last varies from run to run, the summation order does not: bitwise reproducible, no second launch.
// `lds4` holds IEM_TILE/64 + 1 doubles (the extra one is the "I am last" flag).
__device__ __forceinline__ void iem_block_partial(double v, double *__restrict__ partials, long long slot,
                                                  double *__restrict__ lds4, long long n, double *__restrict__ out) {
  constexpr int NW = IEM_TILE / IEM_WAVE;
  v = iem_wave_sum(v);
  if (iem_lane() == 0) lds4[iem_wave()] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double acc = lds4[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) acc += lds4[w];
    bool last = true;
    if (n > 1) {
      // agent-scope (write-through) store of the partial, its completion waited for, then the ticket.
      // (An agent-scope release FENCE would write back the whole L2 once per workgroup.)
      __hip_atomic_store(partials + slot, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      last = iem_last_arrival(reinterpret_cast<unsigned long long *>(partials + n), slot, n);
    } else {
      // a one-workgroup launch: no hand-off at all
      __hip_atomic_store(out, acc, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      last = false;
    }
    lds4[NW] = last ? 1.0 : 0.0;
  }
  __syncthreads();
  if (lds4[NW] == 0.0) return;
  // the partials are read with agent-scope loads (they bypass this CU's L1)
  double acc = 0.0;
  for (long long i = threadIdx.x; i < n; i += IEM_TILE)
    acc += __hip_atomic_load(partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  acc = iem_wave_sum(acc);
  __syncthreads();
  if (iem_lane() == 0) lds4[iem_wave()] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = lds4[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) tot += lds4[w];
    // system scope: `out` may be mapped host memory that iem_obj polls for the value
    __hip_atomic_store(out, tot, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Deterministic sums for output entries that MANY items share (the gradient / J'v / Hv entries of
// finite and first-stage variables: farmer 3, stochastic OPF 24): instead of one f64 atomic per wave
// — whose arrival order, hence rounding, changes from run to run — every workgroup of the call
// reduces its lanes' NV contributions in a fixed order and parks them in `red` (value s of global
// workgroup w at red[off_s + w]); the workgroup that finishes last (iem_last_arrival) sums each
// value's column in a fixed order (one wave per value: lane l takes l, l+64, ..., shuffle tree) and
// leaves the NV totals in lds[0 .. NV) for the generated epilogue, which writes each destination
// entry once.  Layout of `red`: values, then the ticket words.  `lds`: at least NV*(NW+1)+1 doubles.
//   step 1 (every workgroup):  iem_shared_park<NV>(v, red, offs, wg, lds)
//   step 2:                    if (iem_shared_last(red_tickets, wg, n_wg, lds_flag)) { iem_shared_totals<NV>(...); epilogue }
template <int NV>
__device__ __forceinline__ void iem_shared_park(const double (&v)[NV], double *__restrict__ red,
                                                const long long *__restrict__ offs, long long wg,
                                                double *__restrict__ lds) {
  constexpr int NW = IEM_TILE / IEM_WAVE;
#pragma unroll
  for (int s = 0; s < NV; ++s) {
    const double r = iem_wave_sum(v[s]);
    if (iem_lane() == 0) lds[s * NW + iem_wave()] = r;
  }
  __syncthreads();
  for (int s = (int)threadIdx.x; s < NV; s += IEM_TILE) {
    double acc = lds[s * NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) acc += lds[s * NW + w];
    __hip_atomic_store(red + offs[s] + wg, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing thread, before the barrier the ticket sits behind
  __syncthreads();
}
__device__ __forceinline__ bool iem_shared_last(double *__restrict__ tickets, long long wg, long long n_wg,
                                                double *__restrict__ lds_flag) {
  if (threadIdx.x == 0)
    *lds_flag = (n_wg <= 1 || iem_last_arrival(reinterpret_cast<unsigned long long *>(tickets), wg, n_wg)) ? 1.0 : 0.0;
  __syncthreads();
  return *lds_flag != 0.0;
}
// value s was parked by the workgroups [first[s], first[s] + count[s]) of the call
__device__ __forceinline__ void iem_shared_totals(int nv, const double *__restrict__ red, const long long *__restrict__ offs,
                                                  const long long *__restrict__ first, const long long *__restrict__ count,
                                                  double *__restrict__ lds) {
  constexpr int NW = IEM_TILE / IEM_WAVE;
  for (int s = iem_wave(); s < nv; s += NW) {
    const double *__restrict__ col = red + offs[s] + first[s];
    double acc = 0.0;
    for (long long i = iem_lane(); i < count[s]; i += IEM_WAVE)
      acc += __hip_atomic_load(col + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    acc = iem_wave_sum(acc);
    if (iem_lane() == 0) lds[s] = acc;
  }
  __syncthreads();
}
// destination d = {entry, start, cnt}: OUT[entry] = lds[ids[start]] + ... + lds[ids[start + cnt - 1]], written once
__device__ __forceinline__ void iem_shared_write(double *__restrict__ out, const double *__restrict__ lds,
                                                 const long long *__restrict__ dst, const long long *__restrict__ ids, int nd) {
  for (int d = (int)threadIdx.x; d < nd; d += IEM_TILE) {
    const long long start = dst[3 * d + 1], cnt = dst[3 * d + 2];
    double acc = lds[ids[start]];
    for (long long j = 1; j < cnt; ++j) acc += lds[ids[start + j]];
    out[dst[3 * d]] = acc;
  }
}

// gradient entry shared by every lane of the wave (index does not depend on q0)
// Fused memset: workgroup b of nb zeroes its contiguous share of p[0, n) (chunks are multiples of
// 16 doubles so that every workgroup but the first starts on a 128-byte line of the range).
__device__ __forceinline__ void iem_zero_fill(double *__restrict__ p, long long n, long long b, long long nb) {
  long long chunk = (n + nb - 1) / nb;
  chunk = (chunk + 15) & ~15LL;
  const long long lo = b * chunk, hi = lo + chunk < n ? lo + chunk : n;
  for (long long i = lo + threadIdx.x; i < hi; i += IEM_TILE) iem_stg(p + i, 0.0);
}

__device__ __forceinline__ void iem_grad_wave_uniform(double *__restrict__ g, long long idx, double v, bool valid) {
  v = iem_wave_sum(valid ? v : 0.0);
  const unsigned long long m = __ballot(valid);
  if (m != 0ull && iem_lane() == 0) atomicAdd(&g[idx], v);
}

__device__ __forceinline__ void iem_grad_atomic(double *__restrict__ g, long long idx, double v, bool valid) {
  if (valid) atomicAdd(&g[idx], v);
}
// IEM-TILE-REGION-END

// ---- jac_structure! / hess_structure! on the device -----------------------------------------
// One launch per template: thread e handles COO element e of the template's block
// (item k = e / nslots, slot s = e % nslots), so rows/cols are written as two coalesced
// int64 streams.  An index expression is  c + sum_d k[d]*k_d + sum_j gcoef[j]*IA[garr[j]][gbase[j] + sum_d gstep[j][d]*k_d]
// (1-based variable index); Hessian entries are emitted lower-triangular (row >= col).
struct IemIdxDesc {
  long long c, k[3];
  int ng, garr[3];
  long long gcoef[3], gbase[3], gstep[3][3];
};
struct IemStructArgs {
  long long *rows, *cols;
  const IemIdxDesc *ia, *ib;       // per slot; ib == nullptr for the Jacobian
  const long long *const *iarrs;   // device int64 columns
  long long dims0, dims1, n_items, o, o0, base;
  int nslots;
};
__device__ __forceinline__ long long iem_eval_idx(const IemIdxDesc &d, const long long *const *iarrs, long long k0,
                                                 long long k1, long long k2) {
  long long v = d.c + d.k[0] * k0 + d.k[1] * k1 + d.k[2] * k2;
  for (int j = 0; j < d.ng; ++j)
    v += d.gcoef[j] * iarrs[d.garr[j]][d.gbase[j] + d.gstep[j][0] * k0 + d.gstep[j][1] * k1 + d.gstep[j][2] * k2];
  return v;
}
extern "C" __global__ __launch_bounds__(IEM_BLOCK) void iem_structure_kernel(const IemStructArgs A) {
  const long long e = (long long)blockIdx.x * IEM_BLOCK + threadIdx.x;
  if (e >= A.n_items * A.nslots) return;
  const long long k = e / A.nslots;
  const int s = (int)(e - k * A.nslots);
  const long long k0 = k % A.dims0, k1 = (k / A.dims0) % A.dims1, k2 = k / (A.dims0 * A.dims1);
  const long long a = iem_eval_idx(A.ia[s], A.iarrs, k0, k1, k2);
  if (A.ib == nullptr) {
    A.rows[A.o + e] = A.o0 + k + A.base;
    A.cols[A.o + e] = a - 1 + A.base;
  } else {
    const long long b = iem_eval_idx(A.ib[s], A.iarrs, k0, k1, k2);
    A.rows[A.o + e] = (a >= b ? a : b) - 1 + A.base;
    A.cols[A.o + e] = (a >= b ? b : a) - 1 + A.base;
  }
}

// ---- COO -> CSR value assembly (the step right after jac_coord!/hess_coord! in a solver) -----
// One thread per CSR nonzero: sums its (few) COO duplicates in a fixed order — deterministic,
// no atomics.  seg/perm come from a one-off plan (sort of the COO positions by (row, col)).
extern "C" __global__ __launch_bounds__(IEM_BLOCK) void iem_csr_gather_sum(const long long *__restrict__ seg,
                                                                          const long long *__restrict__ perm,
                                                                          const double *__restrict__ coo,
                                                                          double *__restrict__ csr, long long n) {
  const long long i = (long long)blockIdx.x * IEM_BLOCK + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
  for (long long k = seg[i]; k < seg[i + 1]; ++k) acc += coo[perm[k]];
  csr[i] = acc;
}

// the same with 32-bit plan words (COO entries < 2^32): the plan is half of what the kernel reads
extern "C" __global__ __launch_bounds__(IEM_BLOCK) void iem_csr_gather_sum32(const unsigned int *__restrict__ seg,
                                                                            const unsigned int *__restrict__ perm,
                                                                            const double *__restrict__ coo,
                                                                            double *__restrict__ csr, long long n) {
  const long long i = (long long)blockIdx.x * IEM_BLOCK + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
  for (unsigned int k = seg[i]; k < seg[i + 1]; ++k) acc += coo[perm[k]];
  csr[i] = acc;
}

// y = A x for a CSR matrix with 32-bit indices (the assembled KKT matrix: residuals of the chain solver's iterative
// refinement, kkt_chain.py).  Rows are short (4-5 entries on average): one thread per row — except the few LONG rows (the
// column of a first-stage variable in J' has an entry per scenario: 1e5 of them), which `skip_longer` leaves to
// iem_csr_spmv_long_kernel: one workgroup per listed row, fixed summation order (strided partials, LDS tree).
extern "C" __global__ __launch_bounds__(IEM_BLOCK) void iem_csr_spmv_kernel(const int *__restrict__ rowptr, const int *__restrict__ colind,
                                                                           const double *__restrict__ vals, const double *__restrict__ x,
                                                                           double *__restrict__ y, long long n, int skip_longer) {
  const long long i = (long long)blockIdx.x * IEM_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int lo = rowptr[i], hi = rowptr[i + 1];
  if (skip_longer > 0 && hi - lo > skip_longer) return;
  double acc = 0.0;
  for (int k = lo; k < hi; ++k) acc += vals[k] * x[colind[k]];
  y[i] = acc;
}
extern "C" __global__ __launch_bounds__(IEM_BLOCK) void iem_csr_spmv_long_kernel(const int *__restrict__ rowptr, const int *__restrict__ colind,
                                                                                const double *__restrict__ vals, const double *__restrict__ x,
                                                                                double *__restrict__ y, const long long *__restrict__ rows) {
  __shared__ double part_[IEM_BLOCK];
  const long long i = rows[blockIdx.x];
  double acc = 0.0;
  for (int k = rowptr[i] + (int)threadIdx.x; k < rowptr[i + 1]; k += IEM_BLOCK) acc += vals[k] * x[colind[k]];
  part_[threadIdx.x] = acc;
  __syncthreads();
  for (int s = IEM_BLOCK / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) part_[threadIdx.x] += part_[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) y[i] = part_[0];
}

// ---- sums over a non-lane axis of a scatter kind (Program::AxisSum) ----------------------------------
// The kind's kernels parked one addend per item at aux[off + row*n0 + lane]; entry e (blockIdx.y) writes
// out[c + k0*lane] = the sum of the lane's column.  A workgroup takes 64 lanes x 4 row groups: thread (l, g) adds
// the rows g, g+4, g+8, ... of lane l into four interleaved accumulators (sixteen coalesced loads in flight per
// lane instead of one dependent chain), the groups meet in LDS — a FIXED association of the addends, so the sum is
// bitwise reproducible; no atomics.  tab: 5 words per entry {c, k0, n0, rows, off}.
extern "C" __global__ __launch_bounds__(IEM_BLOCK) void iem_axis_sum_kernel(double *__restrict__ out, const double *__restrict__ aux,
                                                                           const long long *__restrict__ tab) {
  __shared__ double part_[IEM_BLOCK];
  const long long *__restrict__ t = tab + 5 * (long long)blockIdx.y;
  const int l = (int)threadIdx.x & 63, g = (int)threadIdx.x >> 6;
  const long long lane = (long long)blockIdx.x * 64 + l;
  const long long n0 = t[2], rows = t[3];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (lane < n0) {
    const double *__restrict__ col = aux + t[4] + lane;
    long long r = g;
    for (; r + 12 < rows; r += 16) {
      a0 += col[r * n0]; a1 += col[(r + 4) * n0]; a2 += col[(r + 8) * n0]; a3 += col[(r + 12) * n0];
    }
    for (; r < rows; r += 4) a0 += col[r * n0];
  }
  part_[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (g == 0 && lane < n0) out[t[0] + t[1] * lane] = (part_[l] + part_[64 + l]) + (part_[128 + l] + part_[192 + l]);
}

// ---- plan-driven gather of parked addends (Program::Gather) -------------------------------------------
// One thread per output entry: its addends were parked by the kind's kernels (exclusive coalesced stores), the plan
// (built on the host from the index expressions) lists where; summed in plan order, written once — the form every
// scatter addend takes that is neither exclusive, nor a shared entry, nor an axis sum.  No atomics, reproducible.
// dest[i] < 0 (= ~entry): the entry also has an exclusive writer among the kind's kernels; the sum is ADDED to it.
extern "C" __global__ __launch_bounds__(IEM_BLOCK) void iem_gather_sum_kernel(double *__restrict__ out, const double *__restrict__ parked,
                                                                             const long long *__restrict__ dest, const long long *__restrict__ seg,
                                                                             const void *__restrict__ perm, long long n, int wide) {
  const long long i = (long long)blockIdx.x * IEM_BLOCK + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
  if (wide) {   // positions beyond 2^31: 64-bit plan
    const long long *__restrict__ p = static_cast<const long long *>(perm);
    for (long long k = seg[i]; k < seg[i + 1]; ++k) acc += parked[p[k]];
  } else {
    const unsigned int *__restrict__ p = static_cast<const unsigned int *>(perm);
    for (long long k = seg[i]; k < seg[i + 1]; ++k) acc += parked[p[k]];
  }
  const long long d = dest[i];
  if (d >= 0) out[d] = acc;      // nobody else writes the entry
  else out[~d] += acc;           // deferred addends of a few items: added to what the kind's kernels stored there
}

// ---- multi-GPU: halo exchange and the one small all-reduce of the path --------------------------
// One process per GPU; every rank owns a MAILBOX in its HBM that its peers map through HIP IPC
// (iem_comm_export / iem_comm_connect).  Both kernels PUSH: a rank writes its few doubles straight
// into the peer's mailbox (over xGMI when the peer is another GPU), then a sequence flag; the peer
// spins on its OWN memory.  One-shot, latency-bound — never a ring (the payload is 8*(1+n_shared)
// bytes for the all-reduce, 8*reach doubles per sharded slab for the halo).  All traffic is
// system-scope (write-through stores, cache-bypassing loads); data is ordered before its flag by
// vmcnt(0) on every storing thread, a workgroup barrier and a system release fence.  Every wait is
// bounded (option "comm_timeout_ms", default 5 s, counted on the 100 MHz wall clock): on expiry the kernel records
// an error (IemCommErr), POISONS what it was to deliver — NaN into the halo entries / the folded entries / the
// objective and the replicated gradient entries, so nothing downstream can consume stale data silently — and
// RUNS ON: no wave ever spins forever.  Sequence numbers live in
// the mailbox and are advanced by the kernels themselves, so both calls can be graph-replayed.
//
// mailbox words (8 bytes each):   [0] status  [1] halo seq  [3] halo ack (from the right)
//   [4,6) halo flags (from the left, one per parity)   [8, 8+G) all-reduce seq per chunk
//   [8+G, 8+G+2WG) all-reduce flags [parity][peer][chunk]
//   then halo data [2][NH] and all-reduce data [2][W][NR]     (W = world, G = chunks, NH, NR as passed)
#define IEM_MB_STATUS 0
#define IEM_MB_HSEQ 1
#define IEM_MB_HACK 3
#define IEM_MB_HFLAG 4
#define IEM_MB_FSEQ 8     // halo FOLD (the transposed exchange: halo copies -> the left neighbour's owned entries)
#define IEM_MB_FACK 9
#define IEM_MB_FFLAG 10   // two parities
#define IEM_MB_RSEQ 12
__device__ __forceinline__ long long iem_mb_rflag(long long G) { return IEM_MB_RSEQ + G; }
__device__ __forceinline__ long long iem_mb_hdata(long long W, long long G) { return IEM_MB_RSEQ + G + 2 * W * G; }
__device__ __forceinline__ long long iem_mb_rdata(long long W, long long G, long long NH) { return iem_mb_hdata(W, G) + 2 * NH; }
__device__ __forceinline__ long long iem_mb_fdata(long long W, long long G, long long NH, long long NR) { return iem_mb_rdata(W, G, NH) + 2 * W * NR; }
__device__ __forceinline__ unsigned long long iem_sys_load(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void iem_sys_store(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double iem_sys_loadd(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void iem_sys_stored(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// Where a time-out is recorded: the mailbox's status word (device memory; iem_comm_status) and a word of mapped
// host memory the library's next host synchronisation point reads WITHOUT a copy (iem_obj, iem_obj_end,
// iem_synchronize return IEM_E_COMM and clear both).  `ticks`: the bound in ticks of the 100 MHz wall clock.
struct IemCommErr {
  unsigned long long *status;    // mailbox word 0
  unsigned long long *hstatus;   // mapped pinned host word (may be nullptr)
  long long ticks;
};
__device__ __forceinline__ void iem_comm_fail(const IemCommErr &E, unsigned long long err_bit) {
  const unsigned long long old = __hip_atomic_fetch_or(E.status, err_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (E.hstatus) __hip_atomic_store(E.hstatus, old | err_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// bounded wait until *p >= want; false (and the error recorded) on time-out
__device__ __forceinline__ bool iem_wait_ge(const unsigned long long *p, unsigned long long want, const IemCommErr &E,
                                            unsigned long long err_bit) {
  const long long t0 = wall_clock64();
  while (iem_sys_load(p) < want) {
    __builtin_amdgcn_s_sleep(4);
    if (wall_clock64() - t0 > E.ticks) {
      iem_comm_fail(E, err_bit);
      return false;
    }
  }
  return true;
}
// publish: every thread's data stores have been issued; make them visible, then raise the flags
__device__ __forceinline__ void iem_publish_fence() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

struct IemHaloArgs {
  double *x;
  unsigned long long *mine, *left, *right;   // mailboxes (left / right: nullptr at the ends of the chain)
  const long long *src, *dst;                // NH positions of x each: what goes right, where the left's arrive
  long long NH, W, G;
  unsigned long long *hstatus; long long ticks;
};
// one workgroup: (1) my last `reach` owned supports of every sharded slab -> the right neighbour's
// mailbox, (2) the left neighbour's -> the halo entries of my x (reference stencil:
// /root/reference/src/transform.jl:535-557, index i-1 at :471-506).
// Runs either as the stand-alone iem_halo_kernel (stream-ordered: iem_halo_exchange) or as the LEADING workgroup of an
// evaluation kernel that cannot touch a halo entry itself (iem_halo_exchange_async: the exchange rides on the first
// such launch — no launch, no stream and no event of its own — and overlaps that kernel's other workgroups; kernels
// launched behind it see the halo entries in x).  Any workgroup size.
__device__ __forceinline__ void iem_halo_wg(const IemHaloArgs &A, double *__restrict__ x) {
  const IemCommErr E = {A.mine + IEM_MB_STATUS, A.hstatus, A.ticks};
  const unsigned long long seq = iem_sys_load(A.mine + IEM_MB_HSEQ) + 1;
  const long long par = (long long)(seq & 1);
  const long long nt = (long long)blockDim.x;
  __shared__ int ok_;
  if (A.right != nullptr) {
    if (threadIdx.x == 0)   // the slot of this parity was last used by seq - 2: the right neighbour must have consumed it
      ok_ = seq <= 2 || iem_wait_ge(A.mine + IEM_MB_HACK, seq - 2, E, 1ULL);
    __syncthreads();
    double *data = reinterpret_cast<double *>(A.right + iem_mb_hdata(A.W, A.G)) + par * A.NH;
    for (long long e = threadIdx.x; e < A.NH; e += nt) iem_sys_stored(data + e, x[A.src[e]]);
    iem_publish_fence();
    if (threadIdx.x == 0) iem_sys_store(A.right + IEM_MB_HFLAG + par, seq);
  }
  if (A.left != nullptr) {
    if (threadIdx.x == 0) {
      ok_ = iem_wait_ge(A.mine + IEM_MB_HFLAG + par, seq, E, 2ULL);
      __threadfence_system();
    }
    __syncthreads();
    const double *data = reinterpret_cast<const double *>(A.mine + iem_mb_hdata(A.W, A.G)) + par * A.NH;
    for (long long e = threadIdx.x; e < A.NH; e += nt) x[A.dst[e]] = ok_ ? iem_sys_loadd(data + e) : __builtin_nan("");   // time-out: poisoned, never stale
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) iem_sys_store(A.left + IEM_MB_HACK, seq);
  }
  __syncthreads();
  if (threadIdx.x == 0) iem_sys_store(A.mine + IEM_MB_HSEQ, seq);
}
extern "C" __global__ __launch_bounds__(IEM_BLOCK) void iem_halo_kernel(const IemHaloArgs A) { iem_halo_wg(A, A.x); }

// The transposed exchange, for vectors in VARIABLE space that a transposed operator produced (J'v): the entries of
// my halo copies hold what my rows owe to the variables the LEFT neighbour owns.  (1) they go into the left
// neighbour's mailbox and are zeroed here, (2) what the right neighbour sent is ADDED to my owned entries — one
// addend per entry, so the result does not depend on arrival order.  Same flags / parity / bounded-wait scheme as
// iem_halo_kernel, words of its own (IEM_MB_F*).
struct IemFoldArgs {
  double *vec;
  unsigned long long *mine, *left, *right;
  const long long *src, *dst;                // as in IemHaloArgs: src = my owned rows the right neighbour copies, dst = my halo copies
  long long NH, W, G, NR;
  unsigned long long *hstatus; long long ticks;
};
extern "C" __global__ __launch_bounds__(IEM_BLOCK) void iem_halo_fold_kernel(const IemFoldArgs A) {
  const IemCommErr E = {A.mine + IEM_MB_STATUS, A.hstatus, A.ticks};
  const unsigned long long seq = iem_sys_load(A.mine + IEM_MB_FSEQ) + 1;
  const long long par = (long long)(seq & 1);
  __shared__ int ok_;
  if (A.left != nullptr) {
    if (threadIdx.x == 0)
      ok_ = seq <= 2 || iem_wait_ge(A.mine + IEM_MB_FACK, seq - 2, E, 8ULL);
    __syncthreads();
    double *data = reinterpret_cast<double *>(A.left + iem_mb_fdata(A.W, A.G, A.NH, A.NR)) + par * A.NH;
    for (long long e = threadIdx.x; e < A.NH; e += IEM_BLOCK) {
      iem_sys_stored(data + e, A.vec[A.dst[e]]);
      A.vec[A.dst[e]] = 0.0;
    }
    iem_publish_fence();
    if (threadIdx.x == 0) iem_sys_store(A.left + IEM_MB_FFLAG + par, seq);
  }
  if (A.right != nullptr) {
    if (threadIdx.x == 0) {
      ok_ = iem_wait_ge(A.mine + IEM_MB_FFLAG + par, seq, E, 16ULL);
      __threadfence_system();
    }
    __syncthreads();
    const double *data = reinterpret_cast<const double *>(A.mine + iem_mb_fdata(A.W, A.G, A.NH, A.NR)) + par * A.NH;
    for (long long e = threadIdx.x; e < A.NH; e += IEM_BLOCK) A.vec[A.src[e]] = ok_ ? A.vec[A.src[e]] + iem_sys_loadd(data + e) : __builtin_nan("");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) iem_sys_store(A.right + IEM_MB_FACK, seq);
  }
  __syncthreads();
  if (threadIdx.x == 0) iem_sys_store(A.mine + IEM_MB_FSEQ, seq);
}

struct IemReduceArgs {
  double *obj, *g;                    // in/out: the scalar objective (device) and the gradient
  const long long *shared;            // NR - 1 positions of g held by every rank (replicated variables)
  unsigned long long *const *peers;   // W mailboxes, peers[rank] = mine
  long long NR, NH, W, rank, G;
  unsigned long long *hstatus; long long ticks;
};
// G workgroups, each on its own chunk of the NR doubles and with flags of its own (no cross-workgroup
// step): my chunk -> slot [rank] of EVERY rank's mailbox; wait for the W flags of the chunk in mine; sum
// the W slots in rank order (the same order on every rank: identical bits everywhere) and write the sums
// back.  SURVEY 8(e): "one small all-reduce per obj / grad!" — one-shot direct writes, never a ring.
extern "C" __global__ __launch_bounds__(IEM_BLOCK) void iem_allreduce_kernel(const IemReduceArgs A) {
  unsigned long long *mine = A.peers[A.rank];
  const IemCommErr E = {mine + IEM_MB_STATUS, A.hstatus, A.ticks};
  const long long c = blockIdx.x, G = A.G;
  const long long per = (A.NR + G - 1) / G, e0 = c * per, e1 = e0 + per < A.NR ? e0 + per : A.NR;
  const unsigned long long seq = iem_sys_load(mine + IEM_MB_RSEQ + c) + 1;
  const long long par = (long long)(seq & 1);
  const long long base = iem_mb_rdata(A.W, G, A.NH) + (par * A.W + A.rank) * A.NR;
  for (long long e = e0 + threadIdx.x; e < e1; e += IEM_BLOCK) {
    const double v = e == 0 ? (A.obj ? *A.obj : 0.0) : A.g[A.shared[e - 1]];
    for (long long p = 0; p < A.W; ++p) iem_sys_stored(reinterpret_cast<double *>(A.peers[p] + base) + e, v);
  }
  iem_publish_fence();
  if (threadIdx.x == 0)
    for (long long p = 0; p < A.W; ++p) iem_sys_store(A.peers[p] + iem_mb_rflag(G) + (par * A.W + A.rank) * G + c, seq);
  __shared__ int ok_;
  if (threadIdx.x == 0) ok_ = 1;
  __syncthreads();
  for (long long p = threadIdx.x; p < A.W; p += IEM_BLOCK)
    if (!iem_wait_ge(mine + iem_mb_rflag(G) + (par * A.W + p) * G + c, seq, E, 4ULL)) ok_ = 0;
  __threadfence_system();
  __syncthreads();
  {
    const double *slots = reinterpret_cast<const double *>(mine + iem_mb_rdata(A.W, G, A.NH)) + par * A.W * A.NR;
    for (long long e = e0 + threadIdx.x; e < e1; e += IEM_BLOCK) {
      double acc = __builtin_nan("");   // a peer's contribution never arrived: the sum is poisoned, not skipped
      if (ok_) {
        acc = iem_sys_loadd(slots + e);
        for (long long p = 1; p < A.W; ++p) acc += iem_sys_loadd(slots + p * A.NR + e);
      }
      if (e == 0) { if (A.obj) *A.obj = acc; } else A.g[A.shared[e - 1]] = acc;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) iem_sys_store(mine + IEM_MB_RSEQ + c, seq);
}

#endif  // IEM_DEVICE_H

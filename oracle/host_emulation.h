// host_emulation.h — host stand-ins for the device primitives of iem_device.h so that the
// GENERATED kernel bodies can be compiled with g++ and run one lane at a time.
// TEST / BASELINE INFRASTRUCTURE ONLY (like everything under oracle/): used by tests/emu.py
// (generator checks without a GPU) and by oracle/cpu_compiled_baseline.py (a compiled,
// per-template CPU baseline for bench.py).  The HIP primitives themselves run only on the GPU.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>

#define IEM_BLOCK 256
#ifndef IEM_TILE
#define IEM_TILE 256
#endif
#define IEM_WAVE 64
#define __global__
#define __device__
#define __forceinline__ inline
#define __launch_bounds__(x)
#define __restrict__
#define __shared__ static thread_local

struct emu_dim3 { unsigned x = 0, y = 0, z = 0; };
static thread_local emu_dim3 threadIdx, blockIdx, gridDim;

// the deferred halo exchange that rides on an evaluation launch as an extra leading workgroup (iem_device.h: iem_halo_wg):
// here it only counts its lanes, so that tests/emu.py can launch a kernel WITH the extra workgroup and check that every
// other workgroup still evaluates its own tile
struct IemHaloArgs { long long calls; };
inline void iem_halo_wg(const IemHaloArgs &A, double *) { ++const_cast<IemHaloArgs &>(A).calls; }

inline long long iem_xcd_remap(long long b, long long nb) {
  const long long p = nb >> 3, r = nb & 7, x = b & 7, i = b >> 3;
  return x * p + (x < r ? x : r) + i;
}
inline int iem_lane() { return (int)(threadIdx.x & 63); }
inline int iem_wave() { return (int)(threadIdx.x >> 6); }

template <int NS>
inline void iem_store_rows_direct(double *out, long long pos0, bool valid, const double (&v)[NS]) {
  if (valid) for (int s = 0; s < NS; ++s) out[pos0 + s] = v[s];
}
template <int NS>
inline void iem_store_rows(double *out, long long pos0, bool valid, const double (&v)[NS], double *) {
  iem_store_rows_direct<NS>(out, pos0, valid, v);
}
// IEM-TILE-REGION-BEGIN (as in csrc/iem_device.h: tests/emu.py repeats this region per workgroup size, each copy in a
// namespace iem_t<size>, for programs whose kernels use more than one)
inline int iem_clamp256(long long v) { return v < 0 ? 0 : (v > IEM_TILE ? IEM_TILE : (int)v); }
template <int NS>
inline void iem_store_block(double *out, long long P0, int v0, int v1, const double (&v)[NS], double *) {
  const int t = (int)threadIdx.x;
  if (t >= v0 && t < v1) for (int s = 0; s < NS; ++s) out[P0 + (long long)t * NS + s] = v[s];
}
// split form: one lane at a time, the lane reads back its own staged values
template <int NS>
inline void iem_stage(const double (&v)[NS], double *lds_reg) {
  const int t = (int)threadIdx.x;
  for (int s = 0; s < NS; ++s) lds_reg[t * NS + s] = v[s];
}
// same ownership rule as the device version (which lines a workgroup writes), lane by lane
template <int NS, int STRIDE>
inline void iem_flush(double *out, long long P0, int v0, int v1, bool first, const double *lds_reg) {
  const int t = (int)threadIdx.x;
  const long long lo = P0 + (long long)v0 * NS, hi_all = P0 + (long long)v1 * NS;
  long long own_lo = lo, own_hi = hi_all;
  if (STRIDE < IEM_TILE) {
    if (v0 >= STRIDE) return;
    const int vu = v1 < STRIDE ? v1 : STRIDE;
    if (!first) own_lo = (lo + 15) & ~15LL;
    own_hi = (P0 + (long long)vu * NS + 15) & ~15LL;
    if (own_hi > hi_all) own_hi = hi_all;
  }
  if (t < v0 || t >= v1) return;
  for (int s = 0; s < NS; ++s) {
    const long long e = P0 + (long long)t * NS + s;
    if (e >= own_lo && e < own_hi) out[e] = lds_reg[t * NS + s];
  }
}
inline void iem_flat_base(long long q, long long E0, long long &r1, long long &r0) { r1 = q / E0; r0 = q - r1 * E0; }
inline void iem_flat_split(long long fr1, long long fr0, int d, long long E0, long long &r1, long long &r0) {
  const long long t = fr0 + d, adv = t / E0;
  r1 = fr1 + adv;
  r0 = t - adv * E0;
}
inline long long iem_ord_lt(long long r1, long long r0, long long lo0, long long w0, long long lo1, long long h1) {
  long long rows = r1 - lo1;
  rows = rows < 0 ? 0 : (rows > h1 ? h1 : rows);
  long long in = 0;
  if (r1 >= lo1 && r1 < lo1 + h1) { in = r0 - lo0; in = in < 0 ? 0 : (in > w0 ? w0 : in); }
  return rows * w0 + in;
}
// one lane at a time: the lane remembers where it staged and writes exactly its own elements that
// fall into the range its workgroup owns (the device version's ownership rule)
static thread_local long long emu_ord_slot = -1;
template <int NS>
inline void iem_stage_ord(const double (&v)[NS], double *lds_reg, long long slot, bool valid) {
  emu_ord_slot = valid ? slot : -1;
  if (valid) for (int s = 0; s < NS; ++s) lds_reg[slot * NS + s] = v[s];
}
template <int NS, int STRIDE>
inline void iem_flush_ord(double *out, long long o, long long ob, long long o16, long long ou, long long oa, const double *lds_reg,
                          long long slot, bool valid) {
  const long long lo = o + NS * ob, hi_all = o + NS * oa;
  long long own_lo = lo, own_hi = hi_all;
  if (STRIDE < IEM_TILE) {
    const long long hi_u = o + NS * ou;
    if (ob > 0) {
      own_lo = (lo + 15) & ~15LL;
      const long long prev_all = o + NS * o16;
      if (own_lo > prev_all) own_lo = prev_all;
    }
    own_hi = (hi_u + 15) & ~15LL;
    if (own_hi > hi_all) own_hi = hi_all;
  }
  if (!valid) return;
  for (int s = 0; s < NS; ++s) {
    const long long e = lo + slot * NS + s;
    if (e >= own_lo && e < own_hi) out[e] = lds_reg[slot * NS + s];
  }
}
inline void __syncthreads() {}
inline void iem_block_partial(double v, double *partials, long long slot, double *, long long n, double *out) {
  partials[slot] += v;                       // lanes run one at a time: the partial accumulates in place
  if (threadIdx.x != IEM_TILE - 1) return;   // last lane of the workgroup takes the ticket
  double &ticket = partials[n];
  ticket += 1.0;
  if (ticket < (double)n) return;
  double tot = 0.0;
  for (long long i = 0; i < n; ++i) tot += partials[i];
  out[0] = tot;
  ticket = 0.0;
}
// deterministic shared-entry reduction, one lane at a time: values accumulate in place (the test
// driver hands a zeroed buffer), the LAST lane of a workgroup takes the ticket, and the workgroup
// whose ticket completes the count does the whole epilogue from that one lane
template <int NV>
inline void iem_shared_park(const double (&v)[NV], double *red, const long long *offs, long long wg, double *) {
  for (int s = 0; s < NV; ++s) red[offs[s] + wg] += v[s];
}
inline bool iem_shared_last(double *tickets, long long, long long n_wg, double *) {
  if (threadIdx.x != IEM_TILE - 1) return false;
  tickets[0] += 1.0;
  if (tickets[0] < (double)n_wg) return false;
  tickets[0] = 0.0;
  return true;
}
inline void iem_shared_totals(int nv, const double *red, const long long *offs, const long long *first, const long long *count, double *lds) {
  for (int s = 0; s < nv; ++s) {
    double acc = 0.0;
    for (long long i = 0; i < count[s]; ++i) acc += red[offs[s] + first[s] + i];
    lds[s] = acc;
  }
}
inline void iem_shared_write(double *out, const double *lds, const long long *dst, const long long *ids, int nd) {
  for (int d = 0; d < nd; ++d) {
    double acc = 0.0;
    for (long long j = 0; j < dst[3 * d + 2]; ++j) acc += lds[ids[dst[3 * d + 1] + j]];
    out[dst[3 * d]] = acc;
  }
}
inline void iem_zero_fill(double *p, long long n, long long b, long long nb) {
  long long chunk = (n + nb - 1) / nb;
  chunk = (chunk + 15) & ~15LL;
  const long long lo = b * chunk, hi = lo + chunk < n ? lo + chunk : n;
  for (long long i = lo + threadIdx.x; i < hi; i += IEM_TILE) p[i] = 0.0;
}
// IEM-TILE-REGION-END
inline void iem_grad_wave_uniform(double *g, long long idx, double v, bool valid) { if (valid) g[idx] += v; }
inline void iem_grad_atomic(double *g, long long idx, double v, bool valid) { if (valid) g[idx] += v; }

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
//     jac_coord!/hess_coord! define).  iem_store_rows<NS> stages a wave's NS×64
//     values in LDS and writes the wave's block — which is CONTIGUOUS in HBM — with
//     NS fully coalesced 512-byte stores instead of 64 strided 8-byte stores per
//     instruction; no atomics, no zero-fill pass;
//   * objective: wave shuffle reduction → LDS → one partial per workgroup, summed in a
//     fixed order by iem_reduce_partials (bitwise reproducible);
//   * gradient entries shared by many items (finite / first-stage variables):
//     wavefront reduction first, then one f64 atomic per wave.
//
// wave = 64 lanes on CDNA4; blocks are 256 threads = 4 waves, each wave owns a
// private LDS staging region, so no __syncthreads() is needed on the store path.
#ifndef IEM_DEVICE_H
#define IEM_DEVICE_H

#define IEM_BLOCK 256
#define IEM_WAVE 64

__device__ __forceinline__ int iem_lane() { return (int)(threadIdx.x & (IEM_WAVE - 1)); }
__device__ __forceinline__ int iem_wave() { return (int)(threadIdx.x >> 6); }

// order LDS traffic of one wave: LDS executes a wave's instructions in issue order, so a
// compiler-level fence plus lgkmcnt(0) is enough; no workgroup barrier.
__device__ __forceinline__ void iem_wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Direct form: lane writes its NS slots at stride NS (kept for A/B measurement).
template <int NS>
__device__ __forceinline__ void iem_store_rows_direct(double *__restrict__ out, long long pos0, bool valid,
                                                      const double (&v)[NS]) {
  if (valid) {
#pragma unroll
    for (int s = 0; s < NS; ++s) out[pos0 + s] = v[s];
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
#pragma unroll
  for (int s = 0; s < NS; ++s) lds_wave[lane * NS + s] = v[s];
  const int first = __ffsll((long long)m) - 1;
  // position of (virtual) lane 0, slot 0 — wave-uniform
  const long long base = __shfl(pos0, first, IEM_WAVE) - (long long)first * NS;
  iem_wave_lds_sync();
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int e = j * IEM_WAVE + lane;  // element of the wave's contiguous block
    const int src = e / NS;             // lane that produced it (NS is a compile-time constant)
    if ((m >> src) & 1ull) out[base + e] = lds_wave[e];
  }
  iem_wave_lds_sync();
}

// ---- reductions -------------------------------------------------------------
__device__ __forceinline__ double iem_wave_sum(double v) {
#pragma unroll
  for (int off = IEM_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, IEM_WAVE);
  return v;  // lane 0 holds the sum
}

// one partial per workgroup, fixed summation order
__device__ __forceinline__ void iem_block_partial(double v, double *__restrict__ partials, long long slot,
                                                  double *__restrict__ lds4) {
  v = iem_wave_sum(v);
  if (iem_lane() == 0) lds4[iem_wave()] = v;
  __syncthreads();
  if (threadIdx.x == 0) partials[slot] = ((lds4[0] + lds4[1]) + lds4[2]) + lds4[3];
}

// gradient entry shared by every lane of the wave (index does not depend on q0)
__device__ __forceinline__ void iem_grad_wave_uniform(double *__restrict__ g, long long idx, double v, bool valid) {
  v = iem_wave_sum(valid ? v : 0.0);
  const unsigned long long m = __ballot(valid);
  if (m != 0ull && iem_lane() == 0) atomicAdd(&g[idx], v);
}

__device__ __forceinline__ void iem_grad_atomic(double *__restrict__ g, long long idx, double v, bool valid) {
  if (valid) atomicAdd(&g[idx], v);
}

// final objective reduction: out[0] = sum(partials[0..n)), one workgroup, fixed order
extern "C" __global__ __launch_bounds__(IEM_BLOCK) void iem_reduce_partials(const double *__restrict__ partials,
                                                                           long long n, double *__restrict__ out) {
  __shared__ double lds[IEM_BLOCK];
  double acc = 0.0;
  for (long long i = threadIdx.x; i < n; i += IEM_BLOCK) acc += partials[i];
  lds[threadIdx.x] = acc;
  __syncthreads();
  for (int s = IEM_BLOCK / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) lds[threadIdx.x] += lds[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = lds[0];
}

// structure fill helpers (int64, one-off)
extern "C" __global__ __launch_bounds__(IEM_BLOCK) void iem_fill_i64(long long *__restrict__ p, long long n, long long v) {
  const long long i = (long long)blockIdx.x * IEM_BLOCK + threadIdx.x;
  if (i < n) p[i] = v;
}

#endif  // IEM_DEVICE_H

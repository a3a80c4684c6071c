// stream_order_probe.hip — how the ORDER in which a launch visits the template blocks of jac_coord! moves the
// write rate.  Pure stores (no arithmetic, no loads), the quadrotor's 18 COO blocks at 10^6 supports, item-major,
// 512-lane workgroups writing whole 128-byte lines, non-temporal.  Variants:
//   fill        one linear stream over the same bytes
//   all         workgroup b writes chunk b of ALL 18 blocks (the shape of iem_jac_g0: 18 fronts in flight chip-wide)
//   major G     the blocks in G groups, workgroup id = g * ntiles + b: the chip works through group 0, then group 1, ...
//   inter G     the same groups, workgroup id = b * G + g: groups interleaved (all fronts stay in flight)
// LDS is requested so that 3 workgroups fit a CU, like the real kernel (48 KB).
// Build: hipcc -O3 --offload-arch=gfx950 tools/stream_order_probe.hip -o tools/stream_order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NT = 18;
struct Plan {
  long o[NT];        // block offsets (doubles)
  int ns[NT];        // slots per support
  int gfirst[NT + 1];// groups: streams [gfirst[g], gfirst[g+1])
  int G;
  long ntiles, n;
  int tile;          // supports per workgroup
  int inter;
};
__device__ __forceinline__ void st(double *p, double v) { __builtin_nontemporal_store(v, p); }

__global__ __launch_bounds__(512) void k_fill(double *out, long tot) {
  // each workgroup a contiguous 496*62-double chunk, like the real kernel's bytes per workgroup
  const long per = 512L * 62;
  long lo = (long)blockIdx.x * per, hi = lo + per < tot ? lo + per : tot;
  for (long i = lo + threadIdx.x; i < hi; i += 512) st(out + i, 1.0);
}
__global__ __launch_bounds__(512) void k_groups(double *out, const Plan P) {
  extern __shared__ double dyn[];
  if (P.n < 0) dyn[threadIdx.x] = 1.0;
  long b; int g;
  if (P.inter) { b = blockIdx.x / P.G; g = (int)(blockIdx.x % P.G); }
  else { g = (int)(blockIdx.x / P.ntiles); b = blockIdx.x % P.ntiles; }
  const long s0 = b * P.tile;
  const long cnt = s0 + P.tile <= P.n ? P.tile : P.n - s0;
  for (int t = P.gfirst[g]; t < P.gfirst[g + 1]; ++t) {
    double *dst = out + P.o[t] + s0 * P.ns[t];
    const long m = cnt * P.ns[t];
    for (long i = threadIdx.x; i < m; i += 512) st(dst + i, 1.0);
  }
}
typedef double d2 __attribute__((ext_vector_type(2)));
// fill variants: W = bytes per lane per store (8 / 16), NTS = non-temporal, block size T, PER = doubles per workgroup
template <int W, int NTS, int T>
__global__ __launch_bounds__(T) void k_fillv(double *out, long tot, long per) {
  long lo = (long)blockIdx.x * per, hi = lo + per < tot ? lo + per : tot;
  if (W == 8) {
    for (long i = lo + threadIdx.x; i < hi; i += T) { if (NTS) __builtin_nontemporal_store(1.0, out + i); else out[i] = 1.0; }
  } else {
    d2 v = {1.0, 1.0};
    for (long i = lo + 2 * threadIdx.x; i < hi; i += 2 * T) { if (NTS) __builtin_nontemporal_store(v, (d2 *)(out + i)); else *(d2 *)(out + i) = v; }
  }
}
// the "all" shape with 16-byte stores
template <int NTS>
__global__ __launch_bounds__(512) void k_all16(double *out, const Plan P) {
  extern __shared__ double dyn[];
  if (P.n < 0) dyn[threadIdx.x] = 1.0;
  const long b = blockIdx.x;
  const long s0 = b * P.tile;
  const long cnt = s0 + P.tile <= P.n ? P.tile : P.n - s0;
  d2 v = {1.0, 1.0};
  for (int t = 0; t < NT; ++t) {
    double *dst = out + P.o[t] + s0 * P.ns[t];
    const long m = cnt * P.ns[t];
    for (long i = 2 * threadIdx.x; i < m; i += 1024) { if (NTS) __builtin_nontemporal_store(v, (d2 *)(dst + i)); else *(d2 *)(dst + i) = v; }
  }
}
// grid-stride fill (the shape of the runtime's own fill kernel): thread g writes 16-byte words g, g + G, g + 2G, ...
template <int NTS, int T>
__global__ __launch_bounds__(T) void k_gs(double *out, long n2) {
  d2 v = {1.0, 1.0};
  const long G = (long)gridDim.x * T;
  for (long i = (long)blockIdx.x * T + threadIdx.x; i < n2; i += G) { if (NTS) __builtin_nontemporal_store(v, (d2 *)out + i); else ((d2 *)out)[i] = v; }
}
// persistent "all": G workgroups walk the tiles b, b + G, ... (every stream's window = G tiles); TS supports per tile
template <int T>
__global__ __launch_bounds__(T) void k_walk(double *out, const Plan P, int w16) {
  extern __shared__ double dyn[];
  if (P.n < 0) dyn[threadIdx.x] = 1.0;
  for (long b = blockIdx.x; b < P.ntiles; b += gridDim.x) {
    const long s0 = b * P.tile;
    const long cnt = s0 + P.tile <= P.n ? P.tile : P.n - s0;
    for (int t = 0; t < NT; ++t) {
      double *dst = out + P.o[t] + s0 * P.ns[t];
      const long m = cnt * P.ns[t];
      if (w16) { d2 v = {1.0, 1.0}; for (long i = 2 * threadIdx.x; i < m; i += 2 * T) __builtin_nontemporal_store(v, (d2 *)(dst + i)); }
      else for (long i = threadIdx.x; i < m; i += T) st(dst + i, 1.0);
    }
  }
}
// mixed: the difference-row blocks (streams 9..17, contiguous in the COO buffer) as ONE linear narrow-window fill
// (workgroup w of the fill part writes doubles [w*per, (w+1)*per) of the region, 2 per lane per round), the nine ODE blocks
// in the "all" shape (512 supports per workgroup).  order 0: fill workgroups first, 1: ODE first, 2: interleaved 1:1 as far as both last
struct Mixed { long o[NT]; int ns[NT]; long n, ntiles, fill_lo, fill_n, per, nfill; int order; };
__global__ __launch_bounds__(512) void k_mixed(double *out, const Mixed P) {
  extern __shared__ double dyn[];
  if (P.n < 0) dyn[threadIdx.x] = 1.0;
  long b = blockIdx.x; bool fill; long w;
  if (P.order == 0) { fill = b < P.nfill; w = fill ? b : b - P.nfill; }
  else if (P.order == 1) { fill = b >= P.ntiles; w = fill ? b - P.ntiles : b; }
  else { const long m = P.nfill < P.ntiles ? P.nfill : P.ntiles;
         if (b < 2 * m) { fill = b & 1; w = b >> 1; } else { fill = P.nfill > P.ntiles; w = b - m; } }
  if (fill) {
    const long lo = P.fill_lo + w * P.per, hi = lo + P.per < P.fill_lo + P.fill_n ? lo + P.per : P.fill_lo + P.fill_n;
    d2 v = {1.0, 1.0};
    for (long i = lo + 2 * threadIdx.x; i < hi; i += 1024) __builtin_nontemporal_store(v, (d2 *)(out + i));
  } else {
    const long s0 = w * 512;
    const long cnt = s0 + 512 <= P.n ? 512 : P.n - s0;
    for (int t = 0; t < 9; ++t) {
      double *dst = out + P.o[t] + s0 * P.ns[t];
      const long m = cnt * P.ns[t];
      for (long i = threadIdx.x; i < m; i += 512) st(dst + i, 1.0);
    }
  }
}
template <class F> double timeit(F f, int iters) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) f();
  (void)hipEventRecord(a); for (int i = 0; i < iters; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / iters;
}
int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : 1000000L;
  const int nbuf = 3;
  int NS[NT] = {2, 5, 2, 5, 2, 4, 5, 4, 6, 3, 3, 3, 3, 3, 3, 3, 3, 3};
  long tot = 0; long o[NT];
  for (int t = 0; t < NT; ++t) { o[t] = tot; tot += ((n * NS[t] + 15) & ~15L); }
  double *buf[nbuf];
  for (int i = 0; i < nbuf; ++i) CK(hipMalloc(&buf[i], tot * 8));
  CK(hipFuncSetAttribute((const void *)k_groups, hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024));
  printf("supports %ld, bytes %.1f MB, %d buffers\n", n, tot * 8 / 1e6, nbuf);
  auto rep = [&](const char *name, int bi, double ms) { printf("%-28s buf %d  %8.4f ms  %7.1f GB/s\n", name, bi, ms, tot * 8.0 / ms / 1e6); fflush(stdout); };
  for (int round = 0; round < 2; ++round) {
    for (int bi = 0; bi < nbuf; ++bi) {
      double *p = buf[bi];
      rep("fill", bi, timeit([&] { k_fill<<<(tot + 512 * 62 - 1) / (512 * 62), 512>>>(p, tot); }, 30));
#define FV(W, N, T, PER) { char nm[64]; snprintf(nm, 64, "fillv w%d nt%d T%d per%ld", W, N, T, (long)(PER)); long per_ = (PER); \
      rep(nm, bi, timeit([&] { k_fillv<W, N, T><<<(tot + per_ - 1) / per_, T>>>(p, tot, per_); }, 30)); }
      FV(8, 1, 512, 512L * 62) FV(8, 0, 512, 512L * 62) FV(16, 1, 512, 512L * 62) FV(16, 0, 512, 512L * 62)
      FV(16, 1, 256, 256L * 4) FV(16, 0, 256, 256L * 4) FV(16, 1, 256, 256L * 16) FV(16, 0, 256, 256L * 16) FV(8, 1, 256, 256L * 4) FV(8, 0, 256, 256L * 4)
      FV(16, 1, 512, 512L * 8) FV(16, 0, 512, 512L * 8) FV(16, 1, 1024, 1024L * 8) FV(16, 0, 1024, 1024L * 8)
      { Plan P; P.n = n; P.tile = 512; P.ntiles = (n + 511) / 512; P.G = 1; P.inter = 0;
        for (int t = 0; t < NT; ++t) { P.o[t] = o[t]; P.ns[t] = NS[t]; }
        CK(hipFuncSetAttribute((const void *)k_all16<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024));
        CK(hipFuncSetAttribute((const void *)k_all16<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024));
        rep("all16 nt t512 lds48", bi, timeit([&] { k_all16<1><<<P.ntiles, 512, 48 * 1024>>>(p, P); }, 30));
        rep("all16 plain t512 lds48", bi, timeit([&] { k_all16<0><<<P.ntiles, 512, 48 * 1024>>>(p, P); }, 30));
        rep("all16 nt t512 lds16", bi, timeit([&] { k_all16<1><<<P.ntiles, 512, 16 * 1024>>>(p, P); }, 30));
        rep("all16 plain t512 lds16", bi, timeit([&] { k_all16<0><<<P.ntiles, 512, 16 * 1024>>>(p, P); }, 30)); }
      for (int tile : {256, 512, 1024}) for (int wgs : {256, 512, 768, 1024}) for (int w16 : {0, 1}) {
        Plan P; P.n = n; P.tile = tile; P.ntiles = (n + tile - 1) / tile; P.G = 1; P.inter = 0;
        for (int t = 0; t < NT; ++t) { P.o[t] = o[t]; P.ns[t] = NS[t]; }
        char nm[64]; snprintf(nm, 64, "walk T256 tile%d wgs%d w%d", tile, wgs, w16 ? 16 : 8);
        rep(nm, bi, timeit([&] { k_walk<256><<<wgs, 256>>>(p, P, w16); }, 30));
        snprintf(nm, 64, "walk T512 tile%d wgs%d w%d", tile, wgs, w16 ? 16 : 8);
        rep(nm, bi, timeit([&] { k_walk<512><<<wgs, 512>>>(p, P, w16); }, 30));
      }
      for (int wgs : {256}) {
        char nm[64]; snprintf(nm, 64, "gridstride nt0 T256 wgs%d", wgs);
        rep(nm, bi, timeit([&] { k_gs<0, 256><<<wgs, 256>>>(p, tot / 2); }, 30));
        snprintf(nm, 64, "gridstride nt1 T256 wgs%d", wgs);
        rep(nm, bi, timeit([&] { k_gs<1, 256><<<wgs, 256>>>(p, tot / 2); }, 30));
        snprintf(nm, 64, "gridstride nt1 T512 wgs%d", wgs);
        rep(nm, bi, timeit([&] { k_gs<1, 512><<<wgs, 512>>>(p, tot / 2); }, 30));
      }
      CK(hipFuncSetAttribute((const void *)k_mixed, hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024));
      for (long per : {1024L}) for (int order : {1}) for (int lds : {48}) {
        Mixed M; M.n = n; M.ntiles = (n + 511) / 512; M.fill_lo = o[9]; M.fill_n = tot - o[9]; M.per = per; M.nfill = (M.fill_n + per - 1) / per; M.order = order;
        for (int t = 0; t < NT; ++t) { M.o[t] = o[t]; M.ns[t] = NS[t]; }
        char nm[64]; snprintf(nm, 64, "mixed per%ld ord%d lds%d", per, order, lds);
        rep(nm, bi, timeit([&] { k_mixed<<<M.ntiles + M.nfill, 512, lds * 1024>>>(p, M); }, 30));
      }
      CK(hipMemsetAsync(p, 0, tot * 8, 0));
      rep("hipMemsetAsync", bi, timeit([&] { (void)hipMemsetAsync(p, 0, tot * 8, 0); }, 30));
      for (int tile : {512}) {
        for (int lds : {48}) {
          struct V { const char *nm; int G; int inter; std::vector<int> cut; };
          std::vector<V> vs = {
            {"all", 1, 0, {0, 18}},
            {"major2 (ode|fd)", 2, 0, {0, 9, 18}},
            {"inter2 (ode|fd)", 2, 1, {0, 9, 18}},
            {"major3", 3, 0, {0, 5, 9, 18}},
            {"major4", 4, 0, {0, 5, 9, 14, 18}},
            {"major6", 6, 0, {0, 3, 6, 9, 12, 15, 18}},
            {"inter6", 6, 1, {0, 3, 6, 9, 12, 15, 18}},
            {"major18", 18, 0, {}},
          };
          for (auto &v : vs) {
            Plan P; P.n = n; P.tile = tile; P.ntiles = (n + tile - 1) / tile; P.G = v.G; P.inter = v.inter;
            for (int t = 0; t < NT; ++t) { P.o[t] = o[t]; P.ns[t] = NS[t]; }
            if (v.cut.empty()) for (int g = 0; g <= 18; ++g) P.gfirst[g] = g;
            else for (size_t g = 0; g < v.cut.size(); ++g) P.gfirst[g] = v.cut[g];
            char nm[64]; snprintf(nm, 64, "%s t%d lds%d", v.nm, tile, lds);
            rep(nm, bi, timeit([&] { k_groups<<<P.ntiles * P.G, 512, lds * 1024>>>(p, P); }, 30));
          }
        }
      }
    }
  }
  return 0;
}

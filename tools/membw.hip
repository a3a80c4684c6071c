// membw.hip — calibrate what a pure-write / copy stream reaches on this GPU, in the
// access shapes the evaluation kernels use (8 B/lane and 16 B/lane, plain and
// non-temporal).  Build: hipcc --offload-arch=gfx950 -O3 -o membw tools/membw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void fill8(double* p, long n, double v) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = v; }
__global__ __launch_bounds__(256) void fill8_nt(double* p, long n, double v) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; if (i < n) __builtin_nontemporal_store(v, p + i); }
__global__ __launch_bounds__(256) void fill16(double2* p, long n2, double v) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; if (i < n2) p[i] = make_double2(v, v); }
__global__ __launch_bounds__(256) void fill16_nt(double2* p, long n2, double v) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) { __builtin_nontemporal_store(v, &p[i].x); __builtin_nontemporal_store(v, &p[i].y); } }
__global__ __launch_bounds__(256) void fill8_gs(double* p, long n, double v) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = v; }
// 8 consecutive 512-byte wave stores per wave (the shape of iem_store_rows<8>)
__global__ __launch_bounds__(256) void fill8_rows(double* p, long n, double v) {
  long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6; int lane = threadIdx.x & 63;
  long base = wave * 512 + lane;
#pragma unroll
  for (int j = 0; j < 8; ++j) if (base + j * 64 < n) p[base + j * 64] = v; }
__global__ __launch_bounds__(256) void copy8(const double* __restrict__ a, double* __restrict__ b, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; if (i < n) b[i] = a[i]; }
__global__ __launch_bounds__(256) void copy16(const double2* __restrict__ a, double2* __restrict__ b, long n2) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; if (i < n2) b[i] = a[i]; }
__global__ __launch_bounds__(256) void read8(const double* __restrict__ a, double* out, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; double v = i < n ? a[i] : 0.0; if (v == 1.2345e300) out[0] = v; }

template <class F> double timeit(F f, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) f();
  hipEventRecord(a); for (int i = 0; i < iters; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / iters;
}
int main(int argc, char** argv) {
  const long n = (argc > 1 ? atol(argv[1]) : 64L) * 1000 * 1000;  // millions of doubles (default 512 MB)
  double *p, *q; CK(hipMalloc(&p, n * 8)); CK(hipMalloc(&q, n * 8)); CK(hipMemset(p, 0, n * 8)); CK(hipMemset(q, 0, n * 8));
  long g = (n + 255) / 256, g2 = (n / 2 + 255) / 256;
  auto rep = [&](const char* name, double ms, double bytes) { printf("%-12s %8.4f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6); };
  rep("fill8", timeit([&] { fill8<<<g, 256>>>(p, n, 1.0); }, 50), n * 8.0);
  rep("fill8_nt", timeit([&] { fill8_nt<<<g, 256>>>(p, n, 1.0); }, 50), n * 8.0);
  rep("fill16", timeit([&] { fill16<<<g2, 256>>>((double2*)p, n / 2, 1.0); }, 50), n * 8.0);
  rep("fill16_nt", timeit([&] { fill16_nt<<<g2, 256>>>((double2*)p, n / 2, 1.0); }, 50), n * 8.0);
  rep("fill8_gs2048", timeit([&] { fill8_gs<<<2048, 256>>>(p, n, 1.0); }, 50), n * 8.0);
  rep("fill8_rows", timeit([&] { fill8_rows<<<(n / 8 + 255) / 256, 256>>>(p, n, 1.0); }, 50), n * 8.0);
  rep("copy8", timeit([&] { copy8<<<g, 256>>>(p, q, n); }, 50), n * 16.0);
  rep("copy16", timeit([&] { copy16<<<g2, 256>>>((double2*)p, (double2*)q, n / 2); }, 50), n * 16.0);
  rep("read8", timeit([&] { read8<<<g, 256>>>(p, q, n); }, 50), n * 8.0);
  rep("hipMemset", timeit([&] { hipMemsetAsync(p, 0, n * 8); }, 50), n * 8.0);
  // alternate two different output buffers, as jac/hess do
  rep("fill8 p,q", timeit([&] { fill8<<<g, 256>>>(p, n, 1.0); fill8<<<g, 256>>>(q, n, 2.0); }, 50), n * 16.0);
  return 0;
}

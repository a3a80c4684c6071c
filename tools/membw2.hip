// membw2.hip — store-pattern study for the COO block layout of jac_coord! (quadrotor):
// 18 template regions, region t holds n*NS[t] doubles, item-major.  No arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NT = 18;
__constant__ int NSc[NT] = {2, 5, 2, 5, 2, 4, 5, 4, 6, 3, 3, 3, 3, 3, 3, 3, 3, 3};
struct Offs { long o[NT]; };

// P1: one support per lane; per template the wave writes NS x 512 B contiguous (current kernel shape)
template <int TPB>
__global__ __launch_bounds__(TPB) void p1(double* out, Offs off, long n) {
  long q = (long)blockIdx.x * TPB + threadIdx.x; int lane = threadIdx.x & 63; long w0 = q - lane;
  if (w0 + 63 >= n) return;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ns = NSc[t];
    double* dst = out + off.o[t] + w0 * ns + lane;
    for (int j = 0; j < ns; ++j) dst[j * 64] = 1.0;
  }
}
// P2: each wave owns R consecutive 64-support tiles: per template NS x R x 512 B contiguous
template <int R>
__global__ __launch_bounds__(256) void p2(double* out, Offs off, long n) {
  long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6; int lane = threadIdx.x & 63;
  long w0 = wave * 64 * R;
  if (w0 + 64 * R - 1 >= n) return;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ns = NSc[t];
    double* dst = out + off.o[t] + w0 * ns + lane;
    for (int j = 0; j < ns * R; ++j) dst[j * 64] = 1.0;
  }
}
// P3: strided direct stores (lane writes its NS slots)
__global__ __launch_bounds__(256) void p3(double* out, Offs off, long n) {
  long q = (long)blockIdx.x * 256 + threadIdx.x; if (q >= n) return;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ns = NSc[t];
    double* dst = out + off.o[t] + q * ns;
    for (int j = 0; j < ns; ++j) dst[j] = 1.0;
  }
}
// P4: upper bound — all 62 values of a wave contiguous (not the COO layout)
__global__ __launch_bounds__(256) void p4(double* out, long n) {
  long q = (long)blockIdx.x * 256 + threadIdx.x; int lane = threadIdx.x & 63; long w0 = q - lane;
  if (w0 + 63 >= n) return;
  double* dst = out + w0 * 62 + lane;
#pragma unroll
  for (int j = 0; j < 62; ++j) dst[j * 64] = 1.0;
}
// P5: P1 with 16-byte stores (each lane writes 2 consecutive doubles; NS*64 doubles per template = NS*32 double2)
__global__ __launch_bounds__(256) void p5(double* out, Offs off, long n) {
  long q = (long)blockIdx.x * 256 + threadIdx.x; int lane = threadIdx.x & 63; long w0 = q - lane;
  if (w0 + 63 >= n) return;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ns = NSc[t];
    double2* dst = (double2*)(out + off.o[t] + w0 * ns) + lane;
    for (int j = 0; j * 64 < ns * 32; ++j) if (j * 64 + lane < ns * 32) dst[j * 64] = make_double2(1.0, 1.0);
  }
}
// P6: contiguous upper bound with 16-byte stores
__global__ __launch_bounds__(256) void p6(double* out, long n) {
  long q = (long)blockIdx.x * 256 + threadIdx.x; int lane = threadIdx.x & 63; long w0 = q - lane;
  if (w0 + 63 >= n) return;
  double2* dst = (double2*)(out + w0 * 62) + lane;
#pragma unroll
  for (int j = 0; j < 31; ++j) dst[j * 64] = make_double2(1.0, 1.0);
}
// P7: one 8-byte store per thread, 62x more threads (fill8 shape)
__global__ __launch_bounds__(256) void p7(double* out, long tot) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; if (i < tot) out[i] = 1.0; }
// P8: P1 with occupancy capped through dynamic LDS (narrower instantaneous write window)
__global__ __launch_bounds__(256) void p8(double* out, Offs off, long n) {
  extern __shared__ double dyn[];
  long q = (long)blockIdx.x * 256 + threadIdx.x; int lane = threadIdx.x & 63; long w0 = q - lane;
  if (w0 + 63 >= n) return;
  if (n < 0) dyn[threadIdx.x] = 1.0;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ns = NSc[t];
    double* dst = out + off.o[t] + w0 * ns + lane;
    for (int j = 0; j < ns; ++j) dst[j * 64] = 1.0;
  }
}
template <class F> double timeit(F f, int iters) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) f();
  (void)hipEventRecord(a); for (int i = 0; i < iters; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / iters;
}
int main() {
  const long n = 1L << 20;  // supports
  int NSh[NT] = {2, 5, 2, 5, 2, 4, 5, 4, 6, 3, 3, 3, 3, 3, 3, 3, 3, 3};
  Offs off; long tot = 0; for (int t = 0; t < NT; ++t) { off.o[t] = tot; tot += n * NSh[t]; }
  double *p, *q; CK(hipMalloc(&p, tot * 8)); CK(hipMalloc(&q, tot * 8));
  auto rep = [&](const char* name, double ms) { printf("%-14s %8.4f ms  %7.1f GB/s\n", name, ms, tot * 8.0 / ms / 1e6); };
  // alternate two output buffers like jac/hess do
  int flip = 0; auto buf = [&] { flip ^= 1; return flip ? p : q; };
  rep("p1<256>", timeit([&] { p1<256><<<n / 256, 256>>>(buf(), off, n); }, 40));
  rep("p1<64>", timeit([&] { p1<64><<<n / 64, 64>>>(buf(), off, n); }, 40));
  rep("p1<1024>", timeit([&] { p1<1024><<<n / 1024, 1024>>>(buf(), off, n); }, 40));
  rep("p2<2>", timeit([&] { p2<2><<<n / 512, 256>>>(buf(), off, n); }, 40));
  rep("p2<4>", timeit([&] { p2<4><<<n / 1024, 256>>>(buf(), off, n); }, 40));
  rep("p2<8>", timeit([&] { p2<8><<<n / 2048, 256>>>(buf(), off, n); }, 40));
  rep("p3 strided", timeit([&] { p3<<<n / 256, 256>>>(buf(), off, n); }, 40));
  rep("p5 rows 16B", timeit([&] { p5<<<n / 256, 256>>>(buf(), off, n); }, 40));
  rep("p6 contig 16B", timeit([&] { p6<<<n / 256, 256>>>(buf(), n); }, 40));
  rep("p7 fill8", timeit([&] { p7<<<(tot + 255) / 256, 256>>>(buf(), tot); }, 40));
  for (int kb : {20, 40, 80, 160}) { char nm[32]; snprintf(nm, 32, "p8 lds %dKB", kb);
    (void)hipFuncSetAttribute((const void*)p8, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
    rep(nm, timeit([&] { p8<<<n / 256, 256, kb * 1024>>>(buf(), off, n); }, 40)); }
  { Offs o2 = off; for (int t = 0; t < NT; ++t) o2.o[t] += 9;   // every region misaligned by 72 bytes
    rep("p1 misalign9", timeit([&] { p1<256><<<n / 256 - 1, 256>>>(buf(), o2, n - 256); }, 40));
    Offs o3 = off; for (int t = 0; t < NT; ++t) o3.o[t] += 8;   // 64-byte aligned only
    rep("p1 misalign8", timeit([&] { p1<256><<<n / 256 - 1, 256>>>(buf(), o3, n - 256); }, 40));
    Offs o4 = off; for (int t = 0; t < NT; ++t) o4.o[t] += 16;  // 128-byte aligned
    rep("p1 align16", timeit([&] { p1<256><<<n / 256 - 1, 256>>>(buf(), o4, n - 256); }, 40)); }
  rep("p4 contiguous", timeit([&] { p4<<<n / 256, 256>>>(buf(), n); }, 40));
  return 0;
}

// FP64 MFMA on gfx950: layout check of v_mfma_f64_16x16x4_f64 against a scalar product, and its issue rate against the
// FP64 vector FMA (the chain KKT solver's block products, csrc/iem_kkt_device.h).   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void layout(const double* A, const double* B, double* C) {   // A 16x4 row-major, B 4x16 row-major, C 16x16
  const int l = threadIdx.x;
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
template <int NACC>
__global__ void rate_mfma(double* out, int iters) {
  d4 acc[NACC];
  for (int n = 0; n < NACC; ++n) acc[n] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[n], 0, 0, 0);
  double s = 0;
  for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void rate_fma(double* out, int iters) {
  double acc[NACC];
  for (int n = 0; n < NACC; ++n) acc[n] = n;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = fma(a, acc[n], b);
  double s = 0;
  for (int n = 0; n < NACC; ++n) s += acc[n];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  std::vector<double> A(64), B(64), C(256), R(256, 0.0);
  for (int i = 0; i < 64; ++i) { A[i] = std::sin(i + 1.0); B[i] = std::cos(2.0 * i + 0.5); }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
  double *dA, *dB, *dC, *dO;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 2048); hipMalloc(&dO, 8 * 256 * 1024 * 16);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  layout<<<1, 64>>>(dA, dB, dC);
  hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost);
  double err = 0; for (int i = 0; i < 256; ++i) err = std::fmax(err, std::fabs(C[i] - R[i]));
  printf("{\"layout_max_abs_err\": %.3e", err);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, wgs = 256 * 8, thr = 256;
  auto time = [&](auto kern, const char* name, double flops_per_thread_iter) {
    kern<<<wgs, thr>>>(dO, 100); hipDeviceSynchronize();
    hipEventRecord(e0); kern<<<wgs, thr>>>(dO, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf(", \"%s_TFLOPs\": %.2f", name, flops_per_thread_iter * iters * (double)wgs * thr / (ms * 1e-3) / 1e12);
  };
  time(rate_mfma<4>, "mfma_f64_16x16x4_4acc", 4 * 2.0 * 16 * 16 * 4 / 64);
  time(rate_mfma<8>, "mfma_f64_16x16x4_8acc", 8 * 2.0 * 16 * 16 * 4 / 64);
  time(rate_fma<8>, "valu_fma_f64_8acc", 8 * 2.0);
  time(rate_fma<16>, "valu_fma_f64_16acc", 16 * 2.0);
  printf("}\n");
  return 0;
}

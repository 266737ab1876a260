// What the matrix pipe sustains on this chip when NOTHING but MFMAs is issued: the ceiling any GEMM kernel sits under.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak_probe tools/proto/mfma_peak_probe.hip && ./mfma_peak_probe
// Every wave keeps NACC independent 32x32 accumulators and issues back-to-back v_mfma_f32_32x32x2_f32 (fp32 pipe, 64
// cycles each, 4096 flop) or v_mfma_f32_32x32x16_bf16 (32 cycles, 32768 flop).  Grid = 256 CUs x WPC workgroups x 4/8
// waves.  Reported: issued TFLOP/s and the fraction of the nominal peak (157.3 TF fp32, 2516 TF bf16 at 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void __launch_bounds__(512) k_f32(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.f + blockIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 123.456f) out[0] = s;
}
template <int NACC>
__global__ void __launch_bounds__(512) k_bf16(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f + j); b[j] = (__bf16)(1.f + j); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 123.456f) out[0] = s;
}

template <class K>
double run(K kern, int threads, int wgs, int iters, double flop_per_mfma, int nacc) {
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double flop = (double)wgs * (threads / 64) * iters * nacc * flop_per_mfma;
  hipFree(out);
  return flop / (ms * 1e-3) / 1e12;
}

int main() {
  // long enough to reach the sustained (power-managed) clock: ~50-100 ms per launch
  printf("fp32  v_mfma_f32_32x32x2_f32 (nominal 157.3 TF)\n");
  for (int wpc : {1, 2}) {
    double t4 = run(k_f32<4>, 256, 256 * wpc, 400000 / wpc, 4096.0, 4);
    double t8 = run(k_f32<4>, 512, 256 * wpc, 200000 / wpc, 4096.0, 4);
    printf("  %d WG/CU: 4 waves/WG %.1f TF (%.0f %%), 8 waves/WG %.1f TF (%.0f %%)\n", wpc, t4, 100 * t4 / 157.3, t8, 100 * t8 / 157.3);
  }
  printf("bf16  v_mfma_f32_32x32x16_bf16 (nominal 2516 TF)\n");
  for (int wpc : {1, 2}) {
    double t4 = run(k_bf16<4>, 256, 256 * wpc, 800000 / wpc, 32768.0, 4);
    double t8 = run(k_bf16<4>, 512, 256 * wpc, 400000 / wpc, 32768.0, 4);
    printf("  %d WG/CU: 4 waves/WG %.1f TF (%.0f %%), 8 waves/WG %.1f TF (%.0f %%)\n", wpc, t4, 100 * t4 / 2516, t8, 100 * t8 / 2516);
  }
  return 0;
}

// What the matrix pipe sustains on this chip when NOTHING but MFMAs is issued: the ceiling any GEMM kernel sits under.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak_probe tools/proto/mfma_peak_probe.hip && ./mfma_peak_probe
// Every wave keeps NACC independent 32x32 accumulators and issues back-to-back v_mfma_f32_32x32x2_f32 (fp32 pipe, 64
// cycles each, 4096 flop) or v_mfma_f32_32x32x16_bf16 (32 cycles, 32768 flop).  Grid = 256 CUs x WPC workgroups x 4/8
// waves.  Reported: issued TFLOP/s and the fraction of the nominal peak (157.3 TF fp32, 2516 TF bf16 at 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int RANDOM>
__global__ void __launch_bounds__(512) k_f32(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  // operands: RANDOM == 0: two constants (little switching activity); 1: per-lane pseudo-random values, a different
  // pair for every accumulator (what a GEMM on real data feeds the pipe)
  float av[NACC], bv[NACC];
  unsigned h = (threadIdx.x + 1) * 2654435761u ^ (blockIdx.x * 40503u);
  for (int i = 0; i < NACC; ++i) {
    h = h * 1664525u + 1013904223u;
    av[i] = RANDOM ? (float)(int)(h >> 8) * (1.f / 8388608.f) - 1.f : threadIdx.x * 1e-3f;
    h = h * 1664525u + 1013904223u;
    bv[i] = RANDOM ? (float)(int)(h >> 8) * (1.f / 8388608.f) - 1.f : 1.f + blockIdx.x * 1e-6f;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 123.456f) out[0] = s;
}
template <int NACC, int RANDOM>
__global__ void __launch_bounds__(512) k_bf16(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 av[NACC], bv[NACC];
  unsigned h = (threadIdx.x + 1) * 2654435761u ^ (blockIdx.x * 40503u);
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 8; ++j) {
      h = h * 1664525u + 1013904223u;
      av[i][j] = (__bf16)(RANDOM ? (float)(int)(h >> 8) * (1.f / 8388608.f) - 1.f : threadIdx.x * 1e-3f + j);
      h = h * 1664525u + 1013904223u;
      bv[i][j] = (__bf16)(RANDOM ? (float)(int)(h >> 8) * (1.f / 8388608.f) - 1.f : 1.f + j);
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[i], acc[i], 0, 0, 0);
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
  printf("fp32  v_mfma_f32_32x32x2_f32 (nominal 157.3 TF), 8 waves/WG, 1 WG/CU\n");
  {
    double c = run(k_f32<4, 0>, 512, 256, 200000, 4096.0, 4);
    double r = run(k_f32<4, 1>, 512, 256, 200000, 4096.0, 4);
    printf("  constant operands %.1f TF (%.0f %%), random operands %.1f TF (%.0f %%)\n", c, 100 * c / 157.3, r, 100 * r / 157.3);
  }
  printf("bf16  v_mfma_f32_32x32x16_bf16 (nominal 2516 TF), 8 waves/WG, 1 WG/CU\n");
  {
    double c = run(k_bf16<4, 0>, 512, 256, 400000, 32768.0, 4);
    double r = run(k_bf16<4, 1>, 512, 256, 400000, 32768.0, 4);
    printf("  constant operands %.1f TF (%.0f %%), random operands %.1f TF (%.0f %%)\n", c, 100 * c / 2516, r, 100 * r / 2516);
  }
  return 0;
}

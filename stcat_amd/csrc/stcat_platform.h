// Platform layer for the STCAT kernels.  The product build is hipcc for gfx950
// only.  -DSTCAT_EMU swaps in tests/emu/hip_emu.h so the SAME kernel bodies can
// be exercised for index logic on a host without a GPU (test infrastructure;
// never shipped, never on the product path).
#pragma once

#ifdef STCAT_EMU
#include "hip_emu.h"
#define STCAT_MFMA_32x32x2(a, b, c) emu_mfma_f32_32x32x2f32((a), (b), (c))
#define STCAT_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  emu::launch((grid), (block), (shmem), [&]() { kernel(__VA_ARGS__); })
#define STCAT_DYN_SHARED(type, name) type* name = reinterpret_cast<type*>(emu::t_dynshared)
#define STCAT_UNROLL
#else
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define STCAT_MFMA_32x32x2(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define STCAT_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kernel, (grid), (block), (shmem), (stream), __VA_ARGS__)
#define STCAT_DYN_SHARED(type, name) \
  extern __shared__ __attribute__((aligned(16))) char stcat_dyn_smem_[]; \
  type* name = reinterpret_cast<type*>(stcat_dyn_smem_)
#define STCAT_UNROLL _Pragma("unroll")
#endif

#define STCAT_WAVE 64
#define STCAT_NEG_INF (-__builtin_inff())

static __device__ __forceinline__ float4 stcat_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
static __device__ __forceinline__ void stcat_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// wave-wide reductions over all 64 lanes
static __device__ __forceinline__ float stcat_wave_sum(float v) {
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
static __device__ __forceinline__ float stcat_wave_max(float v) {
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}

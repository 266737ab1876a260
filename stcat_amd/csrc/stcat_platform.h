// Platform layer for the STCAT kernels.  The product build is hipcc for gfx950
// only.  -DSTCAT_EMU swaps in tests/emu/hip_emu.h so the SAME kernel bodies can
// be exercised for index logic on a host without a GPU (test infrastructure;
// never shipped, never on the product path).
#pragma once

#ifdef STCAT_EMU
#include "hip_emu.h"
#define STCAT_MFMA_32x32x2(a, b, c) emu_mfma_f32_32x32x2f32((a), (b), (c))
#define STCAT_MFMA_BF16_32x32x16(a, b, c) emu_mfma_f32_32x32x16_bf16((a), (b), (c))
#define STCAT_MFMA_F16_32x32x16(a, b, c) emu_mfma_f32_32x32x16_f16((a), (b), (c))
#define STCAT_MFMA_BF16_16x16x32(a, b, c) emu_mfma_f32_16x16x32_bf16((a), (b), (c))
#define STCAT_SCHED_GROUP(mask, n) ((void)0)
#define STCAT_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  emu::launch((grid), (block), (shmem), [&]() { kernel(__VA_ARGS__); })
#define STCAT_DYN_SHARED(type, name) type* name = reinterpret_cast<type*>(emu::t_dynshared)
#define STCAT_UNROLL
#else
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define STCAT_MFMA_32x32x2(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define STCAT_MFMA_BF16_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define STCAT_MFMA_F16_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define STCAT_MFMA_BF16_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
// instruction-scheduler request: next group = n instructions of class mask (0x8 MFMA, 0x20 VMEM read, 0x100 DS read, ...)
#define STCAT_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
#define STCAT_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kernel, (grid), (block), (shmem), (stream), __VA_ARGS__)
#define STCAT_DYN_SHARED(type, name) \
  extern __shared__ __attribute__((aligned(16))) char stcat_dyn_smem_[]; \
  type* name = reinterpret_cast<type*>(stcat_dyn_smem_)
#define STCAT_UNROLL _Pragma("unroll")
#endif

// ---- bounds-checked 16-byte loads through a buffer descriptor: an offset at or beyond `bytes` returns
// zeros in hardware (raw buffer load, stride 0), which is how padding / stride-lattice / tail rows of the
// implicit-GEMM gathers are zero-filled without exec-mask branches.  Offsets are 32-bit byte offsets.
#define STCAT_BUF_OOB 0x80000000u
#ifdef STCAT_EMU
struct stcat_buf_t { const char* base; unsigned bytes; };
static inline stcat_buf_t stcat_make_buf(const void* p, unsigned bytes) { return stcat_buf_t{(const char*)p, bytes}; }
static inline float4 stcat_buf_ld4(stcat_buf_t b, unsigned voff, unsigned soff) {
  const unsigned long off = (unsigned long)voff + soff;
  if (voff >= STCAT_BUF_OOB || off + 16 > b.bytes) return make_float4(0.f, 0.f, 0.f, 0.f);
  return *reinterpret_cast<const float4*>(b.base + off);
}
#else
typedef __amdgpu_buffer_rsrc_t stcat_buf_t;
static __device__ __forceinline__ stcat_buf_t stcat_make_buf(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
static __device__ __forceinline__ float4 stcat_buf_ld4(stcat_buf_t b, unsigned voff, unsigned soff) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b, (int)voff, (int)soff, 0);
  return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
#endif

// ---- LDS-DMA staging and hand-placed synchronisation of the plane-format GEMM (igemm_pl.h) ----------------------
// stcat_glds16: every lane moves 16 bytes HBM -> LDS without touching a VGPR: source = descriptor base + voff + soff
// (bounds-checked: out of range = ZEROS written, measured on gfx950: profiles/r02_probe_lds_dma_tr.log), destination =
// wave-uniform LDS address + lane * 16.  The transfer is counted by vmcnt; nothing orders a later ds_read behind it
// except the issuing wave's s_waitcnt vmcnt + a barrier (MI355X_MICROARCH.md §Two waves per SIMD, item 7).
// stcat_lds_tr4: ds_read_b64_tr_b16 — the 16 lanes of a group read a [4][16] block of 16-bit elements (lane p supplies
// the address of row p >> 2, columns 4 (p & 3) .. +3) and lane i receives column i: the transpose a k-major operand
// needs to become an MFMA fragment (same measurement).
#ifdef STCAT_EMU
static inline void stcat_glds16(stcat_buf_t b, char* lds_wave_base, unsigned voff, unsigned soff) {
  const unsigned long off = (unsigned long)voff + soff;
  char* dst = lds_wave_base + emu::lane() * 16;
  if (voff >= STCAT_BUF_OOB || off + 16 > b.bytes) memset(dst, 0, 16);
  else memcpy(dst, b.base + off, 16);
}
static inline bf16x4 stcat_lds_tr4(const __bf16* addr) {
  emu::WaveState& w = emu::wave();
  const int l = emu::lane();
  w.xp[l] = addr;
  emu::wave_sync();
  bf16x4 r;
  const int g = l & ~15, i = l & 15;
  for (int j = 0; j < 4; ++j) r[j] = static_cast<const __bf16*>(w.xp[g + j * 4 + (i >> 2)])[i & 3];
  emu::wave_sync();
  return r;
}
#define STCAT_WAIT_VM0_LGKM0() ((void)0)
#define STCAT_WAIT_VM0() ((void)0)
#define STCAT_WAIT_VM(n) ((void)0)
#define STCAT_S_BARRIER() __syncthreads()
#define STCAT_SCHED_FENCE() ((void)0)
#define STCAT_WAVE_LDS_FENCE() emu::wave_sync()
#define STCAT_READFIRSTLANE(x) (x)
#else
static __device__ __forceinline__ void stcat_glds16(stcat_buf_t b, char* lds_wave_base, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(b, (void __attribute__((address_space(3)))*)lds_wave_base, 16, (int)voff,
                                           (int)soff, 0, 0);
}
static __device__ __forceinline__ bf16x4 stcat_lds_tr4(const __bf16* addr) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)addr);
  return __builtin_bit_cast(bf16x4, v);
}
#define STCAT_WAIT_VM0_LGKM0() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define STCAT_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define STCAT_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define STCAT_S_BARRIER() __builtin_amdgcn_s_barrier()
#define STCAT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define STCAT_WAVE_LDS_FENCE() ((void)0)  /* DS operations of one wave execute in order */
#define STCAT_READFIRSTLANE(x) __builtin_amdgcn_readfirstlane(x)
#endif

// streaming (non-temporal) access for the epilogue's one-touch operands — the output planes (nobody re-reads 308 MB of
// them before they have left the caches) and the residual planes: with the second prefetch set of the three-plane tile,
// 256 -> 1024 forward 0.276 -> 0.244 ms, 1024 -> 256 data gradient 0.299 -> 0.288 ms (step-like operands, same box)
#if !defined(STCAT_NO_NT) && !defined(STCAT_EMU)
#define STCAT_STORE_STREAM(ptr, val) __builtin_nontemporal_store((val), (ptr))
#define STCAT_LOAD_STREAM(ptr) __builtin_nontemporal_load(ptr)
#else
#define STCAT_STORE_STREAM(ptr, val) (*(ptr) = (val))
#define STCAT_LOAD_STREAM(ptr) (*(ptr))
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

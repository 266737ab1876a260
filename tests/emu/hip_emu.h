// Host-side SIMT emulator for the STCAT HIP kernels — TEST INFRASTRUCTURE ONLY.
//
// The build container has no GPU, so the kernel sources (stcat_amd/csrc/*.h)
// are written against the small macro layer of stcat_amd/csrc/stcat_platform.h.
// With -DSTCAT_EMU that layer maps onto this header: every GPU thread of a
// workgroup is a user-level fiber (hand-rolled x86-64 context switch), a
// workgroup runs on one OS thread, workgroups are spread over the host cores.
// __syncthreads() and the wave-level exchanges behind MFMA / shuffles are
// cooperative barriers between fibers.  It checks INDEX LOGIC (tiling,
// fragment layouts, masks, edges, chain rule wiring) on small problems; it
// says nothing about speed.  Numerics: the emulated mfma_f32_32x32x2f32 is a
// k-ordered fmaf chain like the hardware (cdna_hip_programming.md §3).
#pragma once
#include <sys/mman.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline const char* hipGetErrorString(hipError_t) { return "emulator"; }
static inline hipError_t hipMemsetAsync(void* p, int value, size_t bytes, hipStream_t) {
  memset(p, value, bytes);
  return hipSuccess;
}

typedef void* hipEvent_t;

extern "C" void stcat_emu_switch(void** from_sp, void* to_sp);

namespace emu {
constexpr size_t kStack = 256 * 1024;

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = false;
};
struct WaveState {
  int gen = 0, arrived = 0;
  float xa[64], xb[64];
  int xi[64];
  float xa8[64][8], xb8[64][8];
  const void* xp[64];
};
struct BlockCtx {
  std::vector<Fiber> fibers;
  std::vector<WaveState> waves;
  void* sched_sp = nullptr;
  int cur = 0, nthreads = 0;
  int gen = 0, arrived = 0;
  std::function<void()>* body = nullptr;
  dim3 block_dim;
};
extern thread_local BlockCtx* t_ctx;
extern thread_local dim3 t_threadIdx, t_blockIdx;
extern thread_local dim3 t_blockDim, t_gridDim;
extern thread_local char* t_dynshared;

inline void set_thread_idx(BlockCtx* c, int t) {
  t_threadIdx = dim3(t % c->block_dim.x, (t / c->block_dim.x) % c->block_dim.y,
                     t / (c->block_dim.x * c->block_dim.y));
}
inline void yield_to_sched() {
  BlockCtx* c = t_ctx;
  stcat_emu_switch(&c->fibers[c->cur].sp, c->sched_sp);
  set_thread_idx(c, c->cur);  // resumed
}
inline void block_barrier() {
  BlockCtx* c = t_ctx;
  const int g = c->gen;
  if (++c->arrived == c->nthreads) {
    c->arrived = 0;
    c->gen++;
    return;
  }
  while (c->gen == g) yield_to_sched();
}
inline void wave_sync() {
  BlockCtx* c = t_ctx;
  WaveState& w = c->waves[c->cur >> 6];
  const int g = w.gen;
  if (++w.arrived == 64) {
    w.arrived = 0;
    w.gen++;
    return;
  }
  while (w.gen == g) yield_to_sched();
}
inline WaveState& wave() { return t_ctx->waves[t_ctx->cur >> 6]; }
inline int lane() { return t_ctx->cur & 63; }

void fiber_entry();

inline void run_block(BlockCtx& c) {
  for (int t = 0; t < c.nthreads; ++t) {
    Fiber& f = c.fibers[t];
    f.done = false;
    // initial frame: six callee-saved slots + return address = fiber_entry; entry sees rsp % 16 == 8
    uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStack) & ~uintptr_t(15);
    void** sp = reinterpret_cast<void**>(top - 8);
    *--sp = reinterpret_cast<void*>(&fiber_entry);
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    f.sp = sp;
  }
  for (auto& w : c.waves) { w.gen = 0; w.arrived = 0; }
  c.gen = 0;
  c.arrived = 0;
  int remaining = c.nthreads;
  while (remaining > 0) {
    for (int t = 0; t < c.nthreads; ++t) {
      Fiber& f = c.fibers[t];
      if (f.done) continue;
      c.cur = t;
      set_thread_idx(&c, t);
      stcat_emu_switch(&c.sched_sp, f.sp);
      if (f.done) --remaining;
    }
  }
}

template <class F>
void launch(dim3 grid, dim3 block, size_t dyn_shared, F body_fn) {
  const int nthreads = block.x * block.y * block.z;
  if (nthreads % 64 != 0) { fprintf(stderr, "emu: block size %d not a multiple of 64\n", nthreads); abort(); }
  std::function<void()> body = body_fn;
  const long nblocks = (long)grid.x * grid.y * grid.z;
  std::atomic<long> next{0};
  unsigned hw = std::thread::hardware_concurrency();
  if (const char* e = getenv("STCAT_EMU_THREADS")) hw = atoi(e);
  const int nworkers = (int)std::max<long>(1, std::min<long>(nblocks, hw ? hw : 4));
  auto worker = [&]() {
    BlockCtx c;
    c.nthreads = nthreads;
    c.block_dim = block;
    c.body = &body;
    c.fibers.resize(nthreads);
    c.waves.resize(nthreads / 64);
    char* arena = static_cast<char*>(mmap(nullptr, kStack * nthreads, PROT_READ | PROT_WRITE,
                                          MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (arena == MAP_FAILED) { perror("emu mmap"); abort(); }
    for (int t = 0; t < nthreads; ++t) c.fibers[t].stack = arena + kStack * t;
    std::vector<char> dyn(dyn_shared + 64);
    char* dynp = dyn.data();
    dynp += (16 - (reinterpret_cast<uintptr_t>(dynp) & 15)) & 15;
    t_ctx = &c;
    t_dynshared = dynp;
    t_blockDim = block;
    t_gridDim = grid;
    for (;;) {
      const long b = next.fetch_add(1);
      if (b >= nblocks) break;
      t_blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y)));
      run_block(c);
    }
    t_ctx = nullptr;
    munmap(arena, kStack * nthreads);
  };
  if (nworkers == 1) {
    std::thread th(worker);  // own OS thread: thread_local LDS stays private to the launch
    th.join();
  } else {
    std::vector<std::thread> pool;
    for (int i = 0; i < nworkers; ++i) pool.emplace_back(worker);
    for (auto& th : pool) th.join();
  }
}
}  // namespace emu

#define threadIdx (emu::t_threadIdx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::t_blockDim)
#define gridDim (emu::t_gridDim)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __restrict__

static inline void __syncthreads() { emu::block_barrier(); }
using std::max;
using std::min;

static inline float emu_shfl(float v, int src) {
  emu::wave().xa[emu::lane()] = v;
  emu::wave_sync();
  float r = emu::wave().xa[src & 63];
  emu::wave_sync();
  return r;
}
static inline float __shfl_xor(float v, int mask) { return emu_shfl(v, emu::lane() ^ mask); }
static inline float __shfl(float v, int src) { return emu_shfl(v, src); }
static inline int __shfl_xor(int v, int mask) {
  emu::wave().xi[emu::lane()] = v;
  emu::wave_sync();
  int r = emu::wave().xi[(emu::lane() ^ mask) & 63];
  emu::wave_sync();
  return r;
}

// D = A(32x2) * B(2x32) + C, lane l supplies A[l&31][l>>5] and B[l>>5][l&31];
// lane l reg r holds D[(r&3)+8*(r>>2)+4*(l>>5)][l&31].
static inline f32x16 emu_mfma_f32_32x32x2f32(float a, float b, f32x16 c) {
  emu::WaveState& w = emu::wave();
  const int l = emu::lane();
  w.xa[l] = a;
  w.xb[l] = b;
  emu::wave_sync();
  const int j = l & 31, hi = l >> 5;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    acc = fmaf(w.xa[i], w.xb[j], acc);
    acc = fmaf(w.xa[i + 32], w.xb[j + 32], acc);
    c[r] = acc;
  }
  emu::wave_sync();
  return c;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_32x32x16_bf16: lane l supplies A[l&31][8*(l>>5)+j] and B[8*(l>>5)+j][l&31], j < 8.
static inline f32x16 emu_mfma_f32_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  emu::WaveState& w = emu::wave();
  float (*sa)[8] = w.xa8;
  float (*sb)[8] = w.xb8;
  const int l = emu::lane();
  for (int j = 0; j < 8; ++j) { sa[l][j] = (float)a[j]; sb[l][j] = (float)b[j]; }
  emu::wave_sync();
  const int col = l & 31, hi = l >> 5;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    for (int h = 0; h < 2; ++h)
      for (int j = 0; j < 8; ++j) acc = fmaf(sa[i + 32 * h][j], sb[col + 32 * h][j], acc);
    c[r] = acc;
  }
  emu::wave_sync();
  return c;
}

// v_mfma_f32_16x16x32_bf16: lane l supplies A[l&15][8*(l>>4)+j] and B[8*(l>>4)+j][l&15], j < 8; D: col = l&15,
// row = 4*(l>>4) + reg (cdna_hip_programming.md section 3)
static inline f32x4 emu_mfma_f32_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
  emu::WaveState& w = emu::wave();
  float (*sa)[8] = w.xa8;
  float (*sb)[8] = w.xb8;
  const int l = emu::lane();
  for (int j = 0; j < 8; ++j) { sa[l][j] = (float)a[j]; sb[l][j] = (float)b[j]; }
  emu::wave_sync();
  const int col = l & 15, kg = l >> 4;
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * kg + r;
    float acc = c[r];
    for (int h = 0; h < 4; ++h)
      for (int j = 0; j < 8; ++j) acc = fmaf(sa[i + 16 * h][j], sb[col + 16 * h][j], acc);
    c[r] = acc;
  }
  emu::wave_sync();
  return c;
}

// v_mfma_f32_32x32x16_f16: the same lane mapping on fp16 operands
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
static inline f32x16 emu_mfma_f32_32x32x16_f16(f16x8 a, f16x8 b, f32x16 c) {
  emu::WaveState& w = emu::wave();
  float (*sa)[8] = w.xa8;
  float (*sb)[8] = w.xb8;
  const int l = emu::lane();
  for (int j = 0; j < 8; ++j) { sa[l][j] = (float)a[j]; sb[l][j] = (float)b[j]; }
  emu::wave_sync();
  const int col = l & 31, hi = l >> 5;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    for (int h = 0; h < 2; ++h)
      for (int j = 0; j < 8; ++j) acc = fmaf(sa[i + 32 * h][j], sb[col + 32 * h][j], acc);
    c[r] = acc;
  }
  emu::wave_sync();
  return c;
}

static inline float atomicAdd(float* p, float v) {
  uint32_t* ip = reinterpret_cast<uint32_t*>(p);
  uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED);
  for (;;) {
    float f;
    memcpy(&f, &old, 4);
    f += v;
    uint32_t nv;
    memcpy(&nv, &f, 4);
    if (__atomic_compare_exchange_n(ip, &old, nv, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
      memcpy(&f, &old, 4);
      return f;
    }
  }
}
#define __expf(x) expf(x)
#define __logf(x) logf(x)

#define EMU_DEFINE_GLOBALS                                                        \
  namespace emu {                                                                 \
  thread_local BlockCtx* t_ctx = nullptr;                                         \
  thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;               \
  thread_local char* t_dynshared = nullptr;                                       \
  void fiber_entry() {                                                            \
    BlockCtx* c = t_ctx;                                                          \
    (*c->body)();                                                                 \
    c = t_ctx;                                                                    \
    c->fibers[c->cur].done = true;                                                \
    for (;;) stcat_emu_switch(&c->fibers[c->cur].sp, c->sched_sp);                \
  }                                                                               \
  }                                                                               \
  asm(".text\n.globl stcat_emu_switch\n.type stcat_emu_switch,@function\n"        \
      "stcat_emu_switch:\n"                                                       \
      "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n" \
      "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"                                  \
      "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n");

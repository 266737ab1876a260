// PROTOTYPE (round-2 groundwork, not part of the product library): split-bf16 GEMM whose operands are ALREADY
// split into bf16 hi/lo planes in HBM and travel HBM -> LDS by LDS-DMA (global_load_lds_dwordx4), so the K loop has
// no VGPR staging, no conversion VALU and no ds_write:  C[M,N] = sum_k (Ah+Al)[m,k] * (Bh+Bl)[n,k]   (3 products).
// 128x128 tile, 4 waves, K step 32, two LDS stages (64 KB -> 2 workgroups per CU), XOR-swizzled 64-byte rows
// (swizzle applied to the SOURCE chunk and to the fragment READ; the DMA destination stays lane-linear).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/proto/gemm_planes.hip -o tools/proto/gemm_planes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct P {
  const __bf16* Ah; const __bf16* Al; const __bf16* Bh; const __bf16* Bl;  // [M][K], [N][K]
  float* C;
  int M, N, K;
};

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  // bijective for any grid size: XCD x (= bid % 8) owns a contiguous run of q or q+1 virtual tiles
  const int q = nblk / 8, r = nblk % 8, x = bid % 8, i = bid / 8;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

template <int STAGES, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN) gemm_planes_kernel(P p) {
  constexpr int BK = 32, NW = WM * WN, TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int PLANE_A = BM * BK * 2, PLANE_B = BN * BK * 2;   // rows x 64 B
  constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_B;              // Ah, Al, Bh, Bl
  constexpr int RQA = BM / 16 / NW, RQB = BN / 16 / NW;         // 16-row DMA instructions per wave and plane
  __shared__ __attribute__((aligned(1024))) char smem[STAGES * STAGE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int num_n = p.N / BN;
  const int v = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (v / num_n) * BM, n0 = (v % num_n) * BN;
  const int nk = p.K / BK;

  // DMA source pointers of this lane: wave w stages 16-row slabs q = w*RQ + i of every plane
  const char* srcA[2][RQA];
  const char* srcB[2][RQB];
#pragma unroll
  for (int i = 0; i < RQA; ++i) {
    const int q = wave * RQA + i, r = q * 16 + (lane >> 2), cp = lane & 3, c = cp ^ ((r >> 2) & 3);
    srcA[0][i] = (const char*)(p.Ah + (long)(m0 + r) * p.K) + c * 16;
    srcA[1][i] = (const char*)(p.Al + (long)(m0 + r) * p.K) + c * 16;
  }
#pragma unroll
  for (int i = 0; i < RQB; ++i) {
    const int q = wave * RQB + i, r = q * 16 + (lane >> 2), cp = lane & 3, c = cp ^ ((r >> 2) & 3);
    srcB[0][i] = (const char*)(p.Bh + (long)(n0 + r) * p.K) + c * 16;
    srcB[1][i] = (const char*)(p.Bl + (long)(n0 + r) * p.K) + c * 16;
  }
  auto stage_load = [&](int kt, int st) {
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
      for (int i = 0; i < RQA; ++i)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(srcA[pl][i] + (long)kt * BK * 2),
                                         (void __attribute__((address_space(3)))*)(smem + st * STAGE + pl * PLANE_A + (wave * RQA + i) * 1024),
                                         16, 0, 0);
#pragma unroll
      for (int i = 0; i < RQB; ++i)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(srcB[pl][i] + (long)kt * BK * 2),
                                         (void __attribute__((address_space(3)))*)(smem + st * STAGE + 2 * PLANE_A + pl * PLANE_B + (wave * RQB + i) * 1024),
                                         16, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment byte offsets inside a plane (swizzled), for k-step ks: chunk c = ks*2 + hi
  int aoff[TM][2], boff[TN][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int c = ks * 2 + hi;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int ra = wm * TM * 32 + tm * 32 + l31;
      aoff[tm][ks] = ra * 64 + ((c ^ ((ra >> 2) & 3)) * 16);
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int rb = wn * TN * 32 + tn * 32 + l31;
      boff[tn][ks] = rb * 64 + ((c ^ ((rb >> 2) & 3)) * 16);
    }
  }

#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) stage_load(s, s);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed (everything older than the newest STAGES-2 tiles), and every wave is done with the stage
    // that the next DMA overwrites
    if (STAGES == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * RQA + 2 * RQB) : "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + STAGES - 1 < nk) stage_load(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
    const char* sb = smem + (kt % STAGES) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        ah[tm] = *(const bf16x8*)(sb + 0 * PLANE_A + aoff[tm][ks]);
        al[tm] = *(const bf16x8*)(sb + 1 * PLANE_A + aoff[tm][ks]);
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        bh[tn] = *(const bf16x8*)(sb + 2 * PLANE_A + 0 * PLANE_B + boff[tn][ks]);
        bl[tn] = *(const bf16x8*)(sb + 2 * PLANE_A + 1 * PLANE_B + boff[tn][ks]);
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
    }
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, col = n0 + wn * TN * 32 + tn * 32 + l31;
        p.C[(long)row * p.N + col] = acc[tm][tn][r];
      }
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 50176, N = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 2304;
  const int stages = argc > 4 ? atoi(argv[4]) : 2;
  std::vector<unsigned short> ah((size_t)M * K), al((size_t)M * K), bh((size_t)N * K), bl((size_t)N * K);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
  for (size_t i = 0; i < ah.size(); ++i) { float x = rnd(); ah[i] = f2bf(x); al[i] = f2bf(x - bf2f(ah[i])); }
  for (size_t i = 0; i < bh.size(); ++i) { float x = rnd(); bh[i] = f2bf(x); bl[i] = f2bf(x - bf2f(bh[i])); }
  P p; p.M = M; p.N = N; p.K = K;
  void *dah, *dal, *dbh, *dbl; float* dc;
  CHECK(hipMalloc(&dah, ah.size() * 2)); CHECK(hipMalloc(&dal, al.size() * 2));
  CHECK(hipMalloc(&dbh, bh.size() * 2)); CHECK(hipMalloc(&dbl, bl.size() * 2)); CHECK(hipMalloc(&dc, (size_t)M * N * 4));
  CHECK(hipMemcpy(dah, ah.data(), ah.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dal, al.data(), al.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dbh, bh.data(), bh.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dbl, bl.data(), bl.size() * 2, hipMemcpyHostToDevice));
  p.Ah = (const __bf16*)dah; p.Al = (const __bf16*)dal; p.Bh = (const __bf16*)dbh; p.Bl = (const __bf16*)dbl; p.C = dc;
  const int variant = argc > 5 ? atoi(argv[5]) : 0;  // 0: 128x128/4 waves, 1: 256x128/8 waves, 2: 256x256/8 waves
  auto launch = [&]() {
    if (variant == 0) {
      if (stages == 3) hipLaunchKernelGGL((gemm_planes_kernel<3, 128, 128, 2, 2>), dim3((M / 128) * (N / 128)), dim3(256), 0, 0, p);
      else hipLaunchKernelGGL((gemm_planes_kernel<2, 128, 128, 2, 2>), dim3((M / 128) * (N / 128)), dim3(256), 0, 0, p);
    } else if (variant == 1) {
      if (stages == 3) hipLaunchKernelGGL((gemm_planes_kernel<3, 256, 128, 4, 2>), dim3((M / 256) * (N / 128)), dim3(512), 0, 0, p);
      else hipLaunchKernelGGL((gemm_planes_kernel<2, 256, 128, 4, 2>), dim3((M / 256) * (N / 128)), dim3(512), 0, 0, p);
    } else {
      hipLaunchKernelGGL((gemm_planes_kernel<2, 256, 256, 2, 4>), dim3((M / 256) * (N / 256)), dim3(512), 0, 0, p);
    }
  };
  launch(); CHECK(hipDeviceSynchronize());
  std::vector<float> c((size_t)M * N);
  CHECK(hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0;
  for (int trial = 0; trial < 64; ++trial) {
    const int m = (trial * 7919) % M, n = (trial * 104729) % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) {
      const double a0 = bf2f(ah[(size_t)m * K + k]), a1 = bf2f(al[(size_t)m * K + k]);
      const double b0 = bf2f(bh[(size_t)n * K + k]), b1 = bf2f(bl[(size_t)n * K + k]);
      ref += a0 * b0 + a0 * b1 + a1 * b0;
    }
    maxerr = fmax(maxerr, fabs(ref - c[(size_t)m * N + n]));
  }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < 20; ++i) launch();
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
  printf("variant %d M=%d N=%d K=%d stages=%d: %.3f ms  %.1f TF algorithmic (x3 = %.0f TF issued)  max |err| vs fp64 of the same 3 products: %.3e\n",
         variant, M, N, K, stages, ms, 2.0 * M * N * K / ms / 1e9, 6.0 * M * N * K / ms / 1e9, maxerr);
  return 0;
}

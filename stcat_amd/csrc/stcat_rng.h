// Counter-based dropout decisions (train mode of modal_encoder.py:237-240, query_decoder.py:344/431-436/612/653-658,
// attention.py:381, net_utils.py:24-25).
//
// A site draws one decision per element from (seed, counter): keep iff rand32(seed, offset + index) >= thresh,
// thresh = p * 2^32, kept values scaled by 1/(1-p).  Nothing is stored: the backward kernels regenerate the
// same mask from the same (seed, offset), so dropout adds no HBM traffic.  The generator is the splitmix64
// finaliser over a Weyl sequence (the host-side twin is stcat_amd/ops.py:dropout_keep_mask; the tests compare
// the two bit for bit).  torch's Philox stream is NOT reproduced: train-mode parity is "same arithmetic given
// the same mask", checked by replaying this mask in the fp32 reference.
#pragma once
#include "stcat_platform.h"

struct DropParams {
  unsigned thresh;            // 0 = dropout disabled
  float scale;                // 1 / (1 - p)
  unsigned long long seed;
  unsigned long long offset;  // first counter of this site
  const unsigned long long* base;  // optional DEVICE word added to offset when the kernel starts: the per-step
                                   // advance of the stream lives in HBM, so a captured hipGraph that replays the
                                   // same launch arguments still draws fresh masks every step
};

// fold the device-resident base into the offset (call once at kernel entry)
static __device__ __forceinline__ DropParams stcat_drop_resolve(DropParams d) {
  if (d.thresh != 0u && d.base != nullptr) d.offset += *d.base;
  d.base = nullptr;
  return d;
}

static __device__ __forceinline__ unsigned stcat_rand32(unsigned long long seed, unsigned long long ctr) {
  unsigned long long z = seed + (ctr + 1ull) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (unsigned)(z >> 32);
}

// multiplier of element `idx` of the site: 0 or 1/(1-p)   (1 when disabled)
static __device__ __forceinline__ float stcat_drop_mul(const DropParams& d, unsigned long long idx) {
  if (d.thresh == 0u) return 1.f;
  return stcat_rand32(d.seed, d.offset + idx) >= d.thresh ? d.scale : 0.f;
}

static inline DropParams stcat_make_drop(float p, long seed, long offset, const long* base) {
  DropParams d;
  d.thresh = 0u; d.scale = 1.f; d.seed = (unsigned long long)seed; d.offset = (unsigned long long)offset;
  d.base = reinterpret_cast<const unsigned long long*>(base);
  if (p > 0.f) {
    const double t = (double)p * 4294967296.0;
    d.thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    if (d.thresh == 0u) d.thresh = 1u;
    d.scale = (float)(1.0 / (1.0 - (double)p));
  }
  return d;
}

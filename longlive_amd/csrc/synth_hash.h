// The counter hash behind longlive_amd/synth.py (splitmix64 finaliser over (seed, name, index)), as plain integer arithmetic that
// compiles for the device (ll_synth_hash, elementwise.hip) and for the host (tests/test_synth_hash.py builds it with g++ and
// compares it with synth.hash_normal / hash_uniform bit for bit).  uint64 wrap-around = torch's int64 wrap-around; `>>` on
// uint64 = synth._lsr.
#pragma once
#include <stdint.h>
#ifdef __HIPCC__
#define LL_HD __host__ __device__ __forceinline__
#else
#define LL_HD inline
#endif

LL_HD uint64_t ll_synth_mix(uint64_t x) {
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// float32 uniform in [0, 1) with 24 exact bits (synth.hash_uniform)
LL_HD float ll_synth_uniform(uint64_t idx, uint64_t stream) {
  const uint64_t h = ll_synth_mix(idx * 0x9E3779B97F4A7C15ull + stream);
  return (float)(h >> 40) * (1.0f / 16777216.0f);
}
// float32 approx. N(0, 1): Irwin-Hall sum of 12 uniforms of 16 bits, exact integer sum (synth.hash_normal)
LL_HD float ll_synth_normal(uint64_t idx, uint64_t stream) {
  int64_t acc = 0;
  for (int r = 0; r < 3; ++r) {
    const uint64_t h = ll_synth_mix((idx * 3 + (uint64_t)r) * 0x9E3779B97F4A7C15ull + stream);
    for (int k = 0; k < 4; ++k) acc += (int64_t)((h >> (16 * k)) & 0xFFFF);
  }
  return (float)(acc - 393210) * (1.0f / 65536.0f);
}

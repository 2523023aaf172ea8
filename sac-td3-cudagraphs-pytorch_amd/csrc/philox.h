// Philox4x32-10 counter-based RNG (Salmon et al., "Parallel Random Numbers: As Easy as 1, 2, 3", SC'11;
// constants and round structure as in Random123's philox.h).  Stateless: every draw of the engine is
// a pure function of (seed, stream counter, site, element), so any lane can produce its own numbers
// and a captured hipGraph advances only a device-side counter word.
//
// Stands in for the reference's torch default-generator draws (Normal.rsample agents/nets.py:225,
// normal_/randn_like agents/agent.py:197, agents/nets.py:158) and torchrl's RandomSampler
// (orchestrator.py:338): same distributions, a different (documented) stream.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SACTD3_HD __host__ __device__ __forceinline__
#else
#define SACTD3_HD inline
#endif

struct Philox4 { uint32_t v[4]; };

SACTD3_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  Philox4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// stream ids (third counter word)
#define SACTD3_STREAM_INDEX   0x100u   // replay index draws
#define SACTD3_STREAM_FILL    0x200u   // synthetic buffer fill
#define SACTD3_STREAM_NOISE   0x000u   // + site code

// uniform index in [0, len): Lemire multiply-shift on one 32-bit word (bias <= len / 2^32)
SACTD3_HD uint32_t philox_index(uint64_t seed, uint32_t ctr, uint32_t b, uint32_t len) {
  const Philox4 r = philox4x32_10(ctr, 0u, SACTD3_STREAM_INDEX, b >> 2, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t k = b & 3u;   // selects: a dynamically indexed r.v[] would be placed in scratch memory on the GPU
  const uint32_t w = k == 0 ? r.v[0] : (k == 1 ? r.v[1] : (k == 2 ? r.v[2] : r.v[3]));
  return (uint32_t)(((uint64_t)w * (uint64_t)len) >> 32);
}

SACTD3_HD float philox_u01(uint32_t x) {  // (0,1), 24 random bits
  return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

// cloth_rng.hpp -- numpy's legacy RandomState stream (MT19937) on the device.
//
// ClothEnv.reset draws its scripted reset pulls from the env's np_random (a numpy RandomState seeded through gym's
// seeding, cloth_env.py:332-341) in an order that depends on the cloth state (tier 1 draws a third pull only if the coverage
// is still >= 0.90, cloth_env.py:866). For episodes that reset INSIDE a kernel launch the draws therefore have to happen on
// the device, bit for bit as numpy makes them:
//   RandomState.rand() / uniform(low, high)   low + (high - low) * d,  d = (a * 67108864 + b) / 2^53 with a = next32 >> 5,
//                                             b = next32 >> 6                      (legacy rk_double / mt19937_next_double)
//   RandomState.randint(n)                    masked rejection on 32-bit words: w & mask until <= n - 1, mask = 2^k - 1 >= n - 1
//                                             (buffered_bounded_masked_uint32; range below 2^32)
//   MT19937 itself                            the reference implementation: 624-word state, twist, tempering
// The state layout is numpy's RandomState.get_state(): key[624] + pos (uint32[625] here). The host keeps a numpy
// RandomState per env; its state is uploaded before a launch and downloaded after it (gym_cloth_amd/envs.py).
// These functions are plain C++ (host and device), so tests/test_host_logic.py pins them against numpy on the CPU.
#pragma once

#include <stdint.h>

#ifndef CLOTH_HD
#ifdef __HIPCC__
#define CLOTH_HD __host__ __device__
#else
#define CLOTH_HD
#endif
#endif

namespace clothhip {

constexpr int MT_N = 624, MT_M = 397;
constexpr int MT_WORDS = 626;        // key[624], pos, 1 pad word per env

CLOTH_HD inline uint32_t mt_twist_word(uint32_t cur, uint32_t nxt, uint32_t far) {
    const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}

// mt19937_gen: regenerate the 624 words in place, sequentially (one thread). Not inlined: every draw contains a possible
// refill, and the stepper kernel's cold paths should stay small.
CLOTH_HD __attribute__((noinline)) inline void mt_twist_serial(uint32_t *key) {
    int i = 0;
    for (; i < MT_N - MT_M; i++) key[i] = mt_twist_word(key[i], key[i + 1], key[i + MT_M]);
    for (; i < MT_N - 1; i++) key[i] = mt_twist_word(key[i], key[i + 1], key[i + (MT_M - MT_N)]);
    key[MT_N - 1] = mt_twist_word(key[MT_N - 1], key[0], key[MT_M - 1]);
}

CLOTH_HD inline uint32_t mt_next32(uint32_t *mt) {
    uint32_t pos = mt[MT_N];
    if (pos >= (uint32_t)MT_N) { mt_twist_serial(mt); pos = 0; }
    uint32_t y = mt[pos];
    mt[MT_N] = pos + 1;
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// RandomState.random_sample(): 53-bit double in [0, 1)
CLOTH_HD inline double mt_double(uint32_t *mt) {
    const uint32_t a = mt_next32(mt) >> 5, b = mt_next32(mt) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

// RandomState.uniform(low, high): low + (high - low) * random_sample()
CLOTH_HD inline double mt_uniform(uint32_t *mt, double low, double high) {
    const double range = high - low;
    return low + range * mt_double(mt);
}

// RandomState.randint(n) for 1 <= n <= 2^32: uniform integer in [0, n)
CLOTH_HD inline uint32_t mt_randint(uint32_t *mt, uint32_t n) {
    const uint32_t rng = n - 1;
    if (rng == 0) return 0;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    do { v = mt_next32(mt) & mask; } while (v > rng);
    return v;
}

// ClothEnv._randval_minabs (cloth_env.py:824-832); minabs <= 0 means "None"
CLOTH_HD inline double mt_randval_minabs(uint32_t *mt, double low, double high, double minabs) {
    double val = mt_uniform(mt, low, high);
    if (minabs > 0) {
        int guard = 0;
        while ((val < 0 ? -val : val) < minabs && guard++ < 100000) val = mt_uniform(mt, low, high);
    }
    return val;
}

// advance the stream by n 32-bit words without producing them (the values of the domain-randomisation draws of
// cloth_env.py:786-789 are not used by the '1d' observation path; only the stream position matters)
CLOTH_HD inline void mt_skip_serial(uint32_t *mt, uint64_t n) {
    while (n > 0) {
        uint32_t pos = mt[MT_N];
        if (pos >= (uint32_t)MT_N) { mt_twist_serial(mt); pos = 0; }
        const uint64_t take = n < (uint64_t)(MT_N - pos) ? n : (uint64_t)(MT_N - pos);
        mt[MT_N] = pos + (uint32_t)take;
        n -= take;
    }
}

}  // namespace clothhip

// Philox-4x32-10 counter-based generator shared by the kernels that draw dropout keep-bits
// (gda_act.hip, gda_critic.hip): stateless, keyed on the caller's seed, reproducible under hipGraph replay
// because the step counter is read from device memory.
#pragma once
#include <stdint.h>

struct GdaPhilox {
    static __device__ __forceinline__ void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    static __device__ __forceinline__ void gen(uint64_t seed, uint64_t hi, uint64_t lo, uint32_t (&out)[4]) {
        uint32_t c[4] = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) { round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = c[i];
    }
};

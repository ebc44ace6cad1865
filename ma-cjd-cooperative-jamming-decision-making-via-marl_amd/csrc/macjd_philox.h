// Philox4x32-10 counter-based RNG for the env-step kernel (device side).
//
// Not part of the reference: the reference draws np.random.rand() from the global MT19937 stream
// (simulation/environment.py:341,430), which is inherently sequential.  The batched kernel replaces
// it, when the caller supplies no uniforms, by a counter-based generator keyed by
// (seed; global env index, episode index, step counter, slot/4) so that every (env, episode, step, slot)
// has its own value independent of batch size, sharding over GPUs and launch geometry (include/macjd.h,
// macjd_step_io.u).  One block serves four slots: u = (word + 0.5) * 2^-32.
// Algorithm: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace macjd {

struct Philox4 {
    uint32_t v[4];
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        // one 32 x 32 -> 64 product per multiplier (v_mad_u64_u32 delivers both halves; a mul_hi + mul_lo pair is
        // two quarter-rate instructions)
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Philox4 r;
    r.v[0] = c0; r.v[1] = c1; r.v[2] = c2; r.v[3] = c3;
    return r;
}

// 53-bit uniform in [0,1) from two 32-bit words (hi word first)
__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
    const uint64_t bits = (((uint64_t)hi << 32) | lo) >> 11;
    return (double)bits * (1.0 / 9007199254740992.0);
}

// 32-bit uniform in (0,1): (w + 0.5) * 2^-32, exact in float64
__device__ __forceinline__ double u32_mid(uint32_t w) {
    return ((double)w + 0.5) * (1.0 / 4294967296.0);
}

// env-step stream (include/macjd.h): block `blk` of (global env, episode, step)
__device__ __forceinline__ Philox4 env_philox_block(uint64_t seed, uint64_t genv, uint32_t episode, uint32_t step,
                                                    uint32_t blk) {
    return philox4x32_10((uint32_t)genv, episode, step, blk, (uint32_t)seed,
                         (uint32_t)(seed >> 32) ^ (uint32_t)(genv >> 32));
}

}  // namespace macjd

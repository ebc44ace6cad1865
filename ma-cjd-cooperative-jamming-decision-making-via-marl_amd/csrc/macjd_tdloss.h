// macjd_tdloss.h — the block sums of the TD loss (reference core/qmix.py:155,190-194,212-213), shared by td_loss_kernel
// (macjd_nets.hip) and by the extra workgroup of the mixer's backward launch that logs them (macjd_mixer.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/macjd_nets.h"

namespace macjd {

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// One workgroup (any size that is a multiple of 64, <= 1024): the four sums over the B x Tm1 loss-carrying pairs, the
// logged values into io.stats[0..2] (and the mask sum into stats[3] when asked); returns the mask sum to every thread.
__device__ __forceinline__ float td_loss_sums(const macjd_tdloss_io& io, const bool write_mask_sum) {
    const int M = io.B * io.Tm1;
    float s_m = 0.f, s_e2 = 0.f, s_y = 0.f, s_t = 0.f;
#pragma unroll 4
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const int b = i / io.Tm1, t = i - b * io.Tm1;
        const float r = io.reward[b * io.r_sb + t * io.r_st];
        const float term = io.terminated[b * io.t_sb + t * io.t_st] ? 1.0f : 0.0f;
        const float m = io.filled[b * io.f_sb + t * io.f_st] ? 1.0f : 0.0f;
        const float y = io.y[b * io.y_sb + t];
        const float target = r + io.gamma * (1.0f - term) * io.tq[b * io.tq_sb + t];   // qmix.py:155
        const float e = (y - target) * m;                                // qmix.py:190-193
        s_m += m; s_e2 += e * e; s_y += y; s_t += target;
    }
    // the four block sums share their barriers (3 instead of 12); each sum is a
    // wave_sum per wave, then a wave_sum over the per-wave values by wave 0
    __shared__ float s4[4][16], tot4[4];
    {
        const float q[4] = {wave_sum(s_m), wave_sum(s_e2), wave_sum(s_y), wave_sum(s_t)};
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s4[k][wave] = q[k];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t = (lane < (int)(blockDim.x >> 6)) ? s4[k][lane] : 0.0f;
                t = wave_sum(t);
                if (lane == 0) tot4[k] = t;
            }
        }
        __syncthreads();
    }
    const float tot_m = tot4[0], tot_e2 = tot4[1], tot_y = tot4[2], tot_t = tot4[3];
    if (threadIdx.x == 0) {
        io.stats[0] = tot_e2 / tot_m;                                    // qmix.py:194
        io.stats[1] = tot_y / (float)M;
        io.stats[2] = tot_t / (float)M;
        if (write_mask_sum) io.stats[3] = tot_m;
    }
    return tot_m;
}

}  // namespace macjd

// macjd_nets.hip — fused agent-side kernels for MI355X (gfx950, wave64).  C-ABI: include/macjd_nets.h
//
// qhead_select_kernel: the MP-DQN multi-pass Q-head for ALL discrete actions in one pass, plus the
// availability mask, epsilon-greedy choice and the gather of the chosen power.  Replaces the reference's
// per-action Python loop (core/mac.py:115-164 / core/qmix.py:256-274 over
// RNNAgent.get_q_value_for_action, core/networks.py:131-180): A x (full, one_hot, cat, Linear, ReLU,
// Linear) launches become one launch and the [N, A, H] intermediate never exists.
//
// Mapping: one lane = one (env, agent) row.  The per-action columns W1[:, H+a], the power column
// W1[:, H+A] and w2 are wave-uniform, so they arrive through scalar loads (s_load) and feed the VALU
// as SGPR operands; the only vector memory traffic is the row's own base[H] (16-B loads), its P[A]
// and the outputs.  The A accumulators live in registers (kernel templated on A).
// At rollout / learner sizes (~10^4 rows) that is one wave per CU grinding through H x A serial
// iterations, so for small batches a workgroup is HS = 8 waves over the SAME 64 rows, wave w taking the
// hidden units [w H/8, (w+1) H/8) (weights stay wave-uniform); the partial sums meet in LDS and are
// added in fixed wave order by wave 0, which then does the selection (14 us -> 6 us at 12 288 rows with HS = 4;
// HS = 8 took the rollout step from 0.0739 to 0.0727 ms and the train step from 0.3756 to 0.3726 ms, HS = 2 is 3 % slower).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "../../include/macjd.h"
#include "../../include/macjd_nets.h"
#include "macjd_err.h"
#include "macjd_tdloss.h"
#include "macjd_philox.h"

namespace macjd {

static int set_nets_err(int code, const char* msg) { return set_err(code, "%s", msg); }

__device__ __forceinline__ bool avail_at(const macjd_qhead_io& io, int64_t e, int j, int a) {
    if (!io.avail) return true;
    const int64_t off = e * io.av_se + (int64_t)j * io.av_sj + (int64_t)a * io.av_sa;
    return (io.avail_elem_size == 8) ? (((const int64_t*)io.avail)[off] != 0)
                                     : (((const int32_t*)io.avail)[off] != 0);
}

#ifndef MACJD_QHEAD_HS
#define MACJD_QHEAD_HS 8
#endif
constexpr int QHEAD_HS = MACJD_QHEAD_HS;   // hidden-unit splits (waves per 64 rows) of the small-batch launch

template <int AT, int HS>
__global__ void __launch_bounds__(HS > 4 ? 64 * HS : 256) qhead_select_kernel(const macjd_qhead_io io) {
    constexpr int AMAX = AT ? AT : 64;
    const int A = AT ? AT : io.A;
    const int H = io.H;
    // HS == 1: every thread is a row.  HS > 1: blockDim = 64 * HS, lane = row within the workgroup's 64 rows,
    // wave = hidden-unit split
    const int wave = (HS > 1) ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
    const int64_t n = (HS > 1) ? (int64_t)blockIdx.x * 64 + (threadIdx.x & 63)
                               : (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool row_live = n < io.n_rows;
    if (HS == 1 && !row_live) return;
    __shared__ float s_part[(HS > 1) ? (HS - 1) * AMAX * 64 : 1];
    const int h_chunk = (HS > 1) ? (((H + HS - 1) / HS + 3) & ~3) : H;   // multiple of 4: keeps the 16-B loads
    const int h_lo = wave * h_chunk, h_hi = (h_lo + h_chunk < H) ? h_lo + h_chunk : H;

    float p[AMAX], acc[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; ++a) {
        const bool live = (AT || a < A) && row_live;
        p[a] = live ? io.P_all[n * io.p_ld + a] : 0.0f;
        acc[a] = 0.0f;
    }
    const float* __restrict__ brow = io.base + (row_live ? n : 0) * io.base_ld;
    const bool vec4 = ((io.base_ld & 3) == 0) && ((((uintptr_t)io.base) & 15) == 0);
    for (int h0 = h_lo; h0 < h_hi; h0 += 4) {
        float b4[4];
        if (vec4 && h0 + 4 <= H) {
            const float4 v = *reinterpret_cast<const float4*>(brow + h0);
            b4[0] = v.x; b4[1] = v.y; b4[2] = v.z; b4[3] = v.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) b4[k] = (h0 + k < H) ? brow[h0 + k] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int h = h0 + k;
            if (h >= H) break;
            // wave-uniform row of the first layer: [W_a[h][0..A-1], w_p[h]] is contiguous in W1
            const float* __restrict__ wrow = io.W1 + (int64_t)h * io.w1_ld + H;
            const float wp = wrow[A];
            const float w2h = io.w2[h];
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                if (!AT && a >= A) break;
                float t = b4[k] + wrow[a];        // (W_h h + b1)[h] + W1[h, H+a]
                t = fmaf(p[a], wp, t);            // + W1[h, H+A] * P_a
                t = fmaxf(t, 0.0f);               // ReLU (networks.py:77)
                acc[a] = fmaf(t, w2h, acc[a]);    // second layer (networks.py:78)
            }
        }
    }
    if (HS > 1) {   // partial sums of waves 1..HS-1 -> LDS; wave 0 adds them in wave order and carries on alone
        const int lane = threadIdx.x & 63;
        if (wave > 0) {
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                if (!AT && a >= A) break;
                s_part[((wave - 1) * AMAX + a) * 64 + lane] = acc[a];
            }
        }
        __syncthreads();
        if (wave > 0 || !row_live) return;
        if constexpr (AMAX > 17) {
            // wide action sets: one wave's partial sums at a time (fully unrolled, the (HS - 1) x A loads are all hoisted
            // to the top and hold (HS - 1) x A registers: 231 at A = 33 — scratch spills)
#pragma unroll 1
            for (int w = 0; w < HS - 1; ++w)
#pragma unroll
                for (int a = 0; a < AMAX; ++a) {
                    if (!AT && a >= A) break;
                    acc[a] += s_part[(w * AMAX + a) * 64 + lane];
                }
        } else {
#pragma unroll
            for (int w = 0; w < HS - 1; ++w)
#pragma unroll
                for (int a = 0; a < AMAX; ++a) {
                    if (!AT && a >= A) break;
                    acc[a] += s_part[(w * AMAX + a) * 64 + lane];
                }
        }
    }
    const float b2 = io.b2[0];
#pragma unroll
    for (int a = 0; a < AMAX; ++a) {
        if (!AT && a >= A) break;
        acc[a] += b2;
        if (io.Q) io.Q[n * io.q_ld + a] = acc[a];
    }
    if (io.argmax_out) {   // first maximum of the unmasked values (qmix.py:138-143)
        int am = 0;
        float aq = acc[0];
#pragma unroll
        for (int a = 1; a < AMAX; ++a) {
            if (!AT && a >= A) break;
            if (acc[a] > aq) { aq = acc[a]; am = a; }
        }
        io.argmax_out[n] = am;
    }
    if (io.gather_idx && io.q_gather_out) {   // qmix.py:147
        const int64_t gi = io.gather_idx[n];
        float gq = 0.0f;
#pragma unroll
        for (int a = 0; a < AMAX; ++a) {
            if (!AT && a >= A) break;
            gq = ((int64_t)a == gi) ? acc[a] : gq;
        }
        io.q_gather_out[n] = gq;
    }
    if (!io.T_out32 && !io.T_out64) return;

    // ---- mask, greedy argmax (first maximum), epsilon-greedy; mac.py:142-146, action_selectors.py:34-62
    const int64_t e = n / io.n_agents;
    const int j = (int)(n - e * io.n_agents);
    int best = 0, n_avail = 0;
    float bestq = -INFINITY;
    uint64_t avail_bits = 0;
#pragma unroll
    for (int a = 0; a < AMAX; ++a) {
        if (!AT && a >= A) break;
        const bool av = avail_at(io, e, j, a);
        avail_bits |= av ? (1ull << a) : 0ull;
        n_avail += av ? 1 : 0;
        const float q = av ? acc[a] : -INFINITY;
        if (q > bestq) { bestq = q; best = a; }
    }
    int chosen = best;
    const float epsilon = io.eps_dev ? io.eps_dev[0] : io.epsilon;
    if (!io.greedy_only && epsilon > 0.0f) {
        const uint64_t counter = io.counter + (io.counter_dev ? io.counter_dev[0] : 0ull);
        const Philox4 r = philox4x32_10((uint32_t)n, (uint32_t)((uint64_t)n >> 32), (uint32_t)counter,
                                        (uint32_t)(counter >> 32), (uint32_t)io.seed, (uint32_t)(io.seed >> 32));
        const float u_pick = (float)(r.v[0] >> 8) * (1.0f / 16777216.0f);
        if (u_pick < epsilon) {
            // uniform over the available actions (torch.multinomial over the mask in the reference);
            // with no action available the reference falls back to uniform over all of them
            const int pool = n_avail > 0 ? n_avail : A;
            int k = (int)(((uint64_t)r.v[1] * (uint64_t)pool) >> 32);
            chosen = 0;
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                if (!AT && a >= A) break;
                const bool av = (n_avail > 0) ? ((avail_bits >> a) & 1ull) : true;
                if (av) { if (k == 0) chosen = a; --k; }
            }
        }
    }
    float pc = 0.0f;
#pragma unroll
    for (int a = 0; a < AMAX; ++a) pc = (a == chosen) ? p[a] : pc;
    if (io.T_out32) io.T_out32[e * io.t32_se + (int64_t)j * io.t32_sj] = chosen;
    if (io.T_out64) io.T_out64[e * io.t64_se + (int64_t)j * io.t64_sj] = chosen;
    if (io.P_out) io.P_out[e * io.po_se + (int64_t)j * io.po_sj] = pc;
}

}  // namespace macjd

extern "C" int macjd_qhead_select(const macjd_qhead_io* io, void* hip_stream) {
    using namespace macjd;
    if (!io) return set_nets_err(MACJD_EINVAL, "macjd_qhead_select: NULL io");
    if (io->n_rows < 0 || io->H < 1 || io->A < 1 || io->A > 64 || io->n_agents < 1)
        return set_nets_err(MACJD_EINVAL, "macjd_qhead_select: bad n_rows / H / A / n_agents (A <= 64)");
    if (!io->base || !io->P_all || !io->W1 || !io->w2 || !io->b2)
        return set_nets_err(MACJD_EINVAL, "macjd_qhead_select: NULL input");
    if (!io->Q && !io->T_out32 && !io->T_out64 && !io->argmax_out && !(io->gather_idx && io->q_gather_out))
        return set_nets_err(MACJD_EINVAL, "macjd_qhead_select: no output requested");
    if (io->avail && io->avail_elem_size != 4 && io->avail_elem_size != 8)
        return set_nets_err(MACJD_EINVAL, "macjd_qhead_select: avail_elem_size must be 4 or 8");
    if (io->n_rows == 0) return MACJD_OK;
    hipStream_t s = (hipStream_t)hip_stream;
    // small batches: 64 rows per workgroup, hidden units split over QHEAD_HS waves; large ones: 256 rows per
    // workgroup, one lane does its whole row
    const bool split = io->n_rows < (1 << 16) && io->H >= 16;
#define MACJD_QHEAD_LAUNCH(AT)                                                                                    \
    do {                                                                                                          \
        if (split)                                                                                                \
            hipLaunchKernelGGL((qhead_select_kernel<AT, QHEAD_HS>), dim3((unsigned)((io->n_rows + 63) / 64)),     \
                               dim3(64 * QHEAD_HS), 0, s, *io);                                                   \
        else                                                                                                      \
            hipLaunchKernelGGL((qhead_select_kernel<AT, 1>), dim3((unsigned)((io->n_rows + 255) / 256)), dim3(256), \
                               0, s, *io);                                                                        \
    } while (0)
    switch (io->A) {
        case 5:  MACJD_QHEAD_LAUNCH(5); break;
        case 7:  MACJD_QHEAD_LAUNCH(7); break;
        case 9:  MACJD_QHEAD_LAUNCH(9); break;
        case 17: MACJD_QHEAD_LAUNCH(17); break;
        case 33: MACJD_QHEAD_LAUNCH(33); break;
        default: MACJD_QHEAD_LAUNCH(0); break;
    }
#undef MACJD_QHEAD_LAUNCH
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_nets_err(MACJD_EDEVICE, hipGetErrorString(err));
    return MACJD_OK;
}

// ---------------------------------------------------------------------------------------------
// GRU sequence scan (learner unroll, inference only).
//
// One workgroup = one sequence (b, j) of one network; NW waves, one per SIMD.  The recurrence is a
// 3H x H mat-vec per step, strictly sequential in t, so the design minimises the per-step latency and
// spreads the B*J (x2 networks) independent sequences over as many CUs as there are sequences:
//   * lane l owns hidden units u = l + 64 i (i < H/64), in every wave (the gate math is replicated, so
//     every wave holds the complete new h in registers and no second barrier / broadcast is needed);
//   * wave w owns the K-slice k in [w KW, (w+1) KW) of the mat-vec: its 3 U KW weights W_hh[g H + u][k]
//     stay in VGPRs for all T steps; h[k] is pulled from the owning lane with v_readlane (wave-uniform
//     SGPR operand of the FMA);
//   * the NW partial sums per (gate, unit) meet in a double-buffered LDS tile, one barrier per step,
//     and are added in fixed wave order (deterministic);
//   * gi for step t+1 is loaded while step t computes; wave 0 stores h' (256-B coalesced rows).
namespace macjd {

// v_rcp_f32 / v_exp_f32 directly (1 ulp each): __frcp_rn is the CORRECTLY ROUNDED reciprocal and expands to the full
// div_scale / div_fmas / div_fixup sequence, ~12 instructions per gate on the serial per-step chain
__device__ __forceinline__ float gru_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float gru_tanh(float x) {
    // tanh(x) = 1 - 2 / (exp(2x) + 1); saturates cleanly: exp -> inf gives 1, exp -> 0 gives -1
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f);
}

// Prologue of the scan kernels for a static observation (macjd_gru_io.obs); results: s_gi [3H] in LDS and, when asked,
// io.p_out.  All NW waves of the workgroup take part; ends with a barrier.
// The two halves are independent given the observation row, so a launch that wants both runs them in DIFFERENT
// workgroups (blockIdx.z = 0: input transform + scan, blockIdx.z = 1: actor chain only): the actor's three mat-vecs
// (~9 us) leave the scan's critical path.
template <int H, int NW>
__device__ __forceinline__ void scan_prologue(const macjd_gru_io& io, const int net, const int b, const int j,
                                              float* __restrict__ s_pro, float* __restrict__ s_gi, const int lane,
                                              const int wave, const bool want_gi, const bool want_actor) {
    // Prologue for a static observation: this sequence's ONE observation row x -> gi = W_ih ReLU(fc1 x + b) + b_ih
    // and, when asked, the actor chain sigmoid(L3 ReLU(L2 ReLU(L1 x))).  Small mat-vecs on the VALU: an output is a
    // wave-wide dot product (lanes split k: every weight row is read as coalesced 256-B pieces from L2), outputs are
    // dealt round-robin to the NW waves; layers meet in LDS.
    float* xs = s_pro;              // [S]
    float* v1 = s_pro + 256;        // [<= 256] first hidden vector
    float* v2 = s_pro + 512;        // [<= 256] second hidden vector
    const int64_t row = io.obs_index ? io.obs_index[b] : (int64_t)b;
    const float* x = io.obs + row * io.obs_sb + (int64_t)j * io.obs_sj;
    const int S = io.S;
    for (int k = threadIdx.x; k < S; k += 64 * NW) xs[k] = x[k];
    __syncthreads();
    // (UN outputs per wave and pass, all their weight elements requested before the first is used — one output at a
    // time is one memory latency per output, 75 us for the five layers; NCH = ceil(K / 64) k-chunks per lane, a
    // compile-time count so that no load sits behind a branch)
    auto matvec = [&](const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ vin, int K,
                      int N, auto&& store) {   // store(o, value) for o in [0, N)
        auto body = [&](auto nch_c) {
            constexpr int NCH = decltype(nch_c)::value;
            constexpr int UN = 16 / NCH;          // 16 loads in flight per lane
            float xv[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int k = lane + 64 * c;
                xv[c] = (k < K) ? vin[k] : 0.0f;
            }
            for (int o0 = wave * UN; o0 < N; o0 += NW * UN) {
                float wv[UN][NCH], a[UN];
#pragma unroll
                for (int uu = 0; uu < UN; ++uu) {
                    const int o = (o0 + uu < N) ? o0 + uu : N - 1;        // clamped: no branch around the loads
                    const float* wr = W + (int64_t)o * K;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const int k = lane + 64 * c;
                        wv[uu][c] = wr[k < K ? k : K - 1];
                    }
                }
#pragma unroll
                for (int uu = 0; uu < UN; ++uu) {
                    float t = 0.0f;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) t = fmaf(wv[uu][c], xv[c], t);   // x is zero past K
                    a[uu] = wave_sum(t);
                }
                float mine = 0.0f;
#pragma unroll
                for (int uu = 0; uu < UN; ++uu) mine = (lane == uu) ? a[uu] : mine;
                if (lane < UN && o0 + lane < N) store(o0 + lane, mine + bias[o0 + lane]);
            }
        };
        const int nch = (K + 63) / 64;
        if (nch == 1) body(std::integral_constant<int, 1>{});
        else if (nch == 2) body(std::integral_constant<int, 2>{});
        else body(std::integral_constant<int, 4>{});
    };
    if (want_gi) {
        matvec(io.fc1_w[net], io.fc1_b[net], xs, S, H, [&](int o, float v) { v1[o] = fmaxf(v, 0.0f); });
        __syncthreads();
        matvec(io.w_ih[net], io.b_ih[net], v1, H, 3 * H, [&](int o, float v) { s_gi[o] = v; });
    }
    if (want_actor && io.p_out[net]) {
        const int Ah = io.Ah, A = io.A;
        __syncthreads();    // v1 is read by the W_ih product above: done before it is overwritten
        matvec(io.act_w[net][0], io.act_b[net][0], xs, S, Ah, [&](int o, float v) { v1[o] = fmaxf(v, 0.0f); });
        __syncthreads();
        matvec(io.act_w[net][1], io.act_b[net][1], v1, Ah, Ah, [&](int o, float v) { v2[o] = fmaxf(v, 0.0f); });
        __syncthreads();
        float* po = io.p_out[net] + ((int64_t)b * io.J + j) * A;
        matvec(io.act_w[net][2], io.act_b[net][2], v2, Ah, A, [&](int o, float v) { po[o] = 1.0f / (1.0f + expf(-v)); });
    }
    __syncthreads();
}

template <int H, int NW>
__global__ void __launch_bounds__(64 * NW) gru_sequence_kernel(const macjd_gru_io io) {
    constexpr int U = H / 64;    // hidden units per lane
    constexpr int KW = H / NW;   // K-slice per wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: the readlane selects below stay SALU
    const int net = blockIdx.y;
    const int seq = blockIdx.x;  // b * J + j
    const int b = seq / io.J, j = seq - b * io.J;
    const int T = io.T;
    const float* __restrict__ gi = io.gi[net];
    const float* __restrict__ whh = io.w_hh[net];
    const float* __restrict__ bhh = io.b_hh[net];
    float* __restrict__ out = io.h_out[net];

    __shared__ float s_part[2][NW][3][H];
    __shared__ float s_gi[3 * H];   // in-kernel input transform (static observation), see macjd_gru_io.obs
    __shared__ float s_pro[768];    // prologue scratch: observation row + two hidden vectors
    if (blockIdx.z == 1) {   // actor-only workgroup of a launch that split the prologue
        scan_prologue<H, NW>(io, net, b, j, s_pro, s_gi, lane, wave, false, true);
        return;
    }

    // weights of this wave's K-slice, resident for the whole sequence
    float w[3][U][KW];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const float* row = whh + (int64_t)(g * H + lane + 64 * i) * H + wave * KW;
#pragma unroll
            for (int kk = 0; kk < KW; kk += 4) {
                const float4 v = *reinterpret_cast<const float4*>(row + kk);
                w[g][i][kk] = v.x; w[g][i][kk + 1] = v.y; w[g][i][kk + 2] = v.z; w[g][i][kk + 3] = v.w;
            }
        }
    // gi ring of THREE register slots indexed at compile time (the t-loop is unrolled by 3): step t reads slot t % 3
    // and requests row t + 2 into slot (t + 2) % 3.  No slot is ever copied: a rotation by register moves
    // (cur = next; next = next2) needs the load issued in the same step to have landed, i.e. a full vmcnt(0) wait and
    // one global-load latency on every step of the serial chain (that was 0.8 us / step; the chain itself is ~0.4).
    float bias[3][U], h[U], ring[3][3][U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int u = lane + 64 * i;
#pragma unroll
        for (int g = 0; g < 3; ++g) bias[g][i] = bhh[g * H + u];
        h[i] = io.h0[net] ? io.h0[net][(int64_t)b * (io.h0_sb[net] ? io.h0_sb[net] : (int64_t)io.J * H) + (int64_t)j * H + u] : 0.0f;
    }
    // io.reserved != 0 ("gi_static"): gi is [B, 1, J, 3H], the same input transform at every step (static observation)
    const bool gi_inkernel = io.obs != nullptr;
    const bool gi_static = io.reserved != 0 || gi_inkernel;
    if (gi_inkernel) scan_prologue<H, NW>(io, net, b, j, s_pro, s_gi, lane, wave, true, gridDim.z == 1);
    auto gi_row = [&](int t) -> const float* {
        return gi_inkernel ? (const float*)s_gi
                           : gi_static ? gi + ((int64_t)b * io.J + j) * (3 * H)
                                       : gi + (((int64_t)b * T + t) * io.J + j) * (3 * H);
    };
    if (T > 0) {
        const float* r0 = gi_row(0);
        const float* r1 = gi_row(T > 1 ? 1 : 0);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int i = 0; i < U; ++i) {
                ring[0][g][i] = r0[g * H + lane + 64 * i];
                ring[1][g][i] = r1[g * H + lane + 64 * i];
            }
    }

    // one step of the recurrence; SLOT = t % 3 is a compile-time constant
    auto step = [&](auto slot_c, int t) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int PRE = (SLOT + 2) % 3;
        {   // request the input transform TWO steps ahead (independent of the recurrence)
            const float* rn = gi_row(t + 2 < T ? t + 2 : T - 1);
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int i = 0; i < U; ++i) ring[PRE][g][i] = rn[g * H + lane + 64 * i];
        }
        float acc[3][U];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int i = 0; i < U; ++i) acc[g][i] = 0.0f;
#pragma unroll
        for (int kk = 0; kk < KW; ++kk) {
            const int k = wave * KW + kk;  // wave-uniform
            float hk;  // h[k], broadcast from its owning lane (k is wave-uniform -> SGPR lane select)
            if constexpr (U == 1) {
                hk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h[0]), k));
            } else {
                const float src = (k < 64) ? h[0] : h[U - 1];
                hk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(src), k & 63));
            }
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int i = 0; i < U; ++i) acc[g][i] = fmaf(w[g][i][kk], hk, acc[g][i]);
        }
        const int buf = t & 1;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int i = 0; i < U; ++i) s_part[buf][wave][g][lane + 64 * i] = acc[g][i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int u = lane + 64 * i;
            float gh[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float sacc = 0.0f;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) sacc += s_part[buf][ww][g][u];
                gh[g] = sacc + bias[g][i];
            }
            // sigmoid / tanh on the hardware exp2 + rcp units (~1 ulp each) instead of IEEE division + libm tanhf:
            // these sit on the strictly sequential per-step chain (about a third of its instructions); the 1e-5
            // tolerance on h after T = 100 steps is checked against the step-by-step float32 reference
            const float r = gru_sigmoid(ring[SLOT][0][i] + gh[0]);
            const float z = gru_sigmoid(ring[SLOT][1][i] + gh[1]);
            const float nn = gru_tanh(ring[SLOT][2][i] + r * gh[2]);
            h[i] = (h[i] - nn) * z + nn;
        }
        if (wave == 0) {
            float* orow = out + (((int64_t)b * T + t) * io.J + j) * H;
#pragma unroll
            for (int i = 0; i < U; ++i) orow[lane + 64 * i] = h[i];
        }
    };

    for (int t = 0; t < T; t += 3) {
        step(std::integral_constant<int, 0>{}, t);
        if (t + 1 < T) step(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < T) step(std::integral_constant<int, 2>{}, t + 2);
    }
}


// H = 64, second form of the scan: the four waves split the hidden UNITS, the four lanes of a quad split K.
// Lane l of wave w owns unit u = 16 w + (l >> 2) and the K-quarter [16 (l & 3), +16): 48 weights in VGPRs, the quarter of
// h arrives as four ds_read_b128 (four distinct addresses per wave, 64 B apart: conflict-free broadcasts), 24 packed FMAs
// (v_pk_fma_f32) form the three partial gate sums, the four lanes of a quad meet through two DPP adds (quad_perm:
// v_add_f32_dpp, no LDS round trip, every lane gets the total), every lane of the quad evaluates the gates of its unit
// (bit-identical copies) and lane 0 of the quad publishes h' in the other half of a double-buffered 256-B LDS vector:
// ONE barrier per step, like the K-split form above, but 24 + 6 instead of 64 + ~30 instructions in front of the gates
// and 4 instead of 12 LDS reads behind the barrier (the K-split form pays 16 v_readlane -> SGPR -> FMA hand-overs per
// wave and step, and sums NW partials per gate).  96 sequences x 101 steps: 56 -> 32 us.
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float quad_sum(float x) {
    // x + x[lane ^ 1], then + [lane ^ 2]: quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E
    x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0xB1, 0xF, 0xF, false));
    x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x4E, 0xF, 0xF, false));
    return x;
}
// tanh(x) = 1 - 2 / (2^(x 2 log2 e) + 1) with the constants folded: 5 dependent instructions (gru_tanh above: 7)
__device__ __forceinline__ float gru_tanh_folded(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

template <bool STATIC>
__global__ void __launch_bounds__(256) gru_sequence_units_kernel(const macjd_gru_io io) {
    constexpr int H = 64, NW = 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane & 3;                // K-quarter
    const int u = 16 * wave + (lane >> 2); // hidden unit
    const int net = blockIdx.y;
    const int seq = blockIdx.x;  // b * J + j
    const int b = seq / io.J, j = seq - b * io.J;
    const int T = io.T;
    const float* __restrict__ gi = io.gi[net];
    const float* __restrict__ whh = io.w_hh[net];
    const float* __restrict__ bhh = io.b_hh[net];
    float* __restrict__ out = io.h_out[net];

    __shared__ __attribute__((aligned(16))) float s_h[2][H];
    __shared__ float s_gi[3 * H];
    __shared__ float s_pro[768];
    if (STATIC && blockIdx.z == 1) {   // actor-only workgroup of a launch that split the prologue
        scan_prologue<H, NW>(io, net, b, j, s_pro, s_gi, lane, wave, false, true);
        return;
    }

    float2v w2[3][8];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const float4* row = reinterpret_cast<const float4*>(whh + (int64_t)(g * H + u) * H + 16 * q);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const float4 v = row[k4];
            w2[g][2 * k4] = float2v{v.x, v.y};
            w2[g][2 * k4 + 1] = float2v{v.z, v.w};
        }
    }
    float bias[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bias[g] = bhh[g * H + u];
    float h = io.h0[net] ? io.h0[net][(int64_t)b * (io.h0_sb[net] ? io.h0_sb[net] : (int64_t)io.J * H) + (int64_t)j * H + u] : 0.0f;

    // STATIC (one input transform per sequence: computed by the prologue, or gi [B,1,J,3H]): three registers for the whole
    // scan.  Otherwise gi [B,T,J,3H] streams through a ring of three register slots filled two steps ahead (see
    // gru_sequence_kernel) — by GLOBAL loads only: a pointer that may also address LDS compiles to flat loads, which
    // count on lgkmcnt as well, so the wait for the step's ds_reads would wait for the prefetch too.
    const bool gi_inkernel = STATIC && io.obs != nullptr;
    if (gi_inkernel) scan_prologue<H, NW>(io, net, b, j, s_pro, s_gi, lane, wave, true, gridDim.z == 1);
    float ring[3][3];
    if (STATIC) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float v = gi_inkernel ? s_gi[g * H + u] : gi[((int64_t)b * io.J + j) * (3 * H) + g * H + u];
            ring[0][g] = v; ring[1][g] = v; ring[2][g] = v;
        }
    } else if (T > 0) {
        const float* r0 = gi + (((int64_t)b * T) * io.J + j) * (3 * H);
        const float* r1 = gi + (((int64_t)b * T + (T > 1 ? 1 : 0)) * io.J + j) * (3 * H);
#pragma unroll
        for (int g = 0; g < 3; ++g) { ring[0][g] = r0[g * H + u]; ring[1][g] = r1[g * H + u]; }
    }
    if (q == 0) s_h[0][u] = h;
    __syncthreads();

    auto step = [&](auto slot_c, int t) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int PRE = (SLOT + 2) % 3;
        const int buf = t & 1;
        const float4* hp = reinterpret_cast<const float4*>(&s_h[buf][16 * q]);
        const float4 h0v = hp[0], h1v = hp[1], h2v = hp[2], h3v = hp[3];
        if constexpr (!STATIC) {
            const float* rn = gi + (((int64_t)b * T + (t + 2 < T ? t + 2 : T - 1)) * io.J + j) * (3 * H);
#pragma unroll
            for (int g = 0; g < 3; ++g) ring[PRE][g] = rn[g * H + u];
        }
        const float2v hv[8] = {float2v{h0v.x, h0v.y}, float2v{h0v.z, h0v.w}, float2v{h1v.x, h1v.y}, float2v{h1v.z, h1v.w},
                               float2v{h2v.x, h2v.y}, float2v{h2v.z, h2v.w}, float2v{h3v.x, h3v.y}, float2v{h3v.z, h3v.w}};
        float gh[3];
        float2v acc[3] = {float2v{0.0f, 0.0f}, float2v{0.0f, 0.0f}, float2v{0.0f, 0.0f}};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = __builtin_elementwise_fma(w2[g][kk], hv[kk], acc[g]);
#pragma unroll
        for (int g = 0; g < 3; ++g) gh[g] = quad_sum(acc[g].x + acc[g].y) + bias[g];
        const float r = gru_sigmoid(ring[SLOT][0] + gh[0]);
        const float z = gru_sigmoid(ring[SLOT][1] + gh[1]);
        const float nn = gru_tanh_folded(ring[SLOT][2] + r * gh[2]);
        h = (h - nn) * z + nn;
        if (q == 0) {
            s_h[buf ^ 1][u] = h;
            out[(((int64_t)b * T + t) * io.J + j) * H + u] = h;
        }
        __syncthreads();
    };
    for (int t = 0; t < T; t += 3) {
        step(std::integral_constant<int, 0>{}, t);
        if (t + 1 < T) step(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < T) step(std::integral_constant<int, 2>{}, t + 2);
    }
}

}  // namespace macjd

extern "C" int macjd_gru_sequence(const macjd_gru_io* io, void* hip_stream) {
    using namespace macjd;
    if (!io) return set_nets_err(MACJD_EINVAL, "macjd_gru_sequence: NULL io");
    if (io->n_nets < 1 || io->n_nets > 2 || io->B < 0 || io->T < 0 || io->J < 1)
        return set_nets_err(MACJD_EINVAL, "macjd_gru_sequence: bad n_nets / B / T / J");
    if (io->H != 64 && io->H != 128) return set_nets_err(MACJD_EUNSUPPORTED, "macjd_gru_sequence: H must be 64 or 128");
    for (int n = 0; n < io->n_nets; ++n) {
        if ((!io->gi[n] && !io->obs) || !io->w_hh[n] || !io->b_hh[n] || !io->h_out[n])
            return set_nets_err(MACJD_EINVAL, "macjd_gru_sequence: NULL pointer");
        if (io->obs && (!io->fc1_w[n] || !io->fc1_b[n] || !io->w_ih[n] || !io->b_ih[n]))
            return set_nets_err(MACJD_EINVAL, "macjd_gru_sequence: in-kernel input transform needs fc1 / W_ih");
    }
    if (io->obs && (io->S < 1 || io->S > 256))
        return set_nets_err(MACJD_EINVAL, "macjd_gru_sequence: in-kernel input transform needs 1 <= S <= 256");
    for (int n = 0; n < io->n_nets; ++n)
        if (io->obs && io->p_out[n]) {
            if (io->Ah < 1 || io->Ah > 256 || io->A < 1 || io->A > 64)
                return set_nets_err(MACJD_EINVAL, "macjd_gru_sequence: in-kernel actor needs Ah <= 256, A <= 64");
            for (int l = 0; l < 3; ++l)
                if (!io->act_w[n][l] || !io->act_b[n][l])
                    return set_nets_err(MACJD_EINVAL, "macjd_gru_sequence: in-kernel actor needs its three layers");
        }
    if (io->B == 0 || io->T == 0) return MACJD_OK;
    // with an in-kernel actor chain: a second layer of workgroups (blockIdx.z = 1) computes it beside the scan
    const bool actor = io->obs && (io->p_out[0] || (io->n_nets > 1 && io->p_out[1]));
    const dim3 g((unsigned)(io->B * io->J), (unsigned)io->n_nets, actor ? 2u : 1u);
    hipStream_t s = (hipStream_t)hip_stream;
    // H = 64: the unit-split scan; MACJD_GRU_SCAN=ksplit keeps the K-split form (A/B runs, tests compare the two)
    const bool ksplit = env_options().gru_ksplit;   // (read once; macjd_reload_options() after changing it in-process)
    if (io->H == 64 && !ksplit) {
        if (io->obs || io->reserved) hipLaunchKernelGGL(gru_sequence_units_kernel<true>, g, dim3(256), 0, s, *io);
        else hipLaunchKernelGGL(gru_sequence_units_kernel<false>, g, dim3(256), 0, s, *io);
    }
    else if (io->H == 64) hipLaunchKernelGGL((gru_sequence_kernel<64, 4>), g, dim3(256), 0, s, *io);
    else hipLaunchKernelGGL((gru_sequence_kernel<128, 8>), g, dim3(512), 0, s, *io);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_nets_err(MACJD_EDEVICE, hipGetErrorString(err));
    return MACJD_OK;
}

// ---------------------------------------------------------------------------------------------
// QMix mixer tail: clamp -> (q . w1 + b1) -> ELU -> (. wf + v), forward and backward.
// One wave per row m, lane = embed column e (Em = 64 in the reference config; other sizes loop in chunks
// of 64).  Every load is a contiguous 256-B row segment (w1_raw[m, j, :], b1_raw[m, :], wf_raw[m, :]); the
// two contractions are J fused multiply-adds per lane plus ONE wave reduction (forward) / J wave
// reductions (backward, dL/dq_j).  Replaces ~12 (forward) / ~25 (backward) elementwise + reduce launches
// per mixer call of the unfused form.
namespace macjd {

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

template <bool BACKWARD>
__global__ void __launch_bounds__(256) mixer_tail_kernel(const macjd_mixer_io io) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m >= io.M) return;  // whole waves exit together (one row per wave)
    const int J = io.J, Em = io.Em;
    const float* __restrict__ q = io.q + m * J;
    const float* __restrict__ w1r = io.w1_raw + m * (int64_t)J * Em;
    float ysum = 0.0f;
    float gq_acc[MACJD_MAX_JAMMERS];
    if (BACKWARD) {
#pragma unroll
        for (int j = 0; j < MACJD_MAX_JAMMERS; ++j) gq_acc[j] = 0.0f;
    }
    const float gy = BACKWARD ? io.gy[m] : 0.0f;
    for (int e = lane; e < ((Em + 63) & ~63); e += 64) {
        const bool live = e < Em;
        const float b1r = live ? io.b1_raw[m * (io.b1_ld ? io.b1_ld : (int64_t)Em) + e] : 0.0f;
        const float wfr = live ? io.wf_raw[m * Em + e] : 0.0f;
        float hid = clampf(b1r, -5.0f, 5.0f);
        for (int j = 0; j < J; ++j) {
            const float w = live ? clampf(w1r[(int64_t)j * Em + e], 0.0f, 5.0f) : 0.0f;
            hid = fmaf(q[j], w, hid);                          // bmm(agent_qs, w1) + b1, networks.py:304
        }
        const float h = hid > 0.0f ? hid : expm1f(hid);        // F.elu
        const float wf = clampf(wfr, 0.0f, 5.0f);
        if (!BACKWARD) {
            ysum += live ? h * wf : 0.0f;                      // bmm(hidden, w_final), networks.py:307
        } else if (live) {
            const float ghid = gy * wf * (hid > 0.0f ? 1.0f : h + 1.0f);   // ELU'(x) = exp(x) = elu(x) + 1 for x <= 0
            io.gwf_raw[m * Em + e] = (wfr >= 0.0f && wfr <= 5.0f) ? gy * h : 0.0f;
            io.gb1_raw[m * Em + e] = (b1r >= -5.0f && b1r <= 5.0f) ? ghid : 0.0f;
            for (int j = 0; j < J; ++j) {
                const float wr = w1r[(int64_t)j * Em + e];
                io.gw1_raw[m * (int64_t)J * Em + (int64_t)j * Em + e] = (wr >= 0.0f && wr <= 5.0f) ? ghid * q[j] : 0.0f;
                gq_acc[j] += ghid * clampf(wr, 0.0f, 5.0f);
            }
        }
    }
    if (!BACKWARD) {
        const float tot = wave_sum(ysum);
        if (lane == 0) io.y[m] = tot + clampf(io.v_raw[m], -5.0f, 5.0f);
    } else {
        for (int j = 0; j < J; ++j) {
            const float t = wave_sum(gq_acc[j]);
            if (lane == 0) io.gq[m * J + j] = t;
        }
        if (lane == 0) {
            const float vr = io.v_raw[m];
            io.gv_raw[m] = (vr >= -5.0f && vr <= 5.0f) ? gy : 0.0f;
        }
    }
}

static int mixer_launch(const macjd_mixer_io* io, void* hip_stream, bool backward) {
    if (!io) return set_nets_err(MACJD_EINVAL, "macjd_mixer_tail: NULL io");
    if (io->M < 0 || io->J < 1 || io->J > MACJD_MAX_JAMMERS || io->Em < 1)
        return set_nets_err(MACJD_EINVAL, "macjd_mixer_tail: bad M / J / Em");
    if (!io->q || !io->w1_raw || !io->b1_raw || !io->wf_raw || !io->v_raw)
        return set_nets_err(MACJD_EINVAL, "macjd_mixer_tail: NULL input");
    if (!backward && !io->y) return set_nets_err(MACJD_EINVAL, "macjd_mixer_tail_forward: NULL y");
    if (backward && (!io->gy || !io->gq || !io->gw1_raw || !io->gb1_raw || !io->gwf_raw || !io->gv_raw))
        return set_nets_err(MACJD_EINVAL, "macjd_mixer_tail_backward: NULL gradient pointer");
    if (io->M == 0) return MACJD_OK;
    const dim3 g((unsigned)((io->M + 3) / 4)), b(256);
    hipStream_t s = (hipStream_t)hip_stream;
    if (backward) hipLaunchKernelGGL((mixer_tail_kernel<true>), g, b, 0, s, *io);
    else hipLaunchKernelGGL((mixer_tail_kernel<false>), g, b, 0, s, *io);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_nets_err(MACJD_EDEVICE, hipGetErrorString(err));
    return MACJD_OK;
}

}  // namespace macjd

extern "C" int macjd_mixer_tail_forward(const macjd_mixer_io* io, void* hip_stream) {
    return macjd::mixer_launch(io, hip_stream, false);
}
extern "C" int macjd_mixer_tail_backward(const macjd_mixer_io* io, void* hip_stream) {
    return macjd::mixer_launch(io, hip_stream, true);
}

// ---------------------------------------------------------------------------------------------
// TD target + masked MSE + gradient + logged means: one workgroup, two passes over M = B * (T-1) pairs
// (M = 3168 for the reference batch).  Replaces ~13 elementwise / reduce launches forward and ~8 backward.
namespace macjd {

__global__ void __launch_bounds__(1024) td_loss_kernel(const macjd_tdloss_io io) {
    const float tot_m = td_loss_sums(io, io.gy != nullptr);   // (macjd_tdloss.h)
    if (!io.gy) return;   // the logged sums only (the gradient was formed elsewhere; stats[3] belongs to the caller)
    const float scale = 2.0f / tot_m;
    const int cols = (int)io.gy_cols;
    for (int i = threadIdx.x; i < io.B * cols; i += blockDim.x) {
        const int b = i / cols, t = i - b * cols;
        float g = 0.0f;   // columns past Tm1 (full-length rows) carry no loss
        if (t < io.Tm1) {
            const float term = io.terminated[b * io.t_sb + t * io.t_st] ? 1.0f : 0.0f;
            const float m = io.filled[b * io.f_sb + t * io.f_st] ? 1.0f : 0.0f;
            const float target = io.reward[b * io.r_sb + t * io.r_st] + io.gamma * (1.0f - term) * io.tq[b * io.tq_sb + t];
            g = scale * m * (io.y[b * io.y_sb + t] - target);
        }
        io.gy[b * io.gy_sb + t] = g;
    }
}

}  // namespace macjd

namespace macjd {
// sum of the loss mask of a batch (the tot_m of td_loss_kernel; sums of 0 / 1 are exact in any order): what
// macjd_mixer_fused_backward_td needs of the loss's global sums, from the gathered batch alone
__global__ void __launch_bounds__(1024) td_mask_sum_kernel(const macjd_tdloss_io io, float* __restrict__ out) {
    __shared__ float s_w[16];
    const int M = io.B * io.Tm1;
    float s = 0.0f;
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const int b = i / io.Tm1, t = i - b * io.Tm1;
        s += io.filled[b * io.f_sb + t * io.f_st] ? 1.0f : 0.0f;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = (threadIdx.x < (blockDim.x >> 6)) ? s_w[threadIdx.x] : 0.0f;
        t = wave_sum(t);
        if (threadIdx.x == 0) out[0] = t;
    }
}
}  // namespace macjd

extern "C" int macjd_td_mask_sum(const macjd_tdloss_io* io, float* out, void* hip_stream) {
    using namespace macjd;
    if (!io || !out || io->B < 1 || io->Tm1 < 1 || !io->filled) return set_nets_err(MACJD_EINVAL, "macjd_td_mask_sum: bad argument");
    hipLaunchKernelGGL(td_mask_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)hip_stream, *io, out);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_nets_err(MACJD_EDEVICE, hipGetErrorString(err));
    return MACJD_OK;
}

extern "C" int macjd_td_loss(const macjd_tdloss_io* io, void* hip_stream) {
    using namespace macjd;
    if (!io || io->B < 1 || io->Tm1 < 1) return set_nets_err(MACJD_EINVAL, "macjd_td_loss: bad B / Tm1");
    if (!io->y || !io->tq || !io->reward || !io->terminated || !io->filled || !io->stats)
        return set_nets_err(MACJD_EINVAL, "macjd_td_loss: NULL pointer");
    if (io->y_sb < io->Tm1 || io->tq_sb < io->Tm1 || (io->gy && (io->gy_cols < io->Tm1 || io->gy_sb < io->gy_cols)))
        return set_nets_err(MACJD_EINVAL, "macjd_td_loss: bad strides");
    hipLaunchKernelGGL(td_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)hip_stream, *io);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_nets_err(MACJD_EDEVICE, hipGetErrorString(err));
    return MACJD_OK;
}

// ---------------------------------------------------------------------------------------------
// Gradient clipping + Adam on one flat vector: (1) per-block partial sums of g^2, (2) every block re-reduces the
// <= 256 partials (deterministic order), derives the clip coefficient and the bias corrections, and updates its
// slice.  Replaces ~20 foreach / scalar launches of clip_grad_norm_ + capturable Adam.
namespace macjd {

constexpr int ADAM_BLOCKS = 256;

__global__ void __launch_bounds__(256) adam_sqnorm_kernel(const macjd_adam_io io) {
    __shared__ float smem[4];
    float s = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < io.n; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = io.grad[i];
        s = fmaf(g, g, s);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        io.partials[blockIdx.x] = (smem[0] + smem[1]) + (smem[2] + smem[3]);
        // the optimiser's step counter advances HERE: nothing in this kernel reads it, and the update kernel that follows
        // reads the advanced value (a third one-thread launch used to do this after the update)
        if (blockIdx.x == 0) io.step[0] += 1.0f;
    }
}

// adam_sqnorm_kernel + lnparam_kernel as one launch (macjd_clip_adam_step_ln): workgroups [0, nb) sum the squares of their
// slices of the gradient vector EXCEPT the LayerNorm ranges, workgroup nb + k computes dgamma[k] / dbeta[k] with
// lnparam_kernel's own arithmetic, stores them and contributes their squares as partial nb + k.
__global__ void __launch_bounds__(256) adam_sqnorm_ln_kernel(const macjd_adam_io io, const macjd_lnparam_io ln, const int64_t g_off,
                                                             const int64_t b_off, const int nb) {
    __shared__ float smem[8];
    if ((int)blockIdx.x < nb) {
        float s = 0.0f;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < io.n; i += (int64_t)nb * blockDim.x) {
            const bool is_ln = (i >= g_off && i < g_off + ln.K) || (i >= b_off && i < b_off + ln.K);
            const float g = is_ln ? 0.0f : io.grad[i];
            s = fmaf(g, g, s);
        }
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            io.partials[blockIdx.x] = (smem[0] + smem[1]) + (smem[2] + smem[3]);
            if (blockIdx.x == 0) io.step[0] += 1.0f;   // as in adam_sqnorm_kernel
        }
        return;
    }
    const int k = (int)blockIdx.x - nb;
    float sg = 0.0f, sb = 0.0f;
    for (int c = threadIdx.x; c < ln.C; c += blockDim.x) {
        const float w = ln.W[(int64_t)c * ln.w_ld + k];
        sg = fmaf(w, ln.G[(int64_t)c * ln.g_ld + k], sg);
        sb = fmaf(w, ln.gb[c], sb);
    }
    sg = wave_sum(sg);
    sb = wave_sum(sb);
    if ((threadIdx.x & 63) == 0) { smem[threadIdx.x >> 6] = sg; smem[4 + (threadIdx.x >> 6)] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float dg = (smem[0] + smem[1]) + (smem[2] + smem[3]), db = (smem[4] + smem[5]) + (smem[6] + smem[7]);
        ln.dgamma[k] = dg;
        ln.dbeta[k] = db;
        io.partials[nb + k] = fmaf(db, db, dg * dg);
    }
}

// ---- device-side draw of the next update's episodes (include/macjd_nets.h, macjd_sampler_io) ----
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {   // MurmurHash3 finaliser
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
// one block; every thread reads the counter, thread 0 advances it after the barrier
__device__ __forceinline__ void sample_episodes_block(const macjd_sampler_io& sp) {
    const int64_t c = *sp.counter;
    const int64_t N = (int64_t)*sp.n_stored;
    __syncthreads();
    if (threadIdx.x == 0) *sp.counter = c + 1;
    if (N < 1) return;
    const Philox4 ka = philox4x32_10((uint32_t)c, (uint32_t)((uint64_t)c >> 32), 0x53414d50u, 0u, (uint32_t)sp.seed, (uint32_t)(sp.seed >> 32));
    const Philox4 kb = philox4x32_10((uint32_t)c, (uint32_t)((uint64_t)c >> 32), 0x53414d50u, 1u, (uint32_t)sp.seed, (uint32_t)(sp.seed >> 32));
    const uint32_t key[8] = {ka.v[0], ka.v[1], ka.v[2], ka.v[3], kb.v[0], kb.v[1], kb.v[2], kb.v[3]};
    int k = 2;                                   // bits of the permuted domain: 2^k >= N
    while (((int64_t)1 << k) < N) ++k;
    const int rb = k - k / 2, lb = k / 2;        // right / left half widths (rb >= lb >= 1)
    const uint32_t rmask = (1u << rb) - 1u, lmask = (1u << lb) - 1u;
    for (int t = threadIdx.x; t < sp.n; t += blockDim.x) {
        if (N < sp.n) { sp.idx_out[t] = t % N; continue; }
        uint32_t x = (uint32_t)t;
        for (int walk = 0; walk < 64; ++walk) {  // cycle-walking: expected < 2 passes (2^k < 2 N)
            uint32_t L = x >> rb, R = x & rmask;
#pragma unroll
            for (int r = 0; r < 8; r += 2) {
                L = (L ^ fmix32(R ^ key[r])) & lmask;
                R = (R ^ fmix32(L ^ key[r + 1])) & rmask;
            }
            x = (L << rb) | R;
            if ((int64_t)x < N) break;
        }
        sp.idx_out[t] = ((int64_t)x < N) ? (int64_t)x : (int64_t)(x % (uint32_t)N);
    }
}

__global__ void __launch_bounds__(256) sample_episodes_kernel(const macjd_sampler_io sp) { sample_episodes_block(sp); }

template <bool SAMPLE>
__global__ void __launch_bounds__(256) adam_update_kernel(const macjd_adam_io io, const int n_partials, const macjd_sampler_io next) {
    __shared__ float s_tot;
    if (threadIdx.x < 64) {
        float s = 0.0f;
        for (int i = threadIdx.x; i < n_partials; i += 64) s += io.partials[i];
        s = wave_sum(s);
        if (threadIdx.x == 0) s_tot = s;
    }
    __syncthreads();
    const float total_norm = sqrtf(s_tot);
    float coef = io.max_norm / (total_norm + 1e-6f);
    coef = coef < 1.0f ? coef : 1.0f;
    const float step = io.step[0];   // already advanced by adam_sqnorm_kernel
    const float bc1 = 1.0f - powf(io.beta1, step);
    const float bc2 = 1.0f - powf(io.beta2, step);
    const float step_size = io.lr / bc1;
    const float bc2_sqrt = sqrtf(bc2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < io.n; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = io.grad[i] * coef;
        io.grad[i] = g;   // clip_grad_norm_ scales the gradients in place: .grad holds the clipped values afterwards
        float m = io.exp_avg[i], v = io.exp_avg_sq[i];
        m = m + (1.0f - io.beta1) * (g - m);
        v = v * io.beta2 + (1.0f - io.beta2) * g * g;
        io.exp_avg[i] = m;
        io.exp_avg_sq[i] = v;
        const float denom = sqrtf(v) / bc2_sqrt + io.eps;
        io.param[i] -= step_size * (m / denom);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) io.grad_norm[0] = total_norm;
    // every reader of the current batch's indices ran in an earlier launch of this stream: the next batch is drawn here
    if (SAMPLE && blockIdx.x == 0) sample_episodes_block(next);
}

__global__ void __launch_bounds__(256) gather_rows_kernel(const macjd_gather_io io) {
    const int k = blockIdx.y;       // tensor
    const int row = blockIdx.z;     // output row
    const int64_t words = io.row_bytes[k] >> 2;
    const uint32_t* __restrict__ src = (const uint32_t*)((const char*)io.src[k] + io.idx[row] * io.row_bytes[k]);
    const int64_t dpitch = io.dst_row_bytes[k] ? io.dst_row_bytes[k] : io.row_bytes[k];
    uint32_t* __restrict__ dst = (uint32_t*)((char*)io.dst[k] + (int64_t)row * dpitch);
    const bool vec = ((io.row_bytes[k] & 15) == 0) && ((dpitch & 15) == 0) && ((((uintptr_t)io.src[k]) & 15) == 0) &&
                     ((((uintptr_t)io.dst[k]) & 15) == 0);
    if (vec) {
        const int64_t n4 = words >> 2;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
            ((uint4*)dst)[i] = ((const uint4*)src)[i];
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x)
            dst[i] = src[i];
    }
}

}  // namespace macjd

static int sampler_args_ok(const macjd_sampler_io* sp) {
    return sp && sp->idx_out && sp->n >= 1 && sp->n_stored && sp->counter;
}

extern "C" int macjd_sample_episodes(const macjd_sampler_io* io, void* hip_stream) {
    using namespace macjd;
    if (!sampler_args_ok(io)) return set_nets_err(MACJD_EINVAL, "macjd_sample_episodes: bad argument");
    hipLaunchKernelGGL(sample_episodes_kernel, dim3(1), dim3(256), 0, (hipStream_t)hip_stream, *io);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_nets_err(MACJD_EDEVICE, hipGetErrorString(err));
    return MACJD_OK;
}

extern "C" int macjd_clip_adam_step_sample(const macjd_adam_io* io, const macjd_sampler_io* next, void* hip_stream) {
    using namespace macjd;
    if (!io || io->n < 1 || !io->param || !io->grad || !io->exp_avg || !io->exp_avg_sq || !io->step || !io->grad_norm ||
        !io->partials)
        return set_nets_err(MACJD_EINVAL, "macjd_clip_adam_step: bad argument");
    if (next && !sampler_args_ok(next)) return set_nets_err(MACJD_EINVAL, "macjd_clip_adam_step_sample: bad sampler argument");
    hipStream_t s = (hipStream_t)hip_stream;
    int blocks = (int)((io->n + 255) / 256);
    if (blocks > ADAM_BLOCKS) blocks = ADAM_BLOCKS;
    hipLaunchKernelGGL(adam_sqnorm_kernel, dim3(blocks), dim3(256), 0, s, *io);
    if (next) hipLaunchKernelGGL(adam_update_kernel<true>, dim3(blocks), dim3(256), 0, s, *io, blocks, *next);
    else hipLaunchKernelGGL(adam_update_kernel<false>, dim3(blocks), dim3(256), 0, s, *io, blocks, macjd_sampler_io{});
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_nets_err(MACJD_EDEVICE, hipGetErrorString(err));
    return MACJD_OK;
}

extern "C" int macjd_clip_adam_step(const macjd_adam_io* io, void* hip_stream) {
    return macjd_clip_adam_step_sample(io, nullptr, hip_stream);
}

extern "C" int macjd_clip_adam_step_ln(const macjd_adam_io* io, const macjd_sampler_io* next, const macjd_lnparam_io* ln,
                                       int64_t gamma_off, int64_t beta_off, void* hip_stream) {
    using namespace macjd;
    if (!io || io->n < 1 || !io->param || !io->grad || !io->exp_avg || !io->exp_avg_sq || !io->step || !io->grad_norm ||
        !io->partials)
        return set_nets_err(MACJD_EINVAL, "macjd_clip_adam_step_ln: bad argument");
    if (next && !sampler_args_ok(next)) return set_nets_err(MACJD_EINVAL, "macjd_clip_adam_step_ln: bad sampler argument");
    if (!ln || ln->C < 1 || ln->K < 1 || ln->K > 1024 || !ln->W || !ln->G || !ln->gb || ln->w_ld < ln->K || ln->g_ld < ln->K)
        return set_nets_err(MACJD_EINVAL, "macjd_clip_adam_step_ln: bad LayerNorm argument");
    if (gamma_off < 0 || beta_off < 0 || gamma_off + ln->K > io->n || beta_off + ln->K > io->n ||
        ln->dgamma != io->grad + gamma_off || ln->dbeta != io->grad + beta_off)
        return set_nets_err(MACJD_EINVAL, "macjd_clip_adam_step_ln: dgamma / dbeta must be the named ranges of the gradient vector");
    hipStream_t s = (hipStream_t)hip_stream;
    int blocks = (int)((io->n + 255) / 256);
    if (blocks > ADAM_BLOCKS) blocks = ADAM_BLOCKS;
    hipLaunchKernelGGL(adam_sqnorm_ln_kernel, dim3(blocks + ln->K), dim3(256), 0, s, *io, *ln, gamma_off, beta_off, blocks);
    if (next) hipLaunchKernelGGL(adam_update_kernel<true>, dim3(blocks), dim3(256), 0, s, *io, blocks + ln->K, *next);
    else hipLaunchKernelGGL(adam_update_kernel<false>, dim3(blocks), dim3(256), 0, s, *io, blocks + ln->K, macjd_sampler_io{});
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_nets_err(MACJD_EDEVICE, hipGetErrorString(err));
    return MACJD_OK;
}

extern "C" int macjd_gather_rows(const macjd_gather_io* io, void* hip_stream) {
    using namespace macjd;
    if (!io || io->n_tensors < 1 || io->n_tensors > 8 || io->n_rows < 0 || !io->idx)
        return set_nets_err(MACJD_EINVAL, "macjd_gather_rows: bad argument");
    for (int k = 0; k < io->n_tensors; ++k)
        if (!io->src[k] || !io->dst[k] || io->row_bytes[k] < 4 || (io->row_bytes[k] & 3) || (io->dst_row_bytes[k] & 3) ||
            (io->dst_row_bytes[k] && io->dst_row_bytes[k] < io->row_bytes[k]))
            return set_nets_err(MACJD_EINVAL, "macjd_gather_rows: bad tensor (row bytes must be a positive multiple of 4)");
    if (io->n_rows == 0) return MACJD_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(8, io->n_tensors, io->n_rows), dim3(256), 0, (hipStream_t)hip_stream, *io);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_nets_err(MACJD_EDEVICE, hipGetErrorString(err));
    return MACJD_OK;
}

// ---------------------------------------------------------------------------------------------
// Q-head input rows for the taken action and LayerNorm forward: two small launch-count savers of the learner update
// (one launch instead of arange + eq + cast + cat; one launch instead of torch's moments + normalise pair).
namespace macjd {

__global__ void __launch_bounds__(256) qhead_input_kernel(const macjd_qinput_io io) {
    const int W = io.H + io.A + 1;
    const int64_t total = io.n_rows * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / W;
        const int c = (int)(i - n * W);
        float v;
        if (c < io.H) {
            v = io.h[n * io.h_ld + c];
        } else if (c < io.H + io.A) {
            const int64_t a = (io.idx_elem_size == 8) ? ((const int64_t*)io.idx)[n] : (int64_t)((const int32_t*)io.idx)[n];
            v = (a == (int64_t)(c - io.H)) ? 1.0f : 0.0f;
        } else {
            v = io.P[n];
        }
        io.out[n * io.out_ld + c] = v;
    }
}

// one wave per row (rows strided over the grid's waves); lane l holds columns l, l + 64, ... (S <= 1024)
__global__ void __launch_bounds__(256) layernorm_forward_kernel(const macjd_layernorm_io io) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int S = io.S;
    const float inv_s = 1.0f / (float)S;
    for (int64_t m = wave; m < io.M; m += n_waves) {
        const float* __restrict__ x = io.x + m * io.x_ld;
        float v[16];
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = lane + 64 * i;
            v[i] = (c < S) ? x[c] : 0.0f;
            sum += v[i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float mean = sum * inv_s;
        float sq = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = lane + 64 * i;
            const float d = (c < S) ? v[i] - mean : 0.0f;
            sq = fmaf(d, d, sq);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        const float rstd = rsqrtf(sq * inv_s + io.eps);
        float* __restrict__ y = io.y + m * io.y_ld;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = lane + 64 * i;
            if (c < S) {
                const float g = io.gamma ? io.gamma[c] : 1.0f, b = io.beta ? io.beta[c] : 0.0f;
                y[c] = (v[i] - mean) * rstd * g + b;
                if (io.xhat) io.xhat[m * io.xhat_ld + c] = (v[i] - mean) * rstd;
            }
        }
        if (lane == 0) {
            if (io.mean) io.mean[m] = mean;
            if (io.rstd) io.rstd[m] = rstd;
        }
    }
}

}  // namespace macjd

extern "C" int macjd_qhead_input(const macjd_qinput_io* io, void* hip_stream) {
    using namespace macjd;
    if (!io || io->n_rows < 0 || io->H < 1 || io->A < 1 || !io->h || !io->idx || !io->P || !io->out)
        return set_nets_err(MACJD_EINVAL, "macjd_qhead_input: bad argument");
    if (io->idx_elem_size != 4 && io->idx_elem_size != 8)
        return set_nets_err(MACJD_EINVAL, "macjd_qhead_input: idx_elem_size must be 4 or 8");
    if (io->h_ld < io->H || io->out_ld < io->H + io->A + 1)
        return set_nets_err(MACJD_EINVAL, "macjd_qhead_input: row stride smaller than the row");
    if (io->n_rows == 0) return MACJD_OK;
    const int64_t total = io->n_rows * (io->H + io->A + 1);
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(qhead_input_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                       (hipStream_t)hip_stream, *io);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_nets_err(MACJD_EDEVICE, hipGetErrorString(err));
    return MACJD_OK;
}

extern "C" int macjd_layernorm_forward(const macjd_layernorm_io* io, void* hip_stream) {
    using namespace macjd;
    if (!io || io->M < 0 || io->S < 1 || io->S > 1024 || !io->x || !io->y)
        return set_nets_err(MACJD_EINVAL, "macjd_layernorm_forward: bad argument (1 <= S <= 1024)");
    if (io->x_ld < io->S || io->y_ld < io->S)
        return set_nets_err(MACJD_EINVAL, "macjd_layernorm_forward: row stride smaller than the row");
    if (io->M == 0) return MACJD_OK;
    const int64_t blocks = (io->M + 3) / 4;
    hipLaunchKernelGGL(layernorm_forward_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                       (hipStream_t)hip_stream, *io);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_nets_err(MACJD_EDEVICE, hipGetErrorString(err));
    return MACJD_OK;
}

// ---------------------------------------------------------------------------------------------
// GRU gates of the rollout step: elementwise, no LDS (it runs beside the actor's LDS-heavy dense chain in the replayed
// graph), 4 hidden units per thread with 16-byte loads; writes h' to the hidden buffer and the staging row.
namespace macjd {

template <bool VEC>
__global__ void __launch_bounds__(256) gru_gates_kernel(const macjd_grugates_io io) {
    const int H = io.H, q = H >> 2;
    const int64_t total = io.n_rows * q;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / q;
        const int u = (int)(i - n * q) * 4;
        const float* gi = io.gi + n * io.gi_ld + u;
        const float* gh = io.gh + n * io.gh_ld + u;
        const float* hpp = io.h + n * io.h_ld + u;
        float ir[4], iz[4], in_[4], hr[4], hz[4], hn[4], hp[4], out[4];
        auto ld4 = [](const float* p, float* d) {
            if (VEC) {
                const float4 v = *reinterpret_cast<const float4*>(p);
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) d[k] = p[k];
            }
        };
        ld4(gi, ir); ld4(gi + H, iz); ld4(gi + 2 * H, in_);
        ld4(gh, hr); ld4(gh + H, hz); ld4(gh + 2 * H, hn);
        ld4(hpp, hp);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float r = 1.0f / (1.0f + expf(-(ir[k] + hr[k])));
            const float z = 1.0f / (1.0f + expf(-(iz[k] + hz[k])));
            // tanh(x) = 1 - 2 / (exp(2x) + 1): saturates cleanly (exp -> inf gives 1, exp -> 0 gives -1), abs. error ~1e-7
            const float nn = 1.0f - 2.0f / (expf(2.0f * (in_[k] + r * hn[k])) + 1.0f);
            out[k] = (hp[k] - nn) * z + nn;
        }
        float* o1 = io.h_out + n * io.ho_ld + u;
        if (VEC) *reinterpret_cast<float4*>(o1) = float4{out[0], out[1], out[2], out[3]};
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k) o1[k] = out[k];
        }
        if (io.h_out2) {
            float* o2 = io.h_out2 + n * io.ho2_ld + u;
            if (VEC) *reinterpret_cast<float4*>(o2) = float4{out[0], out[1], out[2], out[3]};
            else {
#pragma unroll
                for (int k = 0; k < 4; ++k) o2[k] = out[k];
            }
        }
    }
}

}  // namespace macjd

extern "C" int macjd_gru_gates(const macjd_grugates_io* io, void* hip_stream) {
    using namespace macjd;
    if (!io || io->n_rows < 0 || io->H < 4 || (io->H & 3) || !io->gi || !io->gh || !io->h || !io->h_out)
        return set_nets_err(MACJD_EINVAL, "macjd_gru_gates: bad argument (H must be a positive multiple of 4)");
    // gi_ld == 0: one input-transform row broadcast to every row (static observation shared by all envs / agents)
    if ((io->gi_ld != 0 && io->gi_ld < 3 * io->H) || io->gh_ld < 3 * io->H || io->h_ld < io->H || io->ho_ld < io->H ||
        (io->h_out2 && io->ho2_ld < io->H))
        return set_nets_err(MACJD_EINVAL, "macjd_gru_gates: row stride smaller than the row");
    if (io->n_rows == 0) return MACJD_OK;
    const int64_t total = io->n_rows * (io->H >> 2);
    const int64_t blocks = (total + 255) / 256;
    auto al = [](const void* p, int64_t ld) { return (((uintptr_t)p) % 16 == 0) && (ld % 4 == 0); };
    const bool vec = al(io->gi, io->gi_ld) && al(io->gh, io->gh_ld) && al(io->h, io->h_ld) && al(io->h_out, io->ho_ld) &&
                     (!io->h_out2 || al(io->h_out2, io->ho2_ld));
    const dim3 g((unsigned)(blocks < 8192 ? blocks : 8192)), b(256);
    if (vec) hipLaunchKernelGGL(gru_gates_kernel<true>, g, b, 0, (hipStream_t)hip_stream, *io);
    else hipLaunchKernelGGL(gru_gates_kernel<false>, g, b, 0, (hipStream_t)hip_stream, *io);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_nets_err(MACJD_EDEVICE, hipGetErrorString(err));
    return MACJD_OK;
}

// ---------------------------------------------------------------------------------------------
// Row dot product for Linear layers with one output feature: 16 lanes per row (float4 each, K <= 1024 in up to 16
// passes), xor-shuffle tree inside the 16-lane group (fixed order: deterministic).
namespace macjd {

__global__ void __launch_bounds__(256) rowdot_kernel(const macjd_rowdot_io io) {
    const int sub = threadIdx.x & 15;
    const int64_t rows_per_block = blockDim.x >> 4;
    const bool vec = ((io.x_ld & 3) == 0) && ((((uintptr_t)io.x) & 15) == 0) && ((((uintptr_t)io.w) & 15) == 0);
    const float bias = io.b ? io.b[0] : 0.0f;
    for (int64_t n = (int64_t)blockIdx.x * rows_per_block + (threadIdx.x >> 4); n < io.n_rows;
         n += (int64_t)gridDim.x * rows_per_block) {
        const float* x = io.x + n * io.x_ld;
        float s = 0.0f;
        for (int k = sub * 4; k < io.K; k += 64) {
            float xv[4], wv[4];
            if (vec) {
                const float4 a = *reinterpret_cast<const float4*>(x + k), c = *reinterpret_cast<const float4*>(io.w + k);
                xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w;
                wv[0] = c.x; wv[1] = c.y; wv[2] = c.z; wv[3] = c.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) { xv[i] = x[k + i]; wv[i] = io.w[k + i]; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s = fmaf(xv[i], wv[i], s);
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 8, 64);
        if (sub == 0) io.y[n] = s + bias;
    }
}

}  // namespace macjd

extern "C" int macjd_rowdot(const macjd_rowdot_io* io, void* hip_stream) {
    using namespace macjd;
    if (!io || io->n_rows < 0 || io->K < 4 || io->K > 1024 || (io->K & 3) || !io->x || !io->w || !io->y)
        return set_nets_err(MACJD_EINVAL, "macjd_rowdot: bad argument (K a multiple of 4 in 4..1024)");
    if (io->x_ld < io->K) return set_nets_err(MACJD_EINVAL, "macjd_rowdot: row stride smaller than the row");
    if (io->n_rows == 0) return MACJD_OK;
    const int64_t blocks = (io->n_rows + 15) / 16;
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                       (hipStream_t)hip_stream, *io);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_nets_err(MACJD_EDEVICE, hipGetErrorString(err));
    return MACJD_OK;
}

// ---------------------------------------------------------------------------------------------
// Backward of the split / ReLU / split around the mixer's merged first layer: one elementwise launch.
namespace macjd {

__global__ void __launch_bounds__(256) splitrelu_backward_kernel(const macjd_splitrelu_bwd_io io, const int Cr) {
    const int W = Cr + io.Cp;
    const int64_t total = io.M * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / W;
        const int c = (int)(i - m * W);
        float v;
        if (c >= Cr) {
            v = io.g_pass ? io.g_pass[m * io.gp_ld + (c - Cr)] : 0.0f;
        } else {
            int k = 0, start = 0;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (q + 1 < io.n_blocks && c >= start + io.width[k]) { start += io.width[k]; ++k; }
            const float* gk = (k == 0) ? io.g[0] : (k == 1) ? io.g[1] : (k == 2) ? io.g[2] : io.g[3];
            const int64_t ld = (k == 0) ? io.g_ld[0] : (k == 1) ? io.g_ld[1] : (k == 2) ? io.g_ld[2] : io.g_ld[3];
            const float* ow = (k == 0) ? io.outer_w[0] : (k == 1) ? io.outer_w[1] : (k == 2) ? io.outer_w[2] : io.outer_w[3];
            const float gv = !gk ? 0.0f : ow ? gk[m * ld] * ow[c - start] : gk[m * ld + (c - start)];
            v = (io.act[m * io.act_ld + c] > 0.0f) ? gv : 0.0f;
        }
        io.gout[m * io.gout_ld + c] = v;
    }
}

}  // namespace macjd

extern "C" int macjd_splitrelu_backward(const macjd_splitrelu_bwd_io* io, void* hip_stream) {
    using namespace macjd;
    if (!io || io->M < 0 || io->n_blocks < 1 || io->n_blocks > 4 || io->Cp < 0 || !io->act || !io->gout)
        return set_nets_err(MACJD_EINVAL, "macjd_splitrelu_backward: bad argument");
    int Cr = 0;
    for (int k = 0; k < io->n_blocks; ++k) {
        if (io->width[k] < 1) return set_nets_err(MACJD_EINVAL, "macjd_splitrelu_backward: bad block width");
        if (io->g[k] && io->g_ld[k] < (io->outer_w[k] ? 1 : io->width[k]))
            return set_nets_err(MACJD_EINVAL, "macjd_splitrelu_backward: bad g_ld");
        Cr += io->width[k];
    }
    if (io->act_ld < Cr || io->gout_ld < Cr + io->Cp || (io->g_pass && io->gp_ld < io->Cp))
        return set_nets_err(MACJD_EINVAL, "macjd_splitrelu_backward: row stride smaller than the row");
    if (io->M == 0) return MACJD_OK;
    const int64_t total = io->M * (Cr + io->Cp);
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(splitrelu_backward_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                       (hipStream_t)hip_stream, *io, Cr);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_nets_err(MACJD_EDEVICE, hipGetErrorString(err));
    return MACJD_OK;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm gamma / beta gradients from the following Linear layer's weight-gradient quantities (see the header).
namespace macjd {

__global__ void __launch_bounds__(256) lnparam_kernel(const macjd_lnparam_io io) {
    __shared__ float s_g[4], s_b[4];
    const int k = blockIdx.x;
    float sg = 0.0f, sb = 0.0f;
    for (int c = threadIdx.x; c < io.C; c += blockDim.x) {
        const float w = io.W[(int64_t)c * io.w_ld + k];
        sg = fmaf(w, io.G[(int64_t)c * io.g_ld + k], sg);
        sb = fmaf(w, io.gb[c], sb);
    }
    sg = wave_sum(sg);
    sb = wave_sum(sb);
    if ((threadIdx.x & 63) == 0) { s_g[threadIdx.x >> 6] = sg; s_b[threadIdx.x >> 6] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        io.dgamma[k] = (s_g[0] + s_g[1]) + (s_g[2] + s_g[3]);
        io.dbeta[k] = (s_b[0] + s_b[1]) + (s_b[2] + s_b[3]);
    }
}

}  // namespace macjd

extern "C" int macjd_layernorm_param_grad(const macjd_lnparam_io* io, void* hip_stream) {
    using namespace macjd;
    if (!io || io->C < 1 || io->K < 1 || !io->W || !io->G || !io->gb || !io->dgamma || !io->dbeta)
        return set_nets_err(MACJD_EINVAL, "macjd_layernorm_param_grad: bad argument");
    if (io->w_ld < io->K || io->g_ld < io->K) return set_nets_err(MACJD_EINVAL, "macjd_layernorm_param_grad: bad leading dimension");
    hipLaunchKernelGGL(lnparam_kernel, dim3((unsigned)io->K), dim3(256), 0, (hipStream_t)hip_stream, *io);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_nets_err(MACJD_EDEVICE, hipGetErrorString(err));
    return MACJD_OK;
}

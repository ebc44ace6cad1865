// macjd_mixer.hip — the QMix mixer (reference core/networks.py:250-315) as one MFMA chain per direction.
// C-ABI and the algebra: include/macjd_nets.h (macjd_mixerf_io).
//
// Geometry.  M = batch x steps is a few thousand rows (3232 for the reference batch), the widest layer has 384 output
// columns: tiling the rows 64 at a time (as the agent's dense-chain kernel does for its 10^4..10^5 rows) would leave
// 4/5 of the chip idle.  Here a workgroup owns 16 rows — ONE MFMA row tile — and its four waves split the output
// COLUMNS of each layer, so 3232 rows are 202 workgroups of 256 threads and a row's whole chain (LayerNorm, merged
// first layer, ReLU, both second layers, clamps, the two contractions with q and w_final, ELU) never leaves the
// workgroup:
//   phase 0  LayerNorm of the 16 rows (wave w: rows 4w..4w+3, 16 lanes per row)              -> LDS As [16][16 SQ]
//   phase 1  [16, 384] = As W1^T + b1, ReLU on the first 320 columns; wave w: column tiles 6w..6w+5  -> LDS Hs
//   phase 2  wave w = embed block e in [16w, 16w+16): the J tiles w1_raw[:, j, e-block] and the tile wf_raw[:, e-block]
//            stay in accumulators; hid = q . clamp(w1) + clamp(b1), ELU, . clamp(wf): registers; the sum over a row's
//            16 columns is a 4-step xor shuffle, the sum over the 4 embed blocks goes through 256 B of LDS.
// Operands.  A (activations) comes from LDS by ds_read_b128 with the permuted-k assignment of macjd_mlp.hip (lane group
// g supplies k = 16 Q + 4 g + jj in MFMA jj of quad Q: one float4 feeds four MFMAs; pitches = 8 mod 16 floats are
// conflict-free).  B (weights) is NOT staged: with 16 rows per workgroup a staged image would be read once, so every
// lane loads its own fragments W[16 t + (lane & 15)][16 Q + 4 g ..+3] straight from L2 (the weights are ~230 KB,
// shared by all workgroups), and ALL of them are requested at the top of the kernel — they depend on nothing — so the
// whole chain waits for memory once.  ~200 VGPRs of fragments at 3j/4r; one wave per SIMD has 512.
// Numerics: v_mfma_f32_16x16x4_f32 is an exact-f32 fmaf chain; the tail uses the same expressions (fmaf chain over the
// agents starting from clamp(b1), expm1f) as mixer_tail_kernel.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/macjd.h"
#include "../../include/macjd_nets.h"
#include "macjd_err.h"
#include "macjd_tdloss.h"

namespace macjd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));   // weight rows are only 4-byte aligned (S = 46)

#ifndef MACJD_MX_ABLATE
#define MACJD_MX_ABLATE 0   // timing experiments only (results wrong when non-zero): 1 no fragment loads, 2 no phase-1
#endif                      // MFMAs, 4 no phase-2 MFMAs, 8 no state loads, 16 no tail math
constexpr int MX_ABL = MACJD_MX_ABLATE;
constexpr int MX_HH = 128, MX_EM = 64;
constexpr int MX_N1 = 2 * MX_HH + 2 * MX_EM;   // 384 columns of the merged first layer
constexpr int MX_RELU = 2 * MX_HH + MX_EM;     // the first 320 of them pass a ReLU
constexpr int MX_LDH = MX_N1 + 8;              // LDS pitch of the first-layer output (= 8 mod 16)
constexpr int MX_KQ2 = MX_HH / 16;             // quads (16 k each) of the second layers

__device__ __forceinline__ float mx_clamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
__device__ __forceinline__ float mx_sum16(float x) {   // sum over the 16 lanes that share lane >> 4
    x += __shfl_xor(x, 1, 64);
    x += __shfl_xor(x, 2, 64);
    x += __shfl_xor(x, 4, 64);
    x += __shfl_xor(x, 8, 64);
    return x;
}

// B fragments (all quads) of one 16-column tile whose weight row is `row` of a row-major [*, K] matrix.  No branch
// around any load: a divergent `if` with a load inside makes hipcc drain the memory counter (s_waitcnt vmcnt(0)) at
// every join, which serialised the ~50 fragment loads of a wave one L2 latency after the other (16 us per launch
// instead of ~5).  RAGGED (K not a multiple of 16, e.g. S = 46): the last quad is four dword loads from clamped
// addresses, zeroed past the row by selects.
template <int NQ, bool RAGGED>
__device__ __forceinline__ void mx_load_frags(f32x4 (&dst)[NQ], const float* __restrict__ W, int K, int row, int g) {
    const float* p = W + (int64_t)row * K;
#pragma unroll
    for (int Q = 0; Q < NQ; ++Q) {
        const int k0 = 16 * Q + 4 * g;
        if (MX_ABL & 1) {
            dst[Q] = f32x4{(float)row, (float)g, 1.0f, 0.5f};
        } else if (!RAGGED || Q < NQ - 1) {   // compile-time
            dst[Q] = *reinterpret_cast<const f32x4_u*>(p + k0);
        } else {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int k = k0 + jj;
                const float v = p[k < K ? k : K - 1];
                dst[Q][jj] = (k < K) ? v : 0.0f;
            }
        }
    }
}

// Second layers + tail for this wave's embed block, shared by the forward kernel and the backward's recomputation:
// acc2[j] = tile (j, eb) of w1_raw WITHOUT its bias, accf = tile eb of wf_raw without its bias.
template <int J>
__device__ __forceinline__ void mx_second_layers(const float* __restrict__ Hs, const f32x4 (&B2)[J][MX_KQ2],
                                                 const f32x4 (&Bf)[MX_KQ2], f32x4 (&acc2)[J], f32x4& accf, int li, int g) {
#pragma unroll
    for (int j = 0; j < J; ++j) acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    accf = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* a1p = Hs + li * MX_LDH + 4 * g;            // h_w1: columns [0, Hh)
    const float* afp = a1p + MX_HH;                         // h_wf: columns [Hh, 2 Hh)
#pragma unroll
    for (int Q = 0; Q < ((MX_ABL & 4) ? 0 : MX_KQ2); ++Q) {
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(a1p + 16 * Q);
        const f32x4 af = *reinterpret_cast<const f32x4*>(afp + 16 * Q);
        // k-step outer, tiles inner: consecutive MFMAs go to different accumulators (a dependent one waits 40 cycles,
        // the issue interval is 32)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
            for (int j = 0; j < J; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[jj], B2[j][Q][jj], acc2[j], 0, 0, 0);
            accf = __builtin_amdgcn_mfma_f32_16x16x4f32(af[jj], Bf[Q][jj], accf, 0, 0, 0);
        }
    }
}

// v_raw of row (lane & 15): h_V . wV2 + bV2, valid in the lanes with g == 0 afterwards (every wave may call it)
__device__ __forceinline__ float mx_v_raw(const float* __restrict__ Hs, const float (&wv)[16], float bV2, int li, int g) {
    float s = 0.0f;   // wv[k] = wV2[16 g + k], preloaded
    const float* hv = Hs + li * MX_LDH + 2 * MX_HH + 16 * g;
#pragma unroll
    for (int k = 0; k < 16; ++k) s = fmaf(hv[k], wv[k], s);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    return s + bV2;
}

// LDS of one forward workgroup (the body below is a device function of (arguments, workgroup index, LDS): the eval and
// the target mixer of one learner update run as ONE grid, macjd_mixer_fused_forward_pair)
template <int SQ>
struct MixerFwdLds {
    alignas(16) float As[16 * (16 * SQ + 8)];
    alignas(16) float Hs[16 * MX_LDH];
    float part[4][16];
};

// LATE2: the second layers' fragments (B2 / Bf, 32 (J + 1) registers) are requested behind the first layer's MFMAs
// instead of at the top: ~130 fewer live registers at J = 3 — two workgroups per CU, which the paired launch (twice the
// workgroups) needs to stay one round — for one exposed L2 latency, which the CU's other workgroup covers.
template <int J, int SQ, bool SAVE, bool LATE2 = false>
__device__ __forceinline__ void mixer_fused_forward_body(const macjd_mixerf_io& io, const int blk, MixerFwdLds<SQ>& L) {
    constexpr int LDA = 16 * SQ + 8;
    constexpr int T1W = MX_N1 / 16 / 4;   // 6 first-layer column tiles per wave
    auto& As = L.As;
    auto& Hs = L.Hs;
    auto& part = L.part;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int64_t m0 = (int64_t)blk * 16;
    const int S = io.S;

    // ---- loads, oldest first = needed first (s_waitcnt counts in issue order): the state rows of the LayerNorm, then
    // every weight fragment of the launch; nothing below depends on anything but kernel arguments ----
    // (the bias / LayerNorm / V-head vectors too: a load next to its use sits behind the s_waitcnt vmcnt(0) that hipcc
    // puts at every divergent join, i.e. each one would cost a full memory latency AND drain the fragment loads)
    float x[SQ], lnw[SQ], lnb[SQ];
    {
        const int64_t m = m0 + 4 * wave + g;
#pragma unroll
        for (int c = 0; c < SQ; ++c) {
            const int col = li + 16 * c, cc = col < S ? col : S - 1;
            x[c] = (MX_ABL & 8) ? (float)col : io.s[(m < io.M ? m : io.M - 1) * io.s_ld + cc];
            lnw[c] = io.ln_w[cc];
            lnb[c] = io.ln_b[cc];
        }
    }
    float b1v[T1W], b2e[J], wv[16];
#pragma unroll
    for (int i = 0; i < T1W; ++i) b1v[i] = io.b1[16 * (T1W * wave + i) + li];
#pragma unroll
    for (int j = 0; j < J; ++j) b2e[j] = io.b2[j * MX_EM + 16 * wave + li];
    const float bfe = io.bf2[16 * wave + li];
#pragma unroll
    for (int k = 0; k < 16; ++k) wv[k] = io.wV2[16 * g + k];
    const float bV2 = io.bV2[0];
    f32x4 B1[T1W][SQ], B2[J][MX_KQ2], Bf[MX_KQ2];
#pragma unroll
    for (int i = 0; i < T1W; ++i) mx_load_frags<SQ, true>(B1[i], io.W1, S, 16 * (T1W * wave + i) + li, g);
    auto load_second = [&]() {
#pragma unroll
        for (int j = 0; j < J; ++j) mx_load_frags<MX_KQ2, false>(B2[j], io.W2, MX_HH, j * MX_EM + 16 * wave + li, g);
        mx_load_frags<MX_KQ2, false>(Bf, io.Wf2, MX_HH, 16 * wave + li, g);
    };
    if (!LATE2) load_second();
    // this lane's rows of q (rows 4g..4g+3 of the tile), for the tail
    float qv[4][J];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + 4 * g + r;
        const int64_t mc = m < io.M ? m : io.M - 1;   // clamped: rows past M are computed on the last row, never stored
#pragma unroll
        for (int j = 0; j < J; ++j) qv[r][j] = io.q[mc * J + j];
    }

    // ---- phase 0: LayerNorm (networks.py:283), wave w: rows 4w..4w+3, lane (g = row within the wave, li = column mod 16)
    {
        const int row = 4 * wave + g;
        const int64_t m = m0 + row;
        float sum = 0.0f;
#pragma unroll
        for (int c = 0; c < SQ; ++c) {
            x[c] = (li + 16 * c < S) ? x[c] : 0.0f;
            sum += x[c];
        }
        const float mean = mx_sum16(sum) / (float)S;
        float sq = 0.0f;
#pragma unroll
        for (int c = 0; c < SQ; ++c) {
            const float d = (li + 16 * c < S) ? x[c] - mean : 0.0f;
            sq = fmaf(d, d, sq);
        }
        const float rstd = rsqrtf(mx_sum16(sq) / (float)S + io.ln_eps);
#pragma unroll
        for (int c = 0; c < SQ; ++c) {
            const int col = li + 16 * c;
            const bool live = col < S;
            const float xh = live ? (x[c] - mean) * rstd : 0.0f;
            const float v = live ? xh * lnw[c] + lnb[c] : 0.0f;               // pad columns: 0 (they meet zero fragments)
            As[row * LDA + col] = v;
            if (SAVE && live && m < io.M) {
                io.sn[m * S + col] = v;
                io.xhat[m * S + col] = xh;
            }
        }
    }
    __syncthreads();

    // ---- phase 1: merged first layer (networks.py:285-299), this wave's 6 column tiles ----
    {
        f32x4 acc[T1W];
#pragma unroll
        for (int i = 0; i < T1W; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* ap = As + li * LDA + 4 * g;
#pragma unroll
        for (int Q = 0; Q < ((MX_ABL & 2) ? 0 : SQ); ++Q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ap + 16 * Q);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int i = 0; i < T1W; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj], B1[i][Q][jj], acc[i], 0, 0, 0);
        }
        if (LATE2) load_second();
#pragma unroll
        for (int i = 0; i < T1W; ++i) {
            const int col = 16 * (T1W * wave + i) + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * g + r;
                float v = acc[i][r] + b1v[i];
                v = (col < MX_RELU) ? fmaxf(v, 0.0f) : v;
                Hs[row * MX_LDH + col] = v;
                if (SAVE && m0 + row < io.M) io.act[(m0 + row) * MX_N1 + col] = v;
            }
        }
    }
    __syncthreads();

    // ---- phase 2: second layers for embed block `wave`, tail (networks.py:301-310) ----
    f32x4 acc2[J], accf;
    mx_second_layers<J>(Hs, B2, Bf, acc2, accf, li, g);
    const int e = 16 * wave + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        float hid = mx_clamp(Hs[row * MX_LDH + MX_RELU + e], -5.0f, 5.0f);
#pragma unroll
        for (int j = 0; j < J; ++j) hid = fmaf(qv[r][j], mx_clamp(acc2[j][r] + b2e[j], 0.0f, 5.0f), hid);   // bmm(q, w1) + b1
        const float h = hid > 0.0f ? hid : expm1f(hid);                                                       // F.elu
        const float t = mx_sum16(h * mx_clamp(accf[r] + bfe, 0.0f, 5.0f));                                    // bmm(hidden, w_final)
        if (li == 0) part[wave][row] = t;
    }
    const float v_raw = (wave == 0) ? mx_v_raw(Hs, wv, bV2, li, g) : 0.0f;
    __syncthreads();
    if (wave == 0 && g == 0) {
        const int64_t m = m0 + li;
        if (m < io.M) io.y[m] = ((part[0][li] + part[1][li]) + (part[2][li] + part[3][li])) + mx_clamp(v_raw, -5.0f, 5.0f);
    }
}

template <int J, int SQ, bool SAVE>
__global__ void __launch_bounds__(256) mixer_fused_forward_kernel(const macjd_mixerf_io io) {
    __shared__ MixerFwdLds<SQ> L;
    mixer_fused_forward_body<J, SQ, SAVE>(io, blockIdx.x, L);
}

// blockIdx.y = 0: the mixer whose activations are saved for a backward (io_a), 1: the plain one (io_b)
template <int J, int SQ>
__global__ void __launch_bounds__(256, (J <= 3) ? 2 : 1) mixer_fused_forward_pair_kernel(const macjd_mixerf_io io_a, const macjd_mixerf_io io_b) {
    __shared__ MixerFwdLds<SQ> L;
    if (blockIdx.y == 0) mixer_fused_forward_body<J, SQ, true, (J <= 3)>(io_a, blockIdx.x, L);
    else mixer_fused_forward_body<J, SQ, false, (J <= 3)>(io_b, blockIdx.x, L);
}

// Backward (see the header): recompute the second layers from `act`, tail gradients, transposed second layers.
// TD = the gradient of the learner's TD loss w.r.t. this mixer's output is formed HERE instead of being read from io.gy
// (macjd_mixer_fused_backward_td): row m = (b, t) gets scale * mask * (y - (r + gamma (1 - term) tq)) for t < Tm1 and 0
// otherwise — td_loss_kernel's own expression, with scale = 2 / *tot_m from a launch that summed the batch's mask earlier
// (macjd_td_mask_sum: it needs the gathered batch only, so it runs long before).  Five loads per row, no reduction: the
// loss launch leaves the update's serial chain (its logged sums are computed off the chain by macjd_td_loss).
template <int J, bool TD = false>
__global__ void __launch_bounds__(256) mixer_fused_backward_kernel(const macjd_mixerf_io io, const macjd_tdloss_io td,
                                                                    const float* __restrict__ tot_m) {
    constexpr int LDG = J * MX_EM + 8;      // pitch of g_w1raw in LDS (= 8 mod 16)
    constexpr int LDF = MX_EM + 8;
    constexpr int KQ1 = J * MX_EM / 16;     // quads of the transposed hyper_w_1.2 product
    constexpr int KQF = MX_EM / 16;
    __shared__ __attribute__((aligned(16))) float Hs[16 * MX_LDH];
    __shared__ __attribute__((aligned(16))) float G1[16 * LDG];
    __shared__ __attribute__((aligned(16))) float Gf[16 * LDF];
    __shared__ float gq_part[4][16][J];
    __shared__ float gv_s[16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, g = lane >> 4;
    if (TD && td.stats && blockIdx.x == gridDim.x - 1) {   // the extra workgroup: the loss's logged sums (macjd_tdloss.h)
        td_loss_sums(td, false);
        return;
    }
    const int64_t m0 = (int64_t)blockIdx.x * 16;

    // ---- weight fragments, requested up front: forward orientation for the recomputation ...
    f32x4 B2[J][MX_KQ2], Bf[MX_KQ2];
#pragma unroll
    for (int j = 0; j < J; ++j) mx_load_frags<MX_KQ2, false>(B2[j], io.W2, MX_HH, j * MX_EM + 16 * wave + li, g);
    mx_load_frags<MX_KQ2, false>(Bf, io.Wf2, MX_HH, 16 * wave + li, g);
    // ... and transposed for the input gradients of the second layers: this wave's output columns are the hidden units
    // n = 16 (2 wave + tt) + li; B[k][n] = W2[k][n] with k = 16 Q + 4 g + jj the row (a w1_raw column): four strided
    // dwords per quad, each a coalesced 64-byte row piece over li
    // (J = 6: both fragment sets together would need more than the 512 registers; the transposed set is then
    // requested after the recomputation, when the forward set is dead)
    constexpr bool EARLY_D = (J <= 3);
    f32x4 D1[2][KQ1], Df[2][KQF];
    auto load_transposed = [&]() {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int n = 16 * (2 * wave + tt) + li;
#pragma unroll
            for (int Q = 0; Q < KQ1; ++Q)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) D1[tt][Q][jj] = io.W2[(int64_t)(16 * Q + 4 * g + jj) * MX_HH + n];
#pragma unroll
            for (int Q = 0; Q < KQF; ++Q)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) Df[tt][Q][jj] = io.Wf2[(int64_t)(16 * Q + 4 * g + jj) * MX_HH + n];
        }
    };
    if (EARLY_D) load_transposed();
    float b2e[J], wv[16], wvo[4];
#pragma unroll
    for (int j = 0; j < J; ++j) b2e[j] = io.b2[j * MX_EM + 16 * wave + li];
    const float bfe = io.bf2[16 * wave + li];
#pragma unroll
    for (int k = 0; k < 16; ++k) wv[k] = io.wV2[16 * g + k];
#pragma unroll
    for (int i = 0; i < 4; ++i) wvo[i] = io.wV2[(threadIdx.x + 256 * i) & (MX_EM - 1)];   // for the V head's outer product
    const float bV2 = io.bV2[0];
    auto load_gy = [&](const int64_t mc) -> float {
        if constexpr (TD) {
            const int cols = (int)td.gy_cols;
            const int b = (int)(mc / cols), t = (int)(mc - (int64_t)b * cols);
            const int tc = t < td.Tm1 ? t : td.Tm1 - 1;     // clamped: no branch around the loads
            const float term = td.terminated[b * td.t_sb + tc * td.t_st] ? 1.0f : 0.0f;
            const float mk = td.filled[b * td.f_sb + tc * td.f_st] ? 1.0f : 0.0f;
            const float target = td.reward[b * td.r_sb + tc * td.r_st] + td.gamma * (1.0f - term) * td.tq[b * td.tq_sb + tc];
            const float scale = 2.0f / tot_m[0];
            const float gv = scale * mk * (td.y[b * td.y_sb + tc] - target);
            return t < td.Tm1 ? gv : 0.0f;
        } else {
            return io.gy[mc];
        }
    };
    const float gy_li = load_gy((m0 + li < io.M) ? m0 + li : io.M - 1);
    float qv[4][J], gyv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + 4 * g + r;
        const int64_t mc = m < io.M ? m : io.M - 1;   // clamped (no branch around a load); rows past M are never stored
        gyv[r] = load_gy(mc);
#pragma unroll
        for (int j = 0; j < J; ++j) qv[r][j] = io.q[mc * J + j];
    }
    // ---- the saved first-layer output of the 16 rows -> LDS (float4 pieces; rows past M: zeros) ----
    for (int idx = threadIdx.x; idx < 16 * (MX_N1 / 4); idx += 256) {
        const int row = idx / (MX_N1 / 4), c4 = idx - row * (MX_N1 / 4);
        const int64_t mc = (m0 + row < io.M) ? m0 + row : io.M - 1;
        *reinterpret_cast<f32x4*>(Hs + row * MX_LDH + 4 * c4) = *reinterpret_cast<const f32x4*>(io.act + mc * MX_N1 + 4 * c4);
    }
    __syncthreads();

    // ---- recompute w1_raw / wf_raw tiles of embed block `wave` and the tail, then its gradients ----
    f32x4 acc2[J], accf;
    mx_second_layers<J>(Hs, B2, Bf, acc2, accf, li, g);
    const int e = 16 * wave + li;
    const float v_raw = (wave == 0) ? mx_v_raw(Hs, wv, bV2, li, g) : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        const int64_t m = m0 + row;
        const float b1r = Hs[row * MX_LDH + MX_RELU + e];
        float hid = mx_clamp(b1r, -5.0f, 5.0f);
        float w1r[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            w1r[j] = acc2[j][r] + b2e[j];
            hid = fmaf(qv[r][j], mx_clamp(w1r[j], 0.0f, 5.0f), hid);
        }
        const float h = hid > 0.0f ? hid : expm1f(hid);
        const float wfr = accf[r] + bfe;
        const float ghid = gyv[r] * mx_clamp(wfr, 0.0f, 5.0f) * (hid > 0.0f ? 1.0f : h + 1.0f);   // ELU'(x) = elu(x) + 1, x <= 0
        const float gwf = (wfr >= 0.0f && wfr <= 5.0f) ? gyv[r] * h : 0.0f;                        // clamp passes inside [lo, hi]
        const float gb1 = (b1r >= -5.0f && b1r <= 5.0f) ? ghid : 0.0f;
        Gf[row * LDF + e] = gwf;
        if (m < io.M) {
            io.g_wfraw[m * MX_EM + e] = gwf;
            io.gout1[m * MX_N1 + MX_RELU + e] = gb1;
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const float gw = (w1r[j] >= 0.0f && w1r[j] <= 5.0f) ? ghid * qv[r][j] : 0.0f;
            G1[row * LDG + j * MX_EM + e] = gw;
            if (m < io.M) io.g_w1raw[m * (J * MX_EM) + j * MX_EM + e] = gw;
            const float t = mx_sum16(ghid * mx_clamp(w1r[j], 0.0f, 5.0f));
            if (li == 0) gq_part[wave][row][j] = t;
        }
    }
    if (!EARLY_D) load_transposed();
    if (wave == 0 && g == 0) {   // v = clamp(v_raw, -5, 5): row li
        const int64_t m = m0 + li;
        const float gv = (v_raw >= -5.0f && v_raw <= 5.0f) ? gy_li : 0.0f;
        gv_s[li] = gv;
        if (m < io.M) io.g_v[m] = gv;
    }
    __syncthreads();
    // dL/dq: the four embed blocks' partial sums in fixed order
    for (int idx = threadIdx.x; idx < 16 * J; idx += 256) {
        const int row = idx / J, j = idx - row * J;
        if (m0 + row < io.M)
            io.gq[(m0 + row) * J + j] = (gq_part[0][row][j] + gq_part[1][row][j]) + (gq_part[2][row][j] + gq_part[3][row][j]);
    }

    // ---- input gradients of the second layers, masked by the first layer's ReLU: columns [0, 2 Hh) of gout1 ----
    {
        f32x4 a1[2], af[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) { a1[tt] = f32x4{0.f, 0.f, 0.f, 0.f}; af[tt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const float* gp = G1 + li * LDG + 4 * g;
        const float* fp = Gf + li * LDF + 4 * g;
#pragma unroll
        for (int Q = 0; Q < KQ1; ++Q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(gp + 16 * Q);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) a1[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj], D1[tt][Q][jj], a1[tt], 0, 0, 0);
        }
#pragma unroll
        for (int Q = 0; Q < KQF; ++Q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(fp + 16 * Q);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) af[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj], Df[tt][Q][jj], af[tt], 0, 0, 0);
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int n = 16 * (2 * wave + tt) + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * g + r;
                const int64_t m = m0 + row;
                if (m < io.M) {
                    io.gout1[m * MX_N1 + n] = (Hs[row * MX_LDH + n] > 0.0f) ? a1[tt][r] : 0.0f;
                    io.gout1[m * MX_N1 + MX_HH + n] = (Hs[row * MX_LDH + MX_HH + n] > 0.0f) ? af[tt][r] : 0.0f;
                }
            }
        }
    }
    // the V head's one-output second layer: outer product g_v wV2, masked: columns [2 Hh, 2 Hh + Em)
#pragma unroll
    for (int i = 0; i < 16 * MX_EM / 256; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int row = idx / MX_EM, k = idx - row * MX_EM;
        if (m0 + row < io.M)
            io.gout1[(m0 + row) * MX_N1 + 2 * MX_HH + k] = (Hs[row * MX_LDH + 2 * MX_HH + k] > 0.0f) ? gv_s[row] * wvo[i] : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Wide scenarios (J = 12: BASELINE.json's 12 jammers / 16 radars, S = 184).  Same geometry and the same expressions as
// the kernels above, but the weight fragments of a whole layer no longer fit a lane's registers (first layer: 6 tiles x
// 12 quads = 288 VGPRs; hyper_w_1's second layer: 12 tiles x 8 quads = 384), so both layers run in PASSES whose
// fragments are loaded one pass ahead: the first layer two column tiles at a time, the J tiles of w1_raw PJ agents at
// a time.  The tail's fmaf chain over the agents runs in agent order across the passes, i.e. it is the chain of
// mixer_tail_kernel / the narrow kernels.
template <int PJ>
__device__ __forceinline__ void mx_load_pass(f32x4 (&dst)[PJ][MX_KQ2], const float* __restrict__ W2, int j0, int wave, int li, int g) {
#pragma unroll
    for (int jj = 0; jj < PJ; ++jj) mx_load_frags<MX_KQ2, false>(dst[jj], W2, MX_HH, (j0 + jj) * MX_EM + 16 * wave + li, g);
}

template <int PJ>
__device__ __forceinline__ void mx_pass_tiles(const float* __restrict__ Hs, const f32x4 (&B2)[PJ][MX_KQ2], f32x4 (&acc2)[PJ],
                                              int li, int g) {
#pragma unroll
    for (int j = 0; j < PJ; ++j) acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* a1p = Hs + li * MX_LDH + 4 * g;            // h_w1: columns [0, Hh)
#pragma unroll
    for (int Q = 0; Q < MX_KQ2; ++Q) {
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(a1p + 16 * Q);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int j = 0; j < PJ; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[jj], B2[j][Q][jj], acc2[j], 0, 0, 0);
    }
}

__device__ __forceinline__ void mx_wf_tile(const float* __restrict__ Hs, const f32x4 (&Bf)[MX_KQ2], f32x4& accf, int li, int g) {
    accf = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* afp = Hs + li * MX_LDH + 4 * g + MX_HH;    // h_wf: columns [Hh, 2 Hh)
#pragma unroll
    for (int Q = 0; Q < MX_KQ2; ++Q) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(afp + 16 * Q);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) accf = __builtin_amdgcn_mfma_f32_16x16x4f32(af[jj], Bf[Q][jj], accf, 0, 0, 0);
    }
}

template <int J, int SQ, bool SAVE, int PJ>
__device__ __forceinline__ void mixer_fused_forward_wide_body(const macjd_mixerf_io& io, const int blk, MixerFwdLds<SQ>& L) {
    static_assert(J % PJ == 0, "whole passes");
    constexpr int LDA = 16 * SQ + 8;
    constexpr int T1W = MX_N1 / 16 / 4;   // 6 first-layer column tiles per wave
    constexpr int P1W = 2;                // ... two per pass
    constexpr int NP = J / PJ;
    auto& As = L.As;
    auto& Hs = L.Hs;
    auto& part = L.part;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int64_t m0 = (int64_t)blk * 16;
    const int S = io.S;

    float x[SQ], lnw[SQ], lnb[SQ];
    {
        const int64_t m = m0 + 4 * wave + g;
#pragma unroll
        for (int c = 0; c < SQ; ++c) {
            const int col = li + 16 * c, cc = col < S ? col : S - 1;
            x[c] = io.s[(m < io.M ? m : io.M - 1) * io.s_ld + cc];
            lnw[c] = io.ln_w[cc];
            lnb[c] = io.ln_b[cc];
        }
    }
    float b1v[T1W], b2e[J], wv[16];
#pragma unroll
    for (int i = 0; i < T1W; ++i) b1v[i] = io.b1[16 * (T1W * wave + i) + li];
#pragma unroll
    for (int j = 0; j < J; ++j) b2e[j] = io.b2[j * MX_EM + 16 * wave + li];
    const float bfe = io.bf2[16 * wave + li];
#pragma unroll
    for (int k = 0; k < 16; ++k) wv[k] = io.wV2[16 * g + k];
    const float bV2 = io.bV2[0];
    // first-layer fragments of passes 0 and 1 (the third pass re-uses the first buffer)
    f32x4 B1a[P1W][SQ], B1b[P1W][SQ];
#pragma unroll
    for (int i = 0; i < P1W; ++i) mx_load_frags<SQ, true>(B1a[i], io.W1, S, 16 * (T1W * wave + i) + li, g);
#pragma unroll
    for (int i = 0; i < P1W; ++i) mx_load_frags<SQ, true>(B1b[i], io.W1, S, 16 * (T1W * wave + P1W + i) + li, g);
    float qv[4][J];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + 4 * g + r;
        const int64_t mc = m < io.M ? m : io.M - 1;
#pragma unroll
        for (int j = 0; j < J; ++j) qv[r][j] = io.q[mc * J + j];
    }

    // ---- phase 0: LayerNorm ----
    {
        const int row = 4 * wave + g;
        const int64_t m = m0 + row;
        float sum = 0.0f;
#pragma unroll
        for (int c = 0; c < SQ; ++c) {
            x[c] = (li + 16 * c < S) ? x[c] : 0.0f;
            sum += x[c];
        }
        const float mean = mx_sum16(sum) / (float)S;
        float sq = 0.0f;
#pragma unroll
        for (int c = 0; c < SQ; ++c) {
            const float d = (li + 16 * c < S) ? x[c] - mean : 0.0f;
            sq = fmaf(d, d, sq);
        }
        const float rstd = rsqrtf(mx_sum16(sq) / (float)S + io.ln_eps);
#pragma unroll
        for (int c = 0; c < SQ; ++c) {
            const int col = li + 16 * c;
            const bool live = col < S;
            const float xh = live ? (x[c] - mean) * rstd : 0.0f;
            const float v = live ? xh * lnw[c] + lnb[c] : 0.0f;
            As[row * LDA + col] = v;
            if (SAVE && live && m < io.M) {
                io.sn[m * S + col] = v;
                io.xhat[m * S + col] = xh;
            }
        }
    }
    __syncthreads();

    // ---- phase 1: merged first layer, this wave's 6 column tiles in three passes of two ----
    auto first_layer_pass = [&](const f32x4 (&B1)[P1W][SQ], int pass) {
        f32x4 acc[P1W];
#pragma unroll
        for (int i = 0; i < P1W; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* ap = As + li * LDA + 4 * g;
#pragma unroll
        for (int Q = 0; Q < SQ; ++Q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ap + 16 * Q);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int i = 0; i < P1W; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj], B1[i][Q][jj], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < P1W; ++i) {
            const int t = P1W * pass + i;
            const int col = 16 * (T1W * wave + t) + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * g + r;
                float v = acc[i][r] + b1v[t];
                v = (col < MX_RELU) ? fmaxf(v, 0.0f) : v;
                Hs[row * MX_LDH + col] = v;
                if (SAVE && m0 + row < io.M) io.act[(m0 + row) * MX_N1 + col] = v;
            }
        }
    };
    first_layer_pass(B1a, 0);
#pragma unroll
    for (int i = 0; i < P1W; ++i) mx_load_frags<SQ, true>(B1a[i], io.W1, S, 16 * (T1W * wave + 2 * P1W + i) + li, g);
    // second-layer fragments of the first agent pass and of w_final: in flight under the rest of phase 1
    f32x4 B2a[PJ][MX_KQ2], B2b[PJ][MX_KQ2], Bf[MX_KQ2];
    mx_load_pass<PJ>(B2a, io.W2, 0, wave, li, g);
    mx_load_frags<MX_KQ2, false>(Bf, io.Wf2, MX_HH, 16 * wave + li, g);
    first_layer_pass(B1b, 1);
    first_layer_pass(B1a, 2);
    __syncthreads();

    // ---- phase 2: second layers for embed block `wave` in NP agent passes, tail ----
    const int e = 16 * wave + li;
    float hid[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) hid[r] = mx_clamp(Hs[(4 * g + r) * MX_LDH + MX_RELU + e], -5.0f, 5.0f);
    f32x4 accf;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        f32x4 acc2[PJ];
        if (p + 1 < NP) {
            if (p & 1) mx_load_pass<PJ>(B2a, io.W2, (p + 1) * PJ, wave, li, g);
            else mx_load_pass<PJ>(B2b, io.W2, (p + 1) * PJ, wave, li, g);
        }
        if (p & 1) mx_pass_tiles<PJ>(Hs, B2b, acc2, li, g);
        else mx_pass_tiles<PJ>(Hs, B2a, acc2, li, g);
        if (p == 0) mx_wf_tile(Hs, Bf, accf, li, g);
#pragma unroll
        for (int jj = 0; jj < PJ; ++jj) {
            const int j = p * PJ + jj;
#pragma unroll
            for (int r = 0; r < 4; ++r) hid[r] = fmaf(qv[r][j], mx_clamp(acc2[jj][r] + b2e[j], 0.0f, 5.0f), hid[r]);   // bmm(q, w1) + b1
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        const float h = hid[r] > 0.0f ? hid[r] : expm1f(hid[r]);                                   // F.elu
        const float t = mx_sum16(h * mx_clamp(accf[r] + bfe, 0.0f, 5.0f));                         // bmm(hidden, w_final)
        if (li == 0) part[wave][row] = t;
    }
    const float v_raw = (wave == 0) ? mx_v_raw(Hs, wv, bV2, li, g) : 0.0f;
    __syncthreads();
    if (wave == 0 && g == 0) {
        const int64_t m = m0 + li;
        if (m < io.M) io.y[m] = ((part[0][li] + part[1][li]) + (part[2][li] + part[3][li])) + mx_clamp(v_raw, -5.0f, 5.0f);
    }
}

template <int J, int SQ, bool SAVE, int PJ>
__global__ void __launch_bounds__(256) mixer_fused_forward_wide_kernel(const macjd_mixerf_io io) {
    __shared__ MixerFwdLds<SQ> L;
    mixer_fused_forward_wide_body<J, SQ, SAVE, PJ>(io, blockIdx.x, L);
}

template <int J, int SQ, int PJ>
__global__ void __launch_bounds__(256) mixer_fused_forward_wide_pair_kernel(const macjd_mixerf_io io_a, const macjd_mixerf_io io_b) {
    __shared__ MixerFwdLds<SQ> L;
    if (blockIdx.y == 0) mixer_fused_forward_wide_body<J, SQ, true, PJ>(io_a, blockIdx.x, L);
    else mixer_fused_forward_wide_body<J, SQ, false, PJ>(io_b, blockIdx.x, L);
}

template <int J, int PJ, bool TD = false>
__global__ void __launch_bounds__(256) mixer_fused_backward_wide_kernel(const macjd_mixerf_io io, const macjd_tdloss_io td,
                                                                         const float* __restrict__ tot_m) {
    static_assert(J % PJ == 0, "whole passes");
    constexpr int NP = J / PJ;
    constexpr int LDG = PJ * MX_EM + 8;     // pitch of ONE pass of g_w1raw in LDS (= 8 mod 16)
    constexpr int LDF = MX_EM + 8;
    constexpr int KQP = PJ * MX_EM / 16;    // quads of one pass of the transposed hyper_w_1.2 product
    constexpr int KQF = MX_EM / 16;
    __shared__ __attribute__((aligned(16))) float Hs[16 * MX_LDH];
    __shared__ __attribute__((aligned(16))) float G1[16 * LDG];
    __shared__ __attribute__((aligned(16))) float Gf[16 * LDF];
    __shared__ float gq_part[4][16][J];
    __shared__ float gv_s[16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, g = lane >> 4;
    if (TD && td.stats && blockIdx.x == gridDim.x - 1) {   // the extra workgroup: the loss's logged sums (macjd_tdloss.h)
        td_loss_sums(td, false);
        return;
    }
    const int64_t m0 = (int64_t)blockIdx.x * 16;

    f32x4 B2a[PJ][MX_KQ2], B2b[PJ][MX_KQ2], Bf[MX_KQ2];
    mx_load_pass<PJ>(B2a, io.W2, 0, wave, li, g);
    mx_load_frags<MX_KQ2, false>(Bf, io.Wf2, MX_HH, 16 * wave + li, g);
    float b2e[J], wv[16], wvo[4];
#pragma unroll
    for (int j = 0; j < J; ++j) b2e[j] = io.b2[j * MX_EM + 16 * wave + li];
    const float bfe = io.bf2[16 * wave + li];
#pragma unroll
    for (int k = 0; k < 16; ++k) wv[k] = io.wV2[16 * g + k];
#pragma unroll
    for (int i = 0; i < 4; ++i) wvo[i] = io.wV2[(threadIdx.x + 256 * i) & (MX_EM - 1)];
    const float bV2 = io.bV2[0];
    auto load_gy = [&](const int64_t mc) -> float {
        if constexpr (TD) {
            const int cols = (int)td.gy_cols;
            const int b = (int)(mc / cols), t = (int)(mc - (int64_t)b * cols);
            const int tc = t < td.Tm1 ? t : td.Tm1 - 1;
            const float term = td.terminated[b * td.t_sb + tc * td.t_st] ? 1.0f : 0.0f;
            const float mk = td.filled[b * td.f_sb + tc * td.f_st] ? 1.0f : 0.0f;
            const float target = td.reward[b * td.r_sb + tc * td.r_st] + td.gamma * (1.0f - term) * td.tq[b * td.tq_sb + tc];
            const float scale = 2.0f / tot_m[0];
            const float gv = scale * mk * (td.y[b * td.y_sb + tc] - target);
            return t < td.Tm1 ? gv : 0.0f;
        } else {
            return io.gy[mc];
        }
    };
    const float gy_li = load_gy((m0 + li < io.M) ? m0 + li : io.M - 1);
    float qv[4][J], gyv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + 4 * g + r;
        const int64_t mc = m < io.M ? m : io.M - 1;
        gyv[r] = load_gy(mc);
#pragma unroll
        for (int j = 0; j < J; ++j) qv[r][j] = io.q[mc * J + j];
    }
    for (int idx = threadIdx.x; idx < 16 * (MX_N1 / 4); idx += 256) {
        const int row = idx / (MX_N1 / 4), c4 = idx - row * (MX_N1 / 4);
        const int64_t mc = (m0 + row < io.M) ? m0 + row : io.M - 1;
        *reinterpret_cast<f32x4*>(Hs + row * MX_LDH + 4 * c4) = *reinterpret_cast<const f32x4*>(io.act + mc * MX_N1 + 4 * c4);
    }
    __syncthreads();

    // ---- sweep 1: recompute the w1_raw tiles of embed block `wave` pass by pass (kept: 4 J values per lane), hid ----
    const int e = 16 * wave + li;
    float w1r[J][4], hid[4], b1r[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        b1r[r] = Hs[(4 * g + r) * MX_LDH + MX_RELU + e];
        hid[r] = mx_clamp(b1r[r], -5.0f, 5.0f);
    }
    f32x4 accf;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        f32x4 acc2[PJ];
        if (p + 1 < NP) {
            if (p & 1) mx_load_pass<PJ>(B2a, io.W2, (p + 1) * PJ, wave, li, g);
            else mx_load_pass<PJ>(B2b, io.W2, (p + 1) * PJ, wave, li, g);
        }
        if (p & 1) mx_pass_tiles<PJ>(Hs, B2b, acc2, li, g);
        else mx_pass_tiles<PJ>(Hs, B2a, acc2, li, g);
        if (p == 0) mx_wf_tile(Hs, Bf, accf, li, g);
#pragma unroll
        for (int jj = 0; jj < PJ; ++jj) {
            const int j = p * PJ + jj;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                w1r[j][r] = acc2[jj][r] + b2e[j];
                hid[r] = fmaf(qv[r][j], mx_clamp(w1r[j][r], 0.0f, 5.0f), hid[r]);
            }
        }
    }
    const float v_raw = (wave == 0) ? mx_v_raw(Hs, wv, bV2, li, g) : 0.0f;
    float ghid[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        const int64_t m = m0 + row;
        const float h = hid[r] > 0.0f ? hid[r] : expm1f(hid[r]);
        const float wfr = accf[r] + bfe;
        ghid[r] = gyv[r] * mx_clamp(wfr, 0.0f, 5.0f) * (hid[r] > 0.0f ? 1.0f : h + 1.0f);
        const float gwf = (wfr >= 0.0f && wfr <= 5.0f) ? gyv[r] * h : 0.0f;
        const float gb1 = (b1r[r] >= -5.0f && b1r[r] <= 5.0f) ? ghid[r] : 0.0f;
        Gf[row * LDF + e] = gwf;
        if (m < io.M) {
            io.g_wfraw[m * MX_EM + e] = gwf;
            io.gout1[m * MX_N1 + MX_RELU + e] = gb1;
        }
    }
    if (wave == 0 && g == 0) {
        const int64_t m = m0 + li;
        const float gv = (v_raw >= -5.0f && v_raw <= 5.0f) ? gy_li : 0.0f;
        gv_s[li] = gv;
        if (m < io.M) io.g_v[m] = gv;
    }

    // ---- sweep 2: per agent pass, the tile gradients -> LDS / HBM, then that pass's share of the transposed product ----
    f32x4 a1[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) a1[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int p = 0; p < NP; ++p) {
        // this pass's transposed fragments: rows k = p PJ Em + 16 Q + 4 g + jj of W2, column n = 16 (2 wave + tt) + li
        f32x4 D1[2][KQP];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int n = 16 * (2 * wave + tt) + li;
#pragma unroll
            for (int Q = 0; Q < KQP; ++Q)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) D1[tt][Q][jj] = io.W2[(int64_t)(p * PJ * MX_EM + 16 * Q + 4 * g + jj) * MX_HH + n];
        }
        if (p > 0) __syncthreads();          // the previous pass's readers of G1 are done
#pragma unroll
        for (int jj = 0; jj < PJ; ++jj) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * g + r;
                const int64_t m = m0 + row;
                // (w1r is indexed with the run-time pass: selected through a compile-time scan so that it stays in registers)
                float w = 0.0f, qj = 0.0f;
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) {
                    w = (pp == p) ? w1r[pp * PJ + jj][r] : w;
                    qj = (pp == p) ? qv[r][pp * PJ + jj] : qj;
                }
                const int j = p * PJ + jj;
                const float gw = (w >= 0.0f && w <= 5.0f) ? ghid[r] * qj : 0.0f;
                G1[row * LDG + jj * MX_EM + e] = gw;
                if (m < io.M) io.g_w1raw[m * (J * MX_EM) + j * MX_EM + e] = gw;
                const float t = mx_sum16(ghid[r] * mx_clamp(w, 0.0f, 5.0f));
                if (li == 0) gq_part[wave][row][j] = t;
            }
        }
        __syncthreads();
        const float* gp = G1 + li * LDG + 4 * g;
#pragma unroll
        for (int Q = 0; Q < KQP; ++Q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(gp + 16 * Q);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) a1[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj], D1[tt][Q][jj], a1[tt], 0, 0, 0);
        }
    }
    // dL/dq: the four embed blocks' partial sums in fixed order (gq_part complete: barrier inside the last pass)
    for (int idx = threadIdx.x; idx < 16 * J; idx += 256) {
        const int row = idx / J, j = idx - row * J;
        if (m0 + row < io.M)
            io.gq[(m0 + row) * J + j] = (gq_part[0][row][j] + gq_part[1][row][j]) + (gq_part[2][row][j] + gq_part[3][row][j]);
    }
    {
        f32x4 Df[2][KQF], af[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int n = 16 * (2 * wave + tt) + li;
            af[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int Q = 0; Q < KQF; ++Q)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) Df[tt][Q][jj] = io.Wf2[(int64_t)(16 * Q + 4 * g + jj) * MX_HH + n];
        }
        const float* fp = Gf + li * LDF + 4 * g;
#pragma unroll
        for (int Q = 0; Q < KQF; ++Q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(fp + 16 * Q);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) af[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj], Df[tt][Q][jj], af[tt], 0, 0, 0);
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int n = 16 * (2 * wave + tt) + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * g + r;
                const int64_t m = m0 + row;
                if (m < io.M) {
                    io.gout1[m * MX_N1 + n] = (Hs[row * MX_LDH + n] > 0.0f) ? a1[tt][r] : 0.0f;
                    io.gout1[m * MX_N1 + MX_HH + n] = (Hs[row * MX_LDH + MX_HH + n] > 0.0f) ? af[tt][r] : 0.0f;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16 * MX_EM / 256; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int row = idx / MX_EM, k = idx - row * MX_EM;
        if (m0 + row < io.M)
            io.gout1[(m0 + row) * MX_N1 + 2 * MX_HH + k] = (Hs[row * MX_LDH + 2 * MX_HH + k] > 0.0f) ? gv_s[row] * wvo[i] : 0.0f;
    }
}

static int mixerf_check(const macjd_mixerf_io* io, bool backward, bool gy_from_td = false) {
    if (!io) return set_err(MACJD_EINVAL, "%s", "macjd_mixer_fused: NULL io");
    if (!macjd_mixer_fused_supported(io->J, io->S, io->Hh, io->Em))
        return set_err(MACJD_EUNSUPPORTED, "%s", "macjd_mixer_fused: unsupported J / S / Hh / Em (see include/macjd_nets.h)");
    if (io->M < 0 || !io->q || !io->W2 || !io->b2 || !io->Wf2 || !io->bf2 || !io->wV2 || !io->bV2)
        return set_err(MACJD_EINVAL, "%s", "macjd_mixer_fused: bad M / NULL input");
    if (!backward) {
        if (!io->s || io->s_ld < io->S || !io->ln_w || !io->ln_b || !io->W1 || !io->b1 || !io->y)
            return set_err(MACJD_EINVAL, "%s", "macjd_mixer_fused_forward: NULL input / output or bad s_ld");
        if (io->save && (!io->sn || !io->xhat || !io->act))
            return set_err(MACJD_EINVAL, "%s", "macjd_mixer_fused_forward: save needs sn / xhat / act");
    } else if (!io->act || (!io->gy && !gy_from_td) || !io->gq || !io->gout1 || !io->g_w1raw || !io->g_wfraw || !io->g_v) {
        return set_err(MACJD_EINVAL, "%s", "macjd_mixer_fused_backward: NULL input / output");
    }
    if (((uintptr_t)io->act) & 15)   // (weights may sit anywhere in a flat parameter vector: their fragment loads assume 4 bytes)
        return set_err(MACJD_EINVAL, "%s", "macjd_mixer_fused: act must be 16-byte aligned");
    return MACJD_OK;
}

}  // namespace macjd

extern "C" int macjd_mixer_fused_supported(int32_t J, int32_t S, int32_t Hh, int32_t Em) {
    if (Hh != macjd::MX_HH || Em != macjd::MX_EM || S < 1) return 0;
    return (J == 2 && S <= 32) || (J == 3 && S <= 48) || (J == 6 && S <= 96) || (J == 12 && S <= 192);
}

extern "C" int macjd_mixer_fused_forward(const macjd_mixerf_io* io, void* hip_stream) {
    using namespace macjd;
    const int rc = mixerf_check(io, false);
    if (rc != MACJD_OK) return rc;
    if (io->M == 0) return MACJD_OK;
    const dim3 grid((unsigned)((io->M + 15) / 16)), block(256);
    hipStream_t s = (hipStream_t)hip_stream;
#define MACJD_MXF(J_, SQ_)                                                                                     \
    do {                                                                                                       \
        if (io->save) hipLaunchKernelGGL((mixer_fused_forward_kernel<J_, SQ_, true>), grid, block, 0, s, *io);  \
        else hipLaunchKernelGGL((mixer_fused_forward_kernel<J_, SQ_, false>), grid, block, 0, s, *io);          \
    } while (0)
    if (io->J == 2) MACJD_MXF(2, 2);
    else if (io->J == 3) MACJD_MXF(3, 3);
    else if (io->J == 6) MACJD_MXF(6, 6);
    else if (io->save) hipLaunchKernelGGL((mixer_fused_forward_wide_kernel<12, 12, true, 4>), grid, block, 0, s, *io);
    else hipLaunchKernelGGL((mixer_fused_forward_wide_kernel<12, 12, false, 4>), grid, block, 0, s, *io);
#undef MACJD_MXF
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "macjd_mixer_fused_forward: %s", hipGetErrorString(err));
    return MACJD_OK;
}

extern "C" int macjd_mixer_fused_forward_pair(const macjd_mixerf_io* saved, const macjd_mixerf_io* plain, void* hip_stream) {
    using namespace macjd;
    int rc = mixerf_check(saved, false);
    if (rc != MACJD_OK) return rc;
    rc = mixerf_check(plain, false);
    if (rc != MACJD_OK) return rc;
    if (!saved->save || plain->save) return set_err(MACJD_EINVAL, "%s", "macjd_mixer_fused_forward_pair: the first mixer saves, the second does not");
    if (saved->J != plain->J || saved->S != plain->S || saved->M != plain->M)
        return set_err(MACJD_EINVAL, "%s", "macjd_mixer_fused_forward_pair: the two mixers differ in J / S / M");
    if (saved->M == 0) return MACJD_OK;
    const dim3 grid((unsigned)((saved->M + 15) / 16), 2), block(256);
    hipStream_t s = (hipStream_t)hip_stream;
    if (saved->J == 2) hipLaunchKernelGGL((mixer_fused_forward_pair_kernel<2, 2>), grid, block, 0, s, *saved, *plain);
    else if (saved->J == 3) hipLaunchKernelGGL((mixer_fused_forward_pair_kernel<3, 3>), grid, block, 0, s, *saved, *plain);
    else if (saved->J == 6) hipLaunchKernelGGL((mixer_fused_forward_pair_kernel<6, 6>), grid, block, 0, s, *saved, *plain);
    else hipLaunchKernelGGL((mixer_fused_forward_wide_pair_kernel<12, 12, 4>), grid, block, 0, s, *saved, *plain);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "macjd_mixer_fused_forward_pair: %s", hipGetErrorString(err));
    return MACJD_OK;
}

extern "C" int macjd_mixer_fused_backward(const macjd_mixerf_io* io, void* hip_stream) {
    using namespace macjd;
    const int rc = mixerf_check(io, true);
    if (rc != MACJD_OK) return rc;
    if (io->M == 0) return MACJD_OK;
    const dim3 grid((unsigned)((io->M + 15) / 16)), block(256);
    hipStream_t s = (hipStream_t)hip_stream;
    const macjd_tdloss_io none{};
    if (io->J == 2) hipLaunchKernelGGL((mixer_fused_backward_kernel<2>), grid, block, 0, s, *io, none, nullptr);
    else if (io->J == 3) hipLaunchKernelGGL((mixer_fused_backward_kernel<3>), grid, block, 0, s, *io, none, nullptr);
    else if (io->J == 6) hipLaunchKernelGGL((mixer_fused_backward_kernel<6>), grid, block, 0, s, *io, none, nullptr);
    else hipLaunchKernelGGL((mixer_fused_backward_wide_kernel<12, 4>), grid, block, 0, s, *io, none, nullptr);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "macjd_mixer_fused_backward: %s", hipGetErrorString(err));
    return MACJD_OK;
}

extern "C" int macjd_mixer_fused_backward_td(const macjd_mixerf_io* io, const macjd_tdloss_io* td, const float* tot_m,
                                             void* hip_stream) {
    using namespace macjd;
    const int rc = mixerf_check(io, true, true);
    if (rc != MACJD_OK) return rc;
    if (!td || !tot_m || td->B < 1 || td->Tm1 < 1 || !td->y || !td->tq || !td->reward || !td->terminated || !td->filled)
        return set_err(MACJD_EINVAL, "%s", "macjd_mixer_fused_backward_td: bad TD-loss argument");
    if (td->gy_cols < td->Tm1 || (int64_t)td->B * td->gy_cols != io->M || td->y_sb < td->Tm1 || td->tq_sb < td->Tm1)
        return set_err(MACJD_EINVAL, "%s", "macjd_mixer_fused_backward_td: rows must be B x gy_cols with gy_cols >= Tm1");
    if (io->M == 0) return MACJD_OK;
    // td->stats != NULL: one workgroup more, which computes the loss's logged sums (stats[0..2]) beside the others
    const dim3 grid((unsigned)((io->M + 15) / 16) + (td->stats ? 1u : 0u)), block(256);
    hipStream_t s = (hipStream_t)hip_stream;
    if (io->J == 2) hipLaunchKernelGGL((mixer_fused_backward_kernel<2, true>), grid, block, 0, s, *io, *td, tot_m);
    else if (io->J == 3) hipLaunchKernelGGL((mixer_fused_backward_kernel<3, true>), grid, block, 0, s, *io, *td, tot_m);
    else if (io->J == 6) hipLaunchKernelGGL((mixer_fused_backward_kernel<6, true>), grid, block, 0, s, *io, *td, tot_m);
    else hipLaunchKernelGGL((mixer_fused_backward_wide_kernel<12, 4, true>), grid, block, 0, s, *io, *td, tot_m);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "macjd_mixer_fused_backward_td: %s", hipGetErrorString(err));
    return MACJD_OK;
}

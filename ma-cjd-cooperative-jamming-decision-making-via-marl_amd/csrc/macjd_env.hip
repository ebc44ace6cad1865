// macjd_env.hip — batched radar-jamming environment step for MI355X (gfx950, wave64).
//
// One lane = one environment.  The kernel is the batched form of
// ElectromagneticEnvironment.step (reference simulation/environment.py:221-477) with the helpers it
// calls (core/radar.py:67-82 detection probability, :90-119 SEARCH/TRACK FSM; core/jammer.py:56-98
// received jamming power).  All arithmetic is float64 in the reference's operation order (the TU is
// built with -ffp-contract=off: Python never fuses multiply-add), except the power /
// received-power numerator which follows NumPy-2's float32 weak-scalar promotion when the power
// action arrives as float32 (see include/macjd.h, MACJD_STEP_ARITH_F64).
//
// Data movement (HBM-bound streaming kernel; see DESIGN.md for the byte count):
//   * per-env inputs/outputs are addressed through caller-supplied element strides, so the runner
//     can keep actions agent-major ([J,E]: every load instruction is a fully coalesced 256-B row) while
//     the reference-shaped env-major layout ([E,J]) stays valid;
//   * per-radar constants that are indexed by the (compile-time unrolled) radar loop are read
//     through wave-uniform scalar loads from the scenario table;
//   * tables that are indexed by the lane's *chosen target radar* (jammer->radar path denominators,
//     D, Pn, receive gain) are staged once per workgroup into LDS and gathered with ds_read.
//
// The whole scenario is templated on (J, R) for the BASELINE.json configurations so that the
// suppression / deception accumulators live in registers; a generic (0,0) instantiation covers
// every other size up to MACJD_MAX_*.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "../../include/macjd.h"
#include "macjd_err.h"
#include "macjd_philox.h"

namespace macjd {

constexpr int MAXR = MACJD_MAX_RADARS;
constexpr int MAXJ = MACJD_MAX_JAMMERS;

// Device-resident scenario tables (one per scenario handle).
struct DevTables {
    int32_t R, J, episode_limit, pad;
    double rp_min, rp_max, pd_A, pd_c1, pd_denB;
    double GaPs[MAXR], Pn[MAXR], D[MAXR], pd_no[MAXR], rd_pen[MAXR], gr[MAXR];
    double pmin[MAXJ], pmax[MAXJ], gj[MAXJ];
    double denom[MAXJ * MAXR];   // packed [j*R + r]; negative = jammer sits on the radar (ignored)
    uint8_t flags[MAXJ * MAXR];  // packed [j*R + r]
    // ---- derived ON THE DEVICE when the scenario is created (scenario_derive_kernel, with the very device functions the
    // step kernels divide with), read by the REGULAR production variant (see env_step_kernel, REG) ----
    int32_t regular, pad2;       // host verdict: every table value in the range where the short division is IEEE division
    double rPn[MAXR];            // refined reciprocal of Pn
    double range_fd[MAXJ], r_range[MAXJ];      // (double)(float)(pmax - pmin) and its refined reciprocal
    double dsel[MAXJ * MAXR], rsel[MAXJ * MAXR];   // divisor of the received-power quotient as the step uses it (1.0 where
                                                   // it does not divide; the float32-rounded value where the division is
                                                   // float32) and its refined reciprocal
    uint8_t rflags[MAXJ * MAXR]; // flags | JR_RECORDABLE (denom >= 0) | JR_LIVE (denom > 1e-18)
};
constexpr uint8_t JR_RECORDABLE = 0x40, JR_LIVE = 0x80;

static thread_local char g_err[512] = "";
int set_err(int code, const char* fmt, const char* a) {
    snprintf(g_err, sizeof(g_err), fmt, a);
    return code;
}

// Process-wide switches of the env launches, read from the environment ONCE (macjd_reload_options re-reads them: tests
// and A/B runs that change a switch inside one process).
static EnvOptions g_env_options = {true, true, false};
static bool g_env_options_loaded = false;
static void load_env_options() {
    const char* a = getenv("MACJD_ENV_REGULAR");
    const char* b = getenv("MACJD_ENV_PD32");
    const char* c = getenv("MACJD_GRU_SCAN");
    g_env_options.regular = !(a && a[0] == '0');
    g_env_options.pd32 = !(b && b[0] == '0');
    g_env_options.gru_ksplit = c && !strcmp(c, "ksplit");
    g_env_options_loaded = true;
}
const EnvOptions& env_options() {
    if (!g_env_options_loaded) load_env_options();
    return g_env_options;
}

// ---- detection probability, core/radar.py:67-82 (constants A, c1, denB precomputed on the host) -------------------
// The two divisions of the formula are done with the hardware's own IEEE division algorithm MINUS its scaling /
// special-case wrapper (v_div_scale x2, v_div_fmas' scale step, v_div_fixup): refined reciprocal r of the divisor (v_rcp +
// two Newton steps), q = n r, q + r (n - d q).  That is bit-for-bit what `/` compiles to whenever no operand scaling
// is needed, which holds here by construction: the divisor of B is the scenario constant denB (|denB| >= 1e-9, else
// the function returns 0 like the reference), its numerator is bounded through Z <= 1e300 (any Z that large gives
// B > 700 -> pd = 1 either way); the divisor of 1 / (1 + exp(-B)) lies in [1, e^709] and every result computed from
// B outside [-700, 700] is replaced by the reference's own saturation values.  Both refined reciprocals of a constant
// divisor are shared by all the evaluations of an env-step (3 instead of 11 instructions per division).
struct PdConsts {
    double A, c1, denB, r_denB;
    bool degenerate;   // |denB| < 1e-9 -> pd = 0 (radar.py:75-76)
};
__device__ __forceinline__ double rcp_refined(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double div_by_refined(double n, double d, double r) {
    const double q = n * r;
    return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}
__device__ __forceinline__ PdConsts pd_consts(double A, double c1, double denB) {
    PdConsts k;
    k.A = A; k.c1 = c1; k.denB = denB;
    k.degenerate = fabs(denB) < 1e-9;
    k.r_denB = rcp_refined(k.degenerate ? 1.0 : denB);
    return k;
}
// N independent evaluations side by side: straight-line code, the exp polynomial's constants are materialised once
// and the N dependency chains interleave (the one-at-a-time form spent ~150 v_mov on constants per env-step and ran
// each chain alone)
template <int N>
__device__ __forceinline__ void det_prob_batch(const double* snr, double* pd, const PdConsts& k) {
    double B[N], den[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double s = (0.0 > snr[i]) ? 0.0 : snr[i];  // Python max(snr, 0.0)
        double Z = s + k.c1;
        Z = (Z < 1e300) ? Z : 1e300;
        B[i] = div_by_refined(10.0 * Z - k.A, k.denB, k.r_denB);
        den[i] = 1.0 + exp(-B[i]);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double r = rcp_refined(den[i]);
        double p = div_by_refined(1.0, den[i], r);
        p = (B[i] > 700.0) ? 1.0 : p;
        p = (B[i] < -700.0) ? 0.0 : p;
        pd[i] = k.degenerate ? 0.0 : p;
    }
}
// REGULAR scenarios (host-checked, see macjd_scenario_create): denB > 0 and not degenerate, every SNR the step can form is
// finite, >= +0 and < 1e100, and B >= (10 c1 - A) / denB >= -700 for every SNR >= 0 — so max(snr, 0), the 1e300 cap and
// the B < -700 / degenerate selects of the general form never act; B > 700 needs no select at all: exp(-B) < 2^-53
// there (0 below -745), 1 + exp(-B) rounds to 1.0 and the quotient is exactly the reference's 1.0.
template <int N>
__device__ __forceinline__ void det_prob_batch_regular(const double* snr, double* pd, const PdConsts& k) {
    double den[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double Z = snr[i] + k.c1;
        const double B = div_by_refined(10.0 * Z - k.A, k.denB, k.r_denB);
        den[i] = 1.0 + exp(-B);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) pd[i] = div_by_refined(1.0, den[i], rcp_refined(den[i]));
}
__device__ __forceinline__ double det_prob(double snr, const PdConsts& k) {
    double p;
    det_prob_batch<1>(&snr, &p, k);
    return p;
}

// PD32 (production variant on regular scenarios): the detection probability in float32 on the transcendental unit —
// pd = 1 / (1 + 2^y), y = -log2(e) B = ky1 snr + ky0 (one fma, v_exp_f32, v_add, v_rcp_f32: 5 instructions instead of ~60
// float64 ones incl. a 13-term exp polynomial and two refined reciprocals) — used where a value within 1e-5 is all the
// consumer needs (the reward terms), and as a FILTER for the Monte-Carlo compares u <= pd.  The radars' SNR itself is a
// float32 quotient there (GaPs * v_rcp_f32(D supp + Pn), the sum still formed in float64): relative error <= 3.5e-7, i.e.
// |dy| <= |ky1| snr 3.5e-7 + (|y| + |ky0|) 1.8e-7 (the fma and the two constants); |dpd/dy| <= 0.173 pd (1 - pd), which is
// < 1e-3 wherever snr > 3 (B > 4.9): |dpd| <= 3e-7 from the argument; v_exp_f32 and v_rcp_f32 1 ulp each and the add's
// rounding add <= 3e-7: **|pd32 - pd| <= 6e-7**; the float32 image of u is within 1.2e-7.  A compare whose two sides differ
// by more than PD32_BOUND = 4e-6 therefore has the same outcome in float64; the others (~8e-6 of all compares) evaluate
// the exact float64 chain, SNR included.  FSM bits, hit decisions and therefore every integer output are those of the float64 kernel bit for bit; reward
// terms differ from it by <= (R + J) 6e-7 (tested: bitwise-equal track / terminated, rewards within 1e-5, on the 70 000-env
// edge batch).  MACJD_ENV_PD32=0 selects the all-float64 form.
constexpr float PD32_BOUND = 4e-6f;
struct Pd32Consts { float ky1, ky0; };
__device__ __forceinline__ Pd32Consts pd32_consts(double A, double c1, double denB) {
    const double l2e = 1.4426950408889634;
    return Pd32Consts{(float)(-l2e * 10.0 / denB), (float)(-l2e * (10.0 * c1 - A) / denB)};
}
__device__ __forceinline__ float det_prob32(float snr, const Pd32Consts& k) {
    const float y = __builtin_fmaf(snr, k.ky1, k.ky0);
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y));
}

// Uniform of (env e, slot) for the step that starts at `step_before` (include/macjd.h, macjd_step_io.u): supplied by the
// caller, or word (slot & 3) of the env's Philox block (slot >> 2).  This on-demand form generates a whole block per
// call; the kernels below generate each block they need ONCE per env-step and pick words out of it.
// Word w (0..3, possibly different per lane) of a block, as shifts on 64-bit pairs.  NOT b.v[w] and not a chain of
// selects on w either: hipcc turns both into a dynamically indexed private array, promotes that array to LDS and, to
// find its slice of it, reads the workgroup size from the dispatch packet — an uncached load from the AQL queue that
// cost the slot kernel 5 - 10 us per launch (3.9 -> 13.6 us at E = 4096; `.amdhsa_user_sgpr_dispatch_ptr 1` in the
// kernel descriptor is the tell-tale).
__device__ __forceinline__ uint32_t philox_word(const Philox4& b, int w) {
    const uint64_t lo = ((uint64_t)b.v[1] << 32) | b.v[0], hi = ((uint64_t)b.v[3] << 32) | b.v[2];
    const uint64_t pair = (w & 2) ? hi : lo;
    return (uint32_t)(pair >> ((w & 1) * 32));
}
template <bool PHILOX_ONLY = false, class IO = macjd_step_io>
__device__ __forceinline__ double draw_uniform(const IO& io, int64_t e, uint32_t episode, int slot,
                                               uint32_t step_before) {
    if (!PHILOX_ONLY && io.u) return io.u[e * io.u_se + (int64_t)slot * io.u_sx];
    const Philox4 r = env_philox_block(io.seed, (uint64_t)(io.env_offset + e), episode, step_before, (uint32_t)(slot >> 2));
    return u32_mid(philox_word(r, slot & 3));
}

// Argument block of the production (FAST) variants: only what that configuration reads or writes, strides as int32.
// A kernel taking the full macjd_step_io keeps ~95 SGPRs of arguments live next to ~70 SGPRs of wave-uniform table
// constants and spills (88 SGPR spills = ~280 v_readlane / v_writelane VALU instructions per env-step at 3j/4r).
struct FastStepIO {
    int64_t n_envs, env_offset;
    uint64_t seed;
    const int32_t* T;
    const float* P32;
    const int32_t* episode;
    uint8_t* track;
    int32_t* step;
    float* reward;
    float* r_dpj;
    uint8_t* terminated;
    float* pd;
    float* snr_with;
    float* r_dpj_sum;
    const double* pe_tables;
    const uint8_t* pe_flags;
    int64_t pe_stride;
    int32_t T_se, T_sx, P_se, P_sx, k_se, k_sx, pd_se, pd_sx, sw_se, sw_sx, pe_tile;
    // macjd_env_step_many: many_T > 0 runs many_T steps of each of n_envs envs as many_T * n_envs work items (time-major:
    // item v = t * n_envs + e), actions of step t at + t * t_stride elements
    int32_t many_T, t_stride;
    // members of macjd_step_io this configuration never has (compile-time constants: the code paths fold away)
    static constexpr const double* u = nullptr;
    static constexpr int64_t u_se = 0, u_sx = 0;
    static constexpr const double* P64 = nullptr;
    static constexpr uint32_t flags = 0;
    static constexpr double* out64 = nullptr;
    static constexpr double* pd64 = nullptr;
    static constexpr double* snr64 = nullptr;
    static constexpr double* prj64 = nullptr;
};

// PE = per-env scenario tables (io.pe_tables, SoA [row][env]): every table read becomes a load from this lane's
// column of the SoA (row index static for the per-jammer / per-radar loops, data-dependent for the reads gathered by
// the chosen target radar); the shared-table variant stages the gathered tables in LDS instead.
// FAST = the production configuration fixed at compile time: uniforms from Philox (io.u == NULL), float32 actions with
// the reference's float32 power arithmetic (io.P32, no MACJD_STEP_ARITH_F64), no float64 diagnostics.  At streaming
// sizes the kernel is VALU-issue bound (PMC: ~1670 VALU instructions per 64-env iteration, VALU busy 73 % of the
// kernel); the run-time mode tests cost ~50 uniform branches and, through the extra live pointers, ~400 SGPR spill
// instructions (v_writelane / v_readlane) per iteration.
// REG = FAST on the shared tables of a REGULAR scenario (DevTables.regular, decided by macjd_scenario_create): every
// division whose divisor is a table value — the received-power quotient (float64 or float32, chosen per (jammer, radar)
// pair), the false-target SNR, the power normalisation — multiplies by the divisor's refined reciprocal from the
// tables (3 instead of 13 / 10 instructions; a float32 quotient is the float64 one rounded once more, which is exact for
// a quotient of two float32 values: 53 >= 2 * 24 + 2 bits), the radars' SNR division drops the scaling wrapper, and
// the guards that a regular scenario can never trigger (non-positive or tiny noise power, SNR < 0, the probability
// formula's saturation selects) are not evaluated.  Same results bit for bit (tests: lane kernel vs (env x slot) kernel,
// which keeps IEEE division and every guard, and both vs the oracle); 1285 -> ~1050 VALU instructions per env-step at 3j/4r.
template <int JT, int RT, bool PE, bool FAST, class IO, bool REG = false, bool PD32 = false>
__global__ void __launch_bounds__(256) env_step_kernel(const DevTables* __restrict__ tb, const IO io) {
    static_assert(!REG || (FAST && !PE && JT && RT), "REG: production variant on shared tables, compiled sizes");
    static_assert(!PD32 || REG, "PD32: float32 detection-probability filter of the regular production variant");
    constexpr int NJ = JT ? JT : MAXJ;
    constexpr int NR = RT ? RT : MAXR;
    const int J = JT ? JT : tb->J;
    const int R = RT ? RT : tb->R;

    // ---- LDS staging of the tables gathered by per-lane target index (shared-table variant) ----
    __shared__ double s_denom[PE ? 1 : NJ * NR];   // REG: the divisor as used (DevTables.dsel)
    __shared__ double s_rsel[REG ? NJ * NR : 1], s_rPn[REG ? NR : 1];
    // REG: the per-radar suppression sums and false-target products of a lane live in its own LDS column and a jammer
    // adds into the row of the radar it chose (one ds_read / ds_write pair on the LDS port) instead of J x R
    // compare / select / fma chains on the VALU, which is the unit this kernel is bound by; row NR takes the
    // contributions of jammers that have none.  Same operations in the same (jammer) order on the same values.
    // (also the per-env-table production variant: its lanes hold ~45 table values each, every register counts)
#ifndef MACJD_PE_SCAT
#define MACJD_PE_SCAT 1   // build knob for A/B runs: 0 keeps the per-env variant's accumulators in registers
#endif
    constexpr bool SCAT = REG || (MACJD_PE_SCAT && PE && FAST && JT && RT);
    constexpr int ACC_W = SCAT ? 256 : 1;
    __shared__ double s_supp[SCAT ? NR + 1 : 1][ACC_W], s_prod[SCAT ? NR + 1 : 1][ACC_W];
    __shared__ double s_D[PE ? 1 : NR], s_Pn[PE ? 1 : NR], s_gr[PE ? 1 : NR];
    __shared__ uint8_t s_flags[PE ? 1 : NJ * NR];
    if (!PE) {
        for (int i = threadIdx.x; i < J * R; i += blockDim.x) {
            s_denom[i] = REG ? tb->dsel[i] : tb->denom[i];
            s_flags[i] = REG ? tb->rflags[i] : tb->flags[i];
            if (REG) s_rsel[i] = tb->rsel[i];
        }
        for (int i = threadIdx.x; i < R; i += blockDim.x) {
            s_D[i] = tb->D[i];
            s_Pn[i] = tb->Pn[i];
            s_gr[i] = tb->gr[i];
            if (REG) s_rPn[i] = tb->rPn[i];
        }
        __syncthreads();
    }

    const double rp_min = tb->rp_min, rp_max = tb->rp_max;
    const PdConsts pdk = pd_consts(tb->pd_A, tb->pd_c1, tb->pd_denB);
    const Pd32Consts pdk32 = pd32_consts(tb->pd_A, tb->pd_c1, tb->pd_denB);
    const int32_t episode_limit = tb->episode_limit;
    const bool arith32 = FAST ? true : ((io.P32 != nullptr) && !(io.flags & MACJD_STEP_ARITH_F64));

    auto env_step_one = [&](const int64_t e, const int32_t tt) {
        // many-step launches (production variant only, macjd_env_step_many): work item (step tt, env e), outputs of the
        // item at index tt * n_envs + e (the step is wave-uniform: a scalar multiply); single-step launches: tt = 0
        bool many = false, last_t = true;
        if constexpr (FAST) {
            if (io.many_T > 0) {   // grid = (envs / 256, steps): no division
                many = true;
                last_t = (tt == io.many_T - 1);
            }
        }
        const int64_t e_item = FAST ? (int64_t)((uint32_t)tt * (uint32_t)io.n_envs + (uint32_t)e) : e;
        int64_t act_extra = 0;   // element offset of step t's actions
        if constexpr (FAST) act_extra = (int64_t)tt * io.t_stride;
        // Element (env e, item k) of a caller-strided array.  Production variant: the BYTE offset is formed in 32 bits
        // (the host checks that every offset of the launch fits) and added to the wave-uniform base pointer, which is
        // the SGPR-base + 32-bit-VGPR-offset addressing form of global_load / global_store — no 64-bit VALU address
        // arithmetic per access (~50 instructions per env-step at 3j/4r).
        auto at = [&](auto* base, int64_t env, auto se, int k, auto sx, int64_t extra = 0) -> decltype(*base)& {
            using T_ = std::remove_reference_t<decltype(*base)>;
            if constexpr (FAST) {
                const uint32_t off = ((uint32_t)env * (uint32_t)se + (uint32_t)k * (uint32_t)sx + (uint32_t)extra) * (uint32_t)sizeof(T_);
                using C_ = std::conditional_t<std::is_const_v<T_>, const char, char>;
                return *reinterpret_cast<T_*>(reinterpret_cast<C_*>(base) + off);
            } else {
                return base[env * (int64_t)se + (int64_t)k * (int64_t)sx + extra];
            }
        };
        const int32_t step_before = at(io.step, e, 1, 0, 0) + tt;
        const int32_t step_count = step_before + 1;  // environment.py:235
        const uint32_t episode = io.episode ? (uint32_t)at(io.episode, e, 1, 0, 0) : 0u;
        // Monte-Carlo uniforms of this env-step (R radar slots, then one per valid deception action): every Philox
        // block is generated once (ceil((R + J) / 4) blocks: 2 at 3j/4r); the radar pass reads its words at
        // compile-time indices, a deception action selects word R + n_dec.
        constexpr bool PRE = JT && RT;   // generic sizes generate a block on demand instead
        constexpr int NBLK = PRE ? (NJ + NR + 3) / 4 : 1;
        uint32_t rw[NBLK * 4];
        const bool own_rng = FAST || !io.u;
        auto gen_blocks = [&](const int b0, const int b1) {
#pragma unroll
            for (int b = b0; b < b1; ++b) {
                const Philox4 blk = env_philox_block(io.seed, (uint64_t)(io.env_offset + e), episode,
                                                     (uint32_t)step_before, (uint32_t)b);
#pragma unroll
                for (int i = 0; i < 4; ++i) rw[b * 4 + i] = blk.v[i];
            }
        };
        if (PRE && own_rng) gen_blocks(0, NBLK);

        // table accessors: this env's SoA column (PE) or the shared tables
        // this env's column of the per-env tables: plain SoA (row stride pe_stride) or tiled (row stride = tile width)
        const int64_t ps = (PE && io.pe_tile > 0) ? (int64_t)io.pe_tile : io.pe_stride;
        const int64_t pe_tile_idx = (PE && io.pe_tile > 0) ? e / io.pe_tile : 0;
        const int64_t pe_col = (PE && io.pe_tile > 0) ? e - pe_tile_idx * io.pe_tile : e;
        const double* __restrict__ pe = PE ? io.pe_tables + pe_tile_idx * (6 * R + 3 * J + J * R) * ps + pe_col : nullptr;
        const uint8_t* __restrict__ pf = PE ? io.pe_flags + pe_tile_idx * (J * R) * ps + pe_col : nullptr;
        // PE, compile-time sizes: every table value this env-step can touch is requested up front so that all the
        // loads are in flight together — left at their use sites they sit behind branches and possibly-aliasing
        // stores and arrive one memory latency at a time.  Static-index rows (per radar / per jammer) directly; the
        // rows gathered by a jammer's chosen target radar after a pre-pass that decodes the J actions.
        constexpr bool HOIST = PE && JT && RT;
        double pv_r[HOIST ? 5 * NR : 1], pv_j[HOIST ? 3 * NJ : 1];
        double gv_denom[HOIST ? NJ : 1], gv_gr[HOIST ? NJ : 1], gv_Pn[HOIST ? NJ : 1], gv_D[HOIST ? NJ : 1];
        uint8_t gv_fl[HOIST ? NJ : 1];
        if (HOIST) {
#pragma unroll
            for (int k = 0; k < 5; ++k)
#pragma unroll
                for (int r = 0; r < NR; ++r) pv_r[k * NR + r] = pe[(int64_t)(k * R + r) * ps];   // GaPs, Pn, D, pd_no, rd_pen
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int j = 0; j < NJ; ++j) pv_j[k * NJ + j] = pe[(int64_t)(6 * R + k * J + j) * ps];   // pmin, pmax, gj
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int32_t Tj = at(io.T, e, io.T_se, j, io.T_sx, act_extra);
                const int tj = ((Tj >= 1) && (Tj <= 2 * R)) ? ((Tj + 1) / 2 - 1) : 0;
                gv_denom[j] = pe[(int64_t)(6 * R + 3 * J + j * R + tj) * ps];
                gv_gr[j] = pe[(int64_t)(5 * R + tj) * ps];
                gv_Pn[j] = pe[(int64_t)(R + tj) * ps];
                gv_D[j] = pe[(int64_t)(2 * R + tj) * ps];
                gv_fl[j] = pf[(int64_t)(j * R + tj) * ps];
            }
        }
        auto t_GaPs = [&](int r) { return HOIST ? pv_r[0 * NR + r] : PE ? pe[(int64_t)(r) * ps] : tb->GaPs[r]; };
        auto t_Pn = [&](int r) { return HOIST ? pv_r[1 * NR + r] : PE ? pe[(int64_t)(R + r) * ps] : tb->Pn[r]; };
        auto t_D = [&](int r) { return HOIST ? pv_r[2 * NR + r] : PE ? pe[(int64_t)(2 * R + r) * ps] : tb->D[r]; };
        auto t_pdno = [&](int r) { return HOIST ? pv_r[3 * NR + r] : PE ? pe[(int64_t)(3 * R + r) * ps] : tb->pd_no[r]; };
        auto t_rdpen = [&](int r) { return HOIST ? pv_r[4 * NR + r] : PE ? pe[(int64_t)(4 * R + r) * ps] : tb->rd_pen[r]; };
        auto t_pmin = [&](int j) { return HOIST ? pv_j[0 * NJ + j] : PE ? pe[(int64_t)(6 * R + j) * ps] : tb->pmin[j]; };
        auto t_pmax = [&](int j) { return HOIST ? pv_j[1 * NJ + j] : PE ? pe[(int64_t)(6 * R + J + j) * ps] : tb->pmax[j]; };
        auto t_gj = [&](int j) { return HOIST ? pv_j[2 * NJ + j] : PE ? pe[(int64_t)(6 * R + 2 * J + j) * ps] : tb->gj[j]; };
        // gathered by jammer j's chosen target radar t
        auto g_gr = [&](int j, int t) { return HOIST ? gv_gr[j] : PE ? pe[(int64_t)(5 * R + t) * ps] : s_gr[t]; };
        auto g_Pn = [&](int j, int t) { return HOIST ? gv_Pn[j] : PE ? pe[(int64_t)(R + t) * ps] : s_Pn[t]; };
        auto g_D = [&](int j, int t) { return HOIST ? gv_D[j] : PE ? pe[(int64_t)(2 * R + t) * ps] : s_D[t]; };
        auto g_denom = [&](int j, int t) {
            return HOIST ? gv_denom[j] : PE ? pe[(int64_t)(6 * R + 3 * J + j * R + t) * ps] : s_denom[j * R + t];
        };
        auto g_flags = [&](int j, int t) {
            return HOIST ? gv_fl[j] : PE ? pf[(int64_t)(j * R + t) * ps] : s_flags[j * R + t];
        };

        // The step runs as straight-line passes over the (compile-time unrolled) jammers and radars: (1) decode, power,
        // received power, suppression sums; (2) SNR per radar; (3) ALL R + J detection probabilities side by side
        // (det_prob_batch); (4) the deception draws in jammer order; (5) detection draws, FSM, reward terms in radar
        // order.  Same arithmetic, operation order and RNG slots as the reference's interleaved loops; the deception
        // quantities of a jammer that is not deceiving are computed on zeros and never used.
        constexpr int NP = PRE ? NR + NJ : 1;
        double supp[NR];   // environment.py:241
        double prod[NR];   // environment.py:443-447
#pragma unroll
        for (int r = 0; r < NR; ++r) { supp[r] = 0.0; prod[r] = 1.0; }
        const int col = SCAT ? (int)threadIdx.x : 0;
        if (SCAT) {
#pragma unroll
            for (int r = 0; r < NR; ++r) { s_supp[r][col] = 0.0; s_prod[r][col] = 1.0; }
        }
        uint32_t supp_mask = 0;  // environment.py:391-394 (set of suppressed radars)
        uint32_t hit_mask = 0;   // radars with >= 1 detected false target
        int n_dec = 0;           // valid deception actions so far (RNG slot R + k)
        double r_p = 0.0;        // environment.py:371-378
        double snr_all[NP], pd_all[NP];   // [0, NR): radars, [NR, NR + NJ): false targets of deceiving jammers
        int dec_tgt[NJ];                  // target radar of jammer j's valid deception action, -1 = none

#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (!JT && j >= J) break;
            // ---- action decode, environment.py:249-268 ----
            const int32_t T = at(io.T, e, io.T_se, j, io.T_sx, act_extra);
            const bool is_jamming = (T >= 1) && (T <= 2 * R);
            const int target = is_jamming ? ((T + 1) / 2 - 1) : 0;
            const int jtype = T % 2;  // 1 = suppression, 0 = deception (only read when is_jamming)

            // ---- power scale + r_p term, environment.py:271-277 ----
            const double pmin = t_pmin(j), pmax = t_pmax(j);
            const double power_range = pmax - pmin;
            double actual_d, norm;
            float actual_f = 0.0f;
            if (arith32) {
                float Pc = at(io.P32, e, io.P_se, j, io.P_sx, act_extra);
                Pc = Pc < 0.0f ? 0.0f : (Pc > 1.0f ? 1.0f : Pc);  // np.clip, NaN propagates
                actual_f = (float)pmin + Pc * (float)power_range;
                actual_d = (double)actual_f;
                if (REG)
                    norm = (power_range > 1e-6) ? (double)(float)div_by_refined((double)(actual_f - (float)pmin), tb->range_fd[j],
                                                                                tb->r_range[j]) : 0.0;
                else
                    norm = (power_range > 1e-6) ? (double)((actual_f - (float)pmin) / (float)power_range) : 0.0;
            } else {
                double Pc = io.P64 ? io.P64[e * io.P_se + (int64_t)j * io.P_sx]
                                   : (double)at(io.P32, e, io.P_se, j, io.P_sx, act_extra);
                Pc = Pc < 0.0 ? 0.0 : (Pc > 1.0 ? 1.0 : Pc);
                actual_d = pmin + Pc * power_range;
                norm = (power_range > 1e-6) ? (actual_d - pmin) / power_range : 0.0;
            }
            r_p += rp_max + (rp_min - rp_max) * norm;  // environment.py:377-378

            // ---- received jamming power, environment.py:280-302, jammer.py:56-98 ----
            const double denom = g_denom(j, target);
            const bool recorded = REG ? (is_jamming && (actual_d > 0.0) && (g_flags(j, target) & JR_RECORDABLE))
                                      : (is_jamming && (actual_d > 0.0) && (denom >= 0.0));
            double prj = 0.0;
            if (REG) {
                // `denom` is the divisor as the reference uses it here (1.0 where it does not divide)
                const uint8_t fl = g_flags(j, target);
                const bool live = recorded && (fl & JR_LIVE);
                const float num = (actual_f * (float)t_gj(j)) * (float)g_gr(j, target);
                const double q64 = div_by_refined((double)num, denom, s_rsel[j * R + target]);
                const double q = (fl & MACJD_JR_WEAK_DENOM) ? (double)(float)q64 : q64;
                prj = (live && q > 0.0) ? q : 0.0;  // Python max(0.0, x)
            } else if (PRE) {
                // branch-free: the quotient is evaluated on a harmless divisor where the reference does not divide
                const bool live = recorded && denom > 1e-18;
                const double dsafe = live ? denom : 1.0;
                const double grj = g_gr(j, target);
                double q;
                if (arith32) {
                    const float num = (actual_f * (float)t_gj(j)) * (float)grj;
                    q = (g_flags(j, target) & MACJD_JR_WEAK_DENOM) ? (double)(num / (float)dsafe) : (double)num / dsafe;
                } else {
                    q = (actual_d * t_gj(j) * grj) / dsafe;
                }
                prj = (live && q > 0.0) ? q : 0.0;  // Python max(0.0, x)
            } else if (recorded && denom > 1e-18) {
                const double grj = g_gr(j, target);
                if (arith32) {
                    const float num = (actual_f * (float)t_gj(j)) * (float)grj;
                    prj = (g_flags(j, target) & MACJD_JR_WEAK_DENOM) ? (double)(num / (float)denom)
                                                                         : (double)num / denom;
                } else {
                    prj = (actual_d * t_gj(j) * grj) / denom;
                }
                prj = (prj > 0.0) ? prj : 0.0;  // Python max(0.0, x)
            }
            if (!FAST && io.prj64) io.prj64[e * J + j] = recorded ? prj : -1.0;

            const bool is_sup = recorded && (jtype == 1);
            const bool is_dec = recorded && (jtype == 0);
            supp_mask |= is_sup ? (1u << target) : 0u;
            if (SCAT) {                    // environment.py:299
                const int row = is_sup ? target : NR;
                s_supp[row][col] = s_supp[row][col] + prj;
            } else {
#pragma unroll
                for (int r = 0; r < NR; ++r)   // fma(prj, 1 or 0, supp) == supp + prj or supp, bit for bit
                    supp[r] = PRE ? __builtin_fma(prj, (is_sup && target == r) ? 1.0 : 0.0, supp[r])
                                  : supp[r] + ((is_sup && target == r) ? prj : 0.0);
            }
            dec_tgt[j] = is_dec ? target : -1;

            // ---- deception: SNR of the false target, environment.py:410-422 ----
            if (REG) {
                const double snr_f = div_by_refined(g_D(j, target) * prj, g_Pn(j, target), s_rPn[target]);
                snr_all[NR + j] = (is_dec && snr_f > 0.0) ? snr_f : 0.0;
            } else if (PRE) {
                const double Pn_t = g_Pn(j, target);
                const bool live = is_dec && Pn_t > 1e-18;
                const double snr_f = (g_D(j, target) * prj) / (live ? Pn_t : 1.0);
                snr_all[NR + j] = (live && snr_f > 0.0) ? snr_f : 0.0;
            } else if (is_dec) {   // generic sizes: one evaluation at a time, only where needed
                const double Pn_t = g_Pn(j, target);
                double snr_f = (Pn_t > 1e-18) ? (g_D(j, target) * prj) / Pn_t : 0.0;
                snr_f = (snr_f > 0.0) ? snr_f : 0.0;
                const double pd_f = det_prob(snr_f, pdk);
                const double u = draw_uniform<FAST>(io, e, episode, R + n_dec, (uint32_t)step_before);
                ++n_dec;
                if (u <= pd_f) {
                    const double safe = pd_f < 0.999999 ? pd_f : 0.999999;  // environment.py:446
                    hit_mask |= (1u << target);
#pragma unroll
                    for (int r = 0; r < NR; ++r) prod[r] = (target == r) ? prod[r] * (1.0 - safe) : prod[r];
                }
            }
        }

        // ---- SNR with jamming, environment.py:316-333 ----
        double snr_w[NR];
        float snr32r[PD32 ? NR : 1];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (!RT && r >= R) break;
            const double Pn = t_Pn(r);
            const double den = t_D(r) * (SCAT ? s_supp[r][col] : supp[r]) + Pn;   // :331
            if (PD32) {                                                      // float32 quotient (see det_prob32)
                snr32r[PD32 ? r : 0] = (float)t_GaPs(r) * __builtin_amdgcn_rcpf((float)den);
                snr_w[r] = 0.0;   // (not formed; the outputs below take the float32 value)
            } else if (REG) {                                                // den >= Pn > 1e-18
                snr_w[r] = div_by_refined(t_GaPs(r), den, rcp_refined(den));
                snr_all[r] = snr_w[r];
            } else if (PRE) {
                const bool live = den > 1e-18;
                const double q = t_GaPs(r) / (live ? den : 1.0);
                snr_w[r] = live ? q : 0.0;                                   // :332
                snr_all[r] = snr_w[r];
            } else {
                snr_w[r] = (den > 1e-18) ? t_GaPs(r) / den : 0.0;            // :332
            }
        }
        // PD32: all R + J compares u <= pd decided here — by the float32 value where the two sides are further apart than
        // its error bound, by ONE shared copy of the exact float64 chain (a rarely taken loop) for the others
        uint32_t le_bits = 0;
        float pd32f[PD32 ? NP : 1];   // (kept as float32 until used: 7 registers instead of 14)
        auto pd_at = [&](int i) -> double { return PD32 ? (double)pd32f[PD32 ? i : 0] : pd_all[i]; };
        if constexpr (PD32) {
            uint32_t w_all[NP];
#pragma unroll
            for (int r = 0; r < NR; ++r) w_all[r] = rw[r];
            {
                int nd = 0;   // RNG slot of jammer j's valid deception action: R + (valid deception actions before it)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    uint32_t w = rw[NR];
#pragma unroll
                    for (int k = 1; k < NJ; ++k) w = (nd == k) ? rw[NR + k] : w;
                    w_all[NR + j] = w;
                    nd += (dec_tgt[j] >= 0) ? 1 : 0;
                }
            }
            uint32_t close_bits = 0;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const float p32 = det_prob32(i < NR ? snr32r[PD32 && i < NR ? i : 0] : (float)snr_all[i], pdk32);
                pd32f[i] = p32;
                const float d = __builtin_fmaf((float)w_all[i], 0x1p-32f, 0x1p-33f) - p32;
                const bool need = (i < NR) || (dec_tgt[i < NR ? 0 : i - NR] >= 0);
                le_bits |= (d <= 0.0f) ? (1u << i) : 0u;
                close_bits |= (need && fabsf(d) <= PD32_BOUND) ? (1u << i) : 0u;
            }
            if (close_bits) {
#pragma unroll 1
                for (int i = 0; i < NP; ++i) {
                    if (!((close_bits >> i) & 1u)) continue;
                    double snr_i = snr_all[NR];
                    uint32_t w_i = w_all[0];
#pragma unroll
                    for (int k = 1; k < NP; ++k) {
                        if (k > NR) snr_i = (i == k) ? snr_all[k] : snr_i;
                        w_i = (i == k) ? w_all[k] : w_i;
                    }
                    if (i < NR) {   // a radar: the exact SNR from the suppression sum (still in this lane's LDS column)
                        const double den_i = tb->D[i] * s_supp[SCAT ? i : 0][col] + tb->Pn[i];
                        snr_i = div_by_refined(tb->GaPs[i], den_i, rcp_refined(den_i));
                    }
                    double p;
                    det_prob_batch_regular<1>(&snr_i, &p, pdk);
                    le_bits = (u32_mid(w_i) <= p) ? (le_bits | (1u << i)) : (le_bits & ~(1u << i));
                }
            }
        } else if (REG) det_prob_batch_regular<NP>(snr_all, pd_all, pdk);    // :337, :425
        else if (PRE) det_prob_batch<NP>(snr_all, pd_all, pdk);

        // ---- deception: Monte-Carlo detection of the false targets in jammer order, environment.py:425-447 ----
        if (PRE) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bool is_dec = dec_tgt[j] >= 0;
                double u = 2.0;
                uint32_t w = 0;
                if (own_rng) {
                    w = rw[NR];
#pragma unroll
                    for (int k = 1; k < NJ; ++k) w = (n_dec == k) ? rw[NR + k] : w;
                    if (!PD32) u = u32_mid(w);
                } else {
                    u = is_dec ? io.u[e * io.u_se + (int64_t)(R + n_dec) * io.u_sx] : 2.0;
                }
                n_dec += is_dec ? 1 : 0;
                const double pd_f = pd_at(NR + j);
                const bool hit = PD32 ? (is_dec && ((le_bits >> (NR + j)) & 1u))
                                      : (is_dec && (u <= pd_f));             // :430-434
                const double safe = pd_f < 0.999999 ? pd_f : 0.999999;      // :446
                hit_mask |= hit ? (1u << dec_tgt[j]) : 0u;
                if (SCAT) {
                    const int row = hit ? dec_tgt[j] : NR;
                    s_prod[row][col] = s_prod[row][col] * (1.0 - safe);
                } else {
#pragma unroll
                    for (int r = 0; r < NR; ++r) prod[r] = (hit && dec_tgt[j] == r) ? prod[r] * (1.0 - safe) : prod[r];
                }
            }
        }

        // ---- detections, FSM, r_d, r_j(suppression); environment.py:337-398 ----
        double r_d = 0.0, r_j = 0.0, r_j_dec = 0.0;
        double pd_r[NR];
        uint32_t track_bits = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (!RT && r >= R) break;
            const double pd = PRE ? pd_at(r) : det_prob(snr_w[r], pdk);     // :337
            bool detected;
            if constexpr (PD32) {
                detected = (le_bits >> r) & 1u;                                // :341
            } else {
                const double u = (PRE && own_rng) ? u32_mid(rw[r]) : draw_uniform<FAST>(io, e, episode, r, (uint32_t)step_before);
                detected = (u <= pd);                                          // :341
            }
            // radar.py:102-117: SEARCH & detected -> TRACK, SEARCH & !detected -> SEARCH,
            // TRACK & !detected -> SEARCH, TRACK & detected -> TRACK.  The next state therefore
            // equals `detected` whatever the previous state was, so the previous state is never
            // read (saves R bytes of HBM reads per env-step); `track` is write-only here.
            const bool tracking = detected;
            track_bits |= tracking ? (1u << r) : 0u;
            r_d += tracking ? t_rdpen(r) : 0.0;                             // :359-366 (post-update state)
            const double red = t_pdno(r) - pd;                               // :396-398
            r_j += ((supp_mask & (1u << r)) && red > 0.0) ? red : 0.0;
            r_j_dec += (hit_mask & (1u << r)) ? 1.0 - (SCAT ? s_prod[r][col] : prod[r]) : 0.0;   // :438-451
            pd_r[r] = pd;
        }
        // per-radar outputs (the only conditional code of the pass: kept behind the arithmetic)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (!RT && r >= R) break;
            const double snr_rep = PD32 ? (double)snr32r[PD32 ? r : 0]
                                        : ((REG || snr_w[r] > 0.0) ? snr_w[r] : 0.0);  // :333 (regular: never negative)
            if (last_t) at(io.track, e, io.k_se, r, io.k_sx) = (track_bits >> r) & 1u;
            if (io.pd) at(io.pd, e, io.pd_se, r, io.pd_sx) = (float)pd_r[r];
            if (io.snr_with) at(io.snr_with, e, io.sw_se, r, io.sw_sx) = (float)snr_rep;
            if (!FAST && io.pd64) io.pd64[e * R + r] = pd_r[r];
            if (!FAST && io.snr64) io.snr64[e * R + r] = snr_rep;
        }
        r_j += r_j_dec;                          // :454
        const double reward = r_d + r_p + r_j;  // :457

        if (!many) at(io.step, e, 1, 0, 0) = step_count;   // many-step: advanced by env_advance_kernel afterwards
        if (io.terminated) at(io.terminated, e_item, 1, 0, 0) = (step_count >= episode_limit) ? 1 : 0;  // :460
        if (io.reward) at(io.reward, e_item, 1, 0, 0) = (float)reward;
        if (io.r_dpj) {
            at(io.r_dpj, e_item, 3, 0, 1) = (float)r_d;
            at(io.r_dpj, e_item, 3, 1, 1) = (float)r_p;
            at(io.r_dpj, e_item, 3, 2, 1) = (float)r_j;
        }
        if (io.r_dpj_sum && !many) {   // many-step: summed over the steps by env_advance_kernel
            at(io.r_dpj_sum, e, 3, 0, 1) += (float)r_d;
            at(io.r_dpj_sum, e, 3, 1, 1) += (float)r_p;
            at(io.r_dpj_sum, e, 3, 2, 1) += (float)r_j;
        }
        if (!FAST && io.out64) {
            io.out64[e * 4 + 0] = reward;
            io.out64[e * 4 + 1] = r_d;
            io.out64[e * 4 + 2] = r_p;
            io.out64[e * 4 + 3] = r_j;
        }
    };
    if constexpr (FAST) {
        // one env per lane, no grid-stride loop: inside a loop every argument and table constant is loop-invariant,
        // gets hoisted and stays live for the whole body (SGPR spills); straight-line code loads them where used
        const uint32_t env = blockIdx.x * blockDim.x + threadIdx.x;   // (the host checks that every offset fits 32 bits)
        if ((int64_t)env < io.n_envs) env_step_one((int64_t)env, (int32_t)blockIdx.y);   // (blockIdx.y = step of a many-step launch)
    } else {
        for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < io.n_envs;
             e += (int64_t)gridDim.x * blockDim.x)
            env_step_one(e, 0);
    }
}

// ---------------------------------------------------------------------------------------------
// env_step_slots_kernel: one lane = one (env, slot) pair, slot = a jammer or a radar.
//
// Workgroup = 64 consecutive envs x (NJW jammer waves + NRW radar waves).  Lane l of every wave works on
// env 64*blockIdx.x + l, so agent-major / radar-major tensors are read and written as whole 256-B rows.
//   phase 1  jammer waves: decode, power, r_p term, received power          -> LDS (prj, meta, r_p term)
//   phase 2  jammer waves: deception false-target Pd + Monte-Carlo hit      -> LDS (1 - min(pd_f, .999999))
//            radar waves (concurrently): suppression sum in jammer order, SNR, Pd, detection draw, FSM,
//            r_d / r_j(suppression) terms; pd / snr / track outputs           -> LDS (r_d, r_j terms)
//   phase 3  radar waves: per-radar deception product in jammer order        -> LDS
//   phase 4  wave 0: the three ordered sums (radar / jammer index order, as the reference), outputs.
// The arithmetic, operation order and RNG slots are identical to env_step_kernel (and the reference);
// only the placement on lanes changes: the per-env critical path is two Pd evaluations instead of J + R,
// there is no per-lane J x R select chain and no divergent deception branch inside a radar/jammer lane.
// Tables indexed by the lane's chosen target radar (<= 128-B rows of a <= 11.5 KB table shared by the whole
// grid) are gathered straight from L1/L2; radar-wave constants are wave-uniform scalar loads.
template <int J, int R, int NJW, int NRW>
__global__ void __launch_bounds__(64 * (NJW + NRW)) env_step_slots_kernel(const DevTables* __restrict__ tb,
                                                                          const macjd_step_io io) {
    constexpr int JPW = (J + NJW - 1) / NJW;
    constexpr int RPW = (R + NRW - 1) / NRW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t e_raw = (int64_t)blockIdx.x * 64 + lane;
    const bool active = e_raw < io.n_envs;
    const int64_t e = active ? e_raw : io.n_envs - 1;  // clamp: inactive lanes compute on a valid env, store nothing

    __shared__ double s_prj[J][64];
    __shared__ double s_rpterm[J][64];
    __shared__ double s_hitfac[J][64];
    __shared__ int s_meta[J][64];  // bits 0..7 target radar, bit 8 valid suppression, bit 9 valid deception
    __shared__ double s_rd[R][64], s_rjs[R][64], s_rjd[R][64];

    const PdConsts pdk = pd_consts(tb->pd_A, tb->pd_c1, tb->pd_denB);
    const bool arith32 = (io.P32 != nullptr) && !(io.flags & MACJD_STEP_ARITH_F64);
    const int32_t step_before = io.step[e];
    const uint32_t episode = io.episode ? (uint32_t)io.episode[e] : 0u;
    const bool jam_wave = wave < NJW;
    // deception slots R .. R+J-1: when they all live in ONE Philox block (3j/4r: slots 4..6 of block 1) the jammer
    // waves generate it in phase 1, off the post-barrier path; otherwise a block per deceiving lane in phase 2
    constexpr bool DEC_ONE_BLOCK = (R >> 2) == ((R + J - 1) >> 2);
    Philox4 dec_blk;
    if (DEC_ONE_BLOCK && jam_wave && !io.u)
        dec_blk = env_philox_block(io.seed, (uint64_t)(io.env_offset + e), episode, (uint32_t)step_before, (uint32_t)(R >> 2));

    // ---------------- phase 1: jammer lanes, environment.py:248-302 ----------------
    double my_prj[JPW];
    int my_meta[JPW];
    double u_radar[RPW];   // radar waves are idle in phase 1: they draw their detection uniforms now (the draws
                           // depend only on (env, step, slot), not on the actions), off the post-barrier path
    if (!jam_wave) {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = (wave - NJW) + rr * NRW;
            u_radar[rr] = (r < R) ? draw_uniform(io, e, episode, r, (uint32_t)step_before) : 2.0;
        }
    }
    if (jam_wave) {
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            const int j = wave + jj * NJW;  // wave-uniform
            my_prj[jj] = 0.0;
            my_meta[jj] = 0;
            if (j < J) {
                const int32_t T = io.T[e * io.T_se + (int64_t)j * io.T_sx];
                const bool is_jamming = (T >= 1) && (T <= 2 * R);
                const int target = is_jamming ? ((T + 1) / 2 - 1) : 0;
                const int jtype = T % 2;
                const double pmin = tb->pmin[j], pmax = tb->pmax[j];
                const double power_range = pmax - pmin;
                double actual_d, norm;
                float actual_f = 0.0f;
                if (arith32) {
                    float Pc = io.P32[e * io.P_se + (int64_t)j * io.P_sx];
                    Pc = Pc < 0.0f ? 0.0f : (Pc > 1.0f ? 1.0f : Pc);
                    actual_f = (float)pmin + Pc * (float)power_range;
                    actual_d = (double)actual_f;
                    norm = (power_range > 1e-6) ? (double)((actual_f - (float)pmin) / (float)power_range) : 0.0;
                } else {
                    double Pc = io.P64 ? io.P64[e * io.P_se + (int64_t)j * io.P_sx]
                                       : (double)io.P32[e * io.P_se + (int64_t)j * io.P_sx];
                    Pc = Pc < 0.0 ? 0.0 : (Pc > 1.0 ? 1.0 : Pc);
                    actual_d = pmin + Pc * power_range;
                    norm = (power_range > 1e-6) ? (actual_d - pmin) / power_range : 0.0;
                }
                const double denom = tb->denom[j * R + target];
                const bool recorded = is_jamming && (actual_d > 0.0) && (denom >= 0.0);
                double prj = 0.0;
                if (recorded && denom > 1e-18) {
                    const double grj = tb->gr[target];
                    if (arith32) {
                        const float num = (actual_f * (float)tb->gj[j]) * (float)grj;
                        prj = (tb->flags[j * R + target] & MACJD_JR_WEAK_DENOM) ? (double)(num / (float)denom)
                                                                               : (double)num / denom;
                    } else {
                        prj = (actual_d * tb->gj[j] * grj) / denom;
                    }
                    prj = (prj > 0.0) ? prj : 0.0;
                }
                const int meta = target | ((recorded && jtype == 1) ? 0x100 : 0) | ((recorded && jtype == 0) ? 0x200 : 0);
                my_prj[jj] = prj;
                my_meta[jj] = meta;
                s_prj[j][lane] = prj;
                s_meta[j][lane] = meta;
                s_rpterm[j][lane] = tb->rp_max + (tb->rp_min - tb->rp_max) * norm;  // environment.py:377
                if (io.prj64 && active) io.prj64[e * J + j] = recorded ? prj : -1.0;
            }
        }
    }
    __syncthreads();

    // ---------------- phase 2 ----------------
    if (jam_wave) {
        // deception: false-target detection, environment.py:410-434
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            const int j = wave + jj * NJW;
            if (j < J) {
                double hitfac = -1.0;
                if (my_meta[jj] & 0x200) {
                    int k = 0;  // valid deception actions of lower-numbered jammers -> RNG slot R + k
                    for (int j2 = 0; j2 < j; ++j2) k += (s_meta[j2][lane] >> 9) & 1;
                    const int target = my_meta[jj] & 0xff;
                    const double Pn_t = tb->Pn[target];
                    double snr_f = (Pn_t > 1e-18) ? (tb->D[target] * my_prj[jj]) / Pn_t : 0.0;
                    snr_f = (snr_f > 0.0) ? snr_f : 0.0;
                    const double pd_f = det_prob(snr_f, pdk);
                    const double u = (DEC_ONE_BLOCK && !io.u) ? u32_mid(philox_word(dec_blk, (R + k) & 3))
                                                              : draw_uniform(io, e, episode, R + k, (uint32_t)step_before);
                    if (u <= pd_f) hitfac = 1.0 - (pd_f < 0.999999 ? pd_f : 0.999999);
                }
                s_hitfac[j][lane] = hitfac;
            }
        }
    } else {
        // radars: detections, FSM, r_d, r_j(suppression); environment.py:316-398
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = (wave - NJW) + rr * NRW;  // wave-uniform
            if (r < R) {
                double supp = 0.0;
                bool targeted = false;
#pragma unroll
                for (int j = 0; j < J; ++j) {  // jammer order, environment.py:299
                    const int meta = s_meta[j][lane];
                    const bool hit = (meta & 0x100) && ((meta & 0xff) == r);
                    supp += hit ? s_prj[j][lane] : 0.0;
                    targeted |= hit;
                }
                const double Pn = tb->Pn[r];
                const double den = tb->D[r] * supp + Pn;
                const double snr_with = (den > 1e-18) ? tb->GaPs[r] / den : 0.0;
                const double pd = det_prob(snr_with, pdk);
                const bool tracking = (u_radar[rr] <= pd);  // next FSM state == detected (radar.py:102-117)
                const double red = tb->pd_no[r] - pd;
                s_rd[r][lane] = tracking ? tb->rd_pen[r] : 0.0;
                s_rjs[r][lane] = (targeted && red > 0.0) ? red : 0.0;
                if (active) {
                    const double snr_rep = (snr_with > 0.0) ? snr_with : 0.0;
                    io.track[e * io.k_se + (int64_t)r * io.k_sx] = tracking ? 1 : 0;
                    if (io.pd) io.pd[e * io.pd_se + (int64_t)r * io.pd_sx] = (float)pd;
                    if (io.snr_with) io.snr_with[e * io.sw_se + (int64_t)r * io.sw_sx] = (float)snr_rep;
                    if (io.pd64) io.pd64[e * R + r] = pd;
                    if (io.snr64) io.snr64[e * R + r] = snr_rep;
                }
            }
        }
    }
    __syncthreads();

    // ---------------- phase 3: per-radar deception product, environment.py:437-451 ----------------
    if (!jam_wave) {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = (wave - NJW) + rr * NRW;
            if (r < R) {
                double prod = 1.0;
                bool any = false;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const int meta = s_meta[j][lane];
                    const double hf = s_hitfac[j][lane];
                    const bool hit = (meta & 0x200) && ((meta & 0xff) == r) && (hf >= 0.0);
                    prod = hit ? prod * hf : prod;
                    any |= hit;
                }
                s_rjd[r][lane] = any ? (1.0 - prod) : 0.0;
            }
        }
    }
    __syncthreads();

    // ---------------- phase 4: ordered sums + per-env outputs, environment.py:353-460 ----------------
    if (wave == 0 && active) {
        double r_d = 0.0, r_p = 0.0, r_j = 0.0, r_j_dec = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) { r_d += s_rd[r][lane]; r_j += s_rjs[r][lane]; r_j_dec += s_rjd[r][lane]; }
#pragma unroll
        for (int j = 0; j < J; ++j) r_p += s_rpterm[j][lane];
        r_j += r_j_dec;
        const double reward = r_d + r_p + r_j;
        const int32_t step_count = step_before + 1;
        io.step[e] = step_count;
        if (io.terminated) io.terminated[e] = (step_count >= tb->episode_limit) ? 1 : 0;
        if (io.reward) io.reward[e] = (float)reward;
        if (io.r_dpj) {
            io.r_dpj[e * 3 + 0] = (float)r_d;
            io.r_dpj[e * 3 + 1] = (float)r_p;
            io.r_dpj[e * 3 + 2] = (float)r_j;
        }
        if (io.r_dpj_sum) {
            io.r_dpj_sum[e * 3 + 0] += (float)r_d;
            io.r_dpj_sum[e * 3 + 1] += (float)r_p;
            io.r_dpj_sum[e * 3 + 2] += (float)r_j;
        }
        if (io.out64) {
            io.out64[e * 4 + 0] = reward;
            io.out64[e * 4 + 1] = r_d;
            io.out64[e * 4 + 2] = r_p;
            io.out64[e * 4 + 3] = r_j;
        }
    }
}

__global__ void env_reset_kernel(int64_t n_envs, int R, uint8_t* track, int64_t k_se, int64_t k_sx, int32_t* step,
                                 const uint8_t* mask, int32_t* episode) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_envs;
         e += (int64_t)gridDim.x * blockDim.x) {
        if (mask && !mask[e]) continue;
        for (int r = 0; r < R; ++r) track[e * k_se + (int64_t)r * k_sx] = 0;  // radar.py:10 initial_state SEARCH
        step[e] = 0;                                                         // environment.py:203
        if (episode) episode[e] += 1;   // a new episode draws fresh Monte-Carlo values (environment.py:341,430)
    }
}

// second half of macjd_env_step_many: the step counters advance by T, the per-episode reward-component sums get the T
// steps' (r_d, r_p, r_j) added in step order (deterministic).  One thread per (env, component): for a given step the
// 3 E values are contiguous, so every load of a wave is one coalesced row piece (one thread per env walking its three
// columns through [T, E, 3] was 31 us at E = 4096, T = 100), eight steps' loads in flight.
__global__ void env_advance_kernel(int64_t n_envs, int32_t T, int32_t* step, const float* r_dpj, float* r_dpj_sum) {
    const int64_t n3 = n_envs * 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < n_envs) step[i] += T;
        if (r_dpj && r_dpj_sum) {
            float a = r_dpj_sum[i];
            int t = 0;
            for (; t + 8 <= T; t += 8) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = r_dpj[(int64_t)(t + k) * n3 + i];
#pragma unroll
                for (int k = 0; k < 8; ++k) a += v[k];
            }
            for (; t < T; ++t) a += r_dpj[(int64_t)t * n3 + i];
            r_dpj_sum[i] = a;
        }
    }
}

// One launch when a scenario is created: the derived tables of DevTables, computed with the device functions the step
// kernels use (a host-computed reciprocal could differ from rcp_refined's in the last bit).
__global__ void scenario_derive_kernel(DevTables* t) {
    const int J = t->J, R = t->R;
    for (int i = threadIdx.x; i < J * R; i += blockDim.x) {
        const double denom = t->denom[i];
        const uint8_t fl = t->flags[i];
        const bool live = denom > 1e-18;
        const double d = live ? ((fl & MACJD_JR_WEAK_DENOM) ? (double)(float)denom : denom) : 1.0;
        t->dsel[i] = d;
        t->rsel[i] = rcp_refined(d);
        t->rflags[i] = (uint8_t)(fl | (denom >= 0.0 ? JR_RECORDABLE : 0) | (live ? JR_LIVE : 0));
    }
    for (int r = threadIdx.x; r < R; r += blockDim.x) t->rPn[r] = rcp_refined(t->Pn[r] > 1e-18 ? t->Pn[r] : 1.0);
    for (int j = threadIdx.x; j < J; j += blockDim.x) {
        const double rf = (double)(float)(t->pmax[j] - t->pmin[j]);
        t->range_fd[j] = rf;
        t->r_range[j] = rcp_refined(rf > 0.0 ? rf : 1.0);
    }
}

}  // namespace macjd

using macjd::DevTables;
using macjd::set_err;

struct macjd_scenario {
    DevTables host;
    DevTables* dev;
    int device;
};

extern "C" {

int macjd_abi_version(void) { return MACJD_ABI_VERSION; }
const char* macjd_last_error(void) { return macjd::g_err; }

void macjd_reload_options(void) { macjd::load_env_options(); }

int macjd_device_count(void) {
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "hipGetDeviceCount: %s", hipGetErrorString(err));
    return n;
}

int macjd_scenario_create(const macjd_scenario_desc* d, macjd_scenario** out) {
    if (!d || !out) return set_err(MACJD_EINVAL, "macjd_scenario_create: NULL argument");
    const int R = d->n_radars, J = d->n_jammers;
    if (R < 1 || R > MACJD_MAX_RADARS || J < 1 || J > MACJD_MAX_JAMMERS)
        return set_err(MACJD_EINVAL, "macjd_scenario_create: n_radars / n_jammers out of range");
    if (!d->radar_GaPs || !d->radar_Pn || !d->radar_D || !d->radar_pd_no || !d->radar_rd_pen || !d->radar_gr ||
        !d->jam_pmin || !d->jam_pmax || !d->jam_gj || !d->jr_denom || !d->jr_flags)
        return set_err(MACJD_EINVAL, "macjd_scenario_create: NULL table pointer");
    macjd_scenario* s = new (std::nothrow) macjd_scenario();
    if (!s) return set_err(MACJD_ENOMEM, "macjd_scenario_create: host allocation failed");
    DevTables& t = s->host;
    memset(&t, 0, sizeof(t));
    t.R = R; t.J = J; t.episode_limit = d->episode_limit;
    t.rp_min = d->rp_min; t.rp_max = d->rp_max;
    t.pd_A = d->pd_A; t.pd_c1 = d->pd_c1; t.pd_denB = d->pd_denB;
    for (int r = 0; r < R; ++r) {
        t.GaPs[r] = d->radar_GaPs[r]; t.Pn[r] = d->radar_Pn[r]; t.D[r] = d->radar_D[r];
        t.pd_no[r] = d->radar_pd_no[r]; t.rd_pen[r] = d->radar_rd_pen[r]; t.gr[r] = d->radar_gr[r];
    }
    for (int j = 0; j < J; ++j) {
        t.pmin[j] = d->jam_pmin[j]; t.pmax[j] = d->jam_pmax[j]; t.gj[j] = d->jam_gj[j];
    }
    for (int i = 0; i < J * R; ++i) { t.denom[i] = d->jr_denom[i]; t.flags[i] = d->jr_flags[i]; }
    // REGULAR scenario (env_step_kernel, REG): every quantity a division of the step can meet lies far inside the range
    // where the hardware's IEEE division needs neither operand scaling nor special cases (magnitudes in [1e-30, 1e30]
    // give quotients and residuals within 1e-220 .. 1e100), the float32 numerator cannot overflow, the noise power is
    // positive (SNR denominators >= Pn > 1e-18, SNRs >= +0), and the probability formula's argument stays above -700.
    {
        auto mag = [](double v) { return v >= 1e-30 && v <= 1e30; };
        bool ok = t.pd_denB >= 1e-9 && t.pd_denB <= 1e30 && fabs(t.pd_A) <= 1e30 && fabs(t.pd_c1) <= 1e30 &&
                  (10.0 * t.pd_c1 - t.pd_A) / t.pd_denB >= -700.0;
        double gr_max = 0.0, pg_max = 0.0;
        for (int r = 0; r < R && ok; ++r) {
            ok = mag(t.GaPs[r]) && t.Pn[r] > 1e-18 && t.Pn[r] <= 1e30 && (t.D[r] == 0.0 || mag(t.D[r])) &&
                 (t.gr[r] == 0.0 || mag(t.gr[r]));
            gr_max = t.gr[r] > gr_max ? t.gr[r] : gr_max;
        }
        for (int j = 0; j < J && ok; ++j) {
            const double range = t.pmax[j] - t.pmin[j], pm = fabs(t.pmin[j]) > fabs(t.pmax[j]) ? fabs(t.pmin[j]) : fabs(t.pmax[j]);
            ok = pm <= 1e30 && (t.gj[j] == 0.0 || mag(t.gj[j])) && (range <= 1e-6 || mag(range));
            pg_max = pm * t.gj[j] > pg_max ? pm * t.gj[j] : pg_max;
        }
        for (int i = 0; i < J * R && ok; ++i) ok = !(t.denom[i] > 1e-18) || t.denom[i] <= 1e30;
        t.regular = (ok && pg_max * gr_max < 1e30) ? 1 : 0;
    }
    hipError_t err = hipGetDevice(&s->device);
    if (err == hipSuccess) err = hipMalloc((void**)&s->dev, sizeof(DevTables));
    if (err == hipSuccess) err = hipMemcpy(s->dev, &t, sizeof(DevTables), hipMemcpyHostToDevice);
    if (err == hipSuccess) {
        hipLaunchKernelGGL(macjd::scenario_derive_kernel, dim3(1), dim3(256), 0, (hipStream_t)0, s->dev);
        err = hipGetLastError();
        if (err == hipSuccess) err = hipDeviceSynchronize();
    }
    if (err != hipSuccess) {
        if (s->dev) (void)hipFree(s->dev);
        delete s;
        return set_err(MACJD_EDEVICE, "macjd_scenario_create: %s", hipGetErrorString(err));
    }
    *out = s;
    return MACJD_OK;
}

void macjd_scenario_destroy(macjd_scenario* s) {
    if (!s) return;
    if (s->dev) (void)hipFree(s->dev);
    delete s;
}

int macjd_scenario_dims(const macjd_scenario* s, int32_t* n_radars, int32_t* n_jammers, int32_t* episode_limit) {
    if (!s) return set_err(MACJD_EINVAL, "macjd_scenario_dims: NULL scenario");
    if (n_radars) *n_radars = s->host.R;
    if (n_jammers) *n_jammers = s->host.J;
    if (episode_limit) *episode_limit = s->host.episode_limit;
    return MACJD_OK;
}

int macjd_scenario_is_regular(const macjd_scenario* s) {
    if (!s) return set_err(MACJD_EINVAL, "macjd_scenario_is_regular: NULL scenario");
    return s->host.regular;
}

int macjd_env_reset(const macjd_scenario* s, int64_t n_envs, uint8_t* track, int64_t k_se, int64_t k_sx,
                    int32_t* step, const uint8_t* mask, int32_t* episode, void* hip_stream) {
    if (!s || !track || !step || n_envs < 0) return set_err(MACJD_EINVAL, "macjd_env_reset: bad argument");
    if (n_envs == 0) return MACJD_OK;
    const int block = 256;
    int64_t grid = (n_envs + block - 1) / block;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(macjd::env_reset_kernel, dim3((unsigned)grid), dim3(block), 0, (hipStream_t)hip_stream, n_envs,
                       s->host.R, track, k_se, k_sx, step, mask, episode);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "macjd_env_reset launch: %s", hipGetErrorString(err));
    return MACJD_OK;
}

static int validate_io(const macjd_scenario* s, const macjd_step_io* io) {
    if (!s || !io) return set_err(MACJD_EINVAL, "macjd_env_step: NULL scenario / io");
    if (io->n_envs < 0) return set_err(MACJD_EINVAL, "macjd_env_step: n_envs < 0");
    if (!io->T || !io->track || !io->step) return set_err(MACJD_EINVAL, "macjd_env_step: T / track / step is NULL");
    if ((io->P32 == nullptr) == (io->P64 == nullptr))
        return set_err(MACJD_EINVAL, "macjd_env_step: exactly one of P32 / P64 must be given");
    if ((io->pd && (io->pd_se == 0 && io->pd_sx == 0)) || (io->snr_with && (io->sw_se == 0 && io->sw_sx == 0)))
        return set_err(MACJD_EINVAL, "macjd_env_step: output strides are zero");
    if (io->k_se == 0 && io->k_sx == 0) return set_err(MACJD_EINVAL, "macjd_env_step: track strides are zero");
    if (io->pe_tables && (!io->pe_flags || io->pe_tile < 0 || (io->pe_tile == 0 && io->pe_stride < io->n_envs)))
        return set_err(MACJD_EINVAL, "macjd_env_step: per-env tables need pe_flags and pe_stride >= n_envs (or pe_tile > 0)");
    return MACJD_OK;
}

static int launch_step(const macjd_scenario* s, const macjd_step_io* io, hipStream_t stream, int32_t many_T = 0,
                       int64_t t_stride = 0) {
    const int64_t E = io->n_envs * (many_T > 0 ? many_T : 1);   // work items of the launch
    const int J = s->host.J, R = s->host.R;
    const bool has_slot_kernel = (J == 3 && R == 4) || (J == 6 && R == 8) || (J == 12 && R == 16) || (J == 2 && R == 2);
    // measured crossover on MI355X (3j/4r, round 2): slot kernel 3.9 / 4.2 / 7.3 / 21.8 us vs lane kernel 5.6 / 5.7 /
    // 7.5 / 13.5 us at E = 2^12 / 2^14 / 2^16 / 2^18
    bool slot_kernel = has_slot_kernel && E < (1 << 16);
    if (io->flags & MACJD_STEP_LANE_KERNEL) slot_kernel = false;
    if ((io->flags & MACJD_STEP_SLOT_KERNEL) && has_slot_kernel) slot_kernel = true;
    const bool per_env = io->pe_tables != nullptr;
    if (per_env) slot_kernel = false;   // per-env tables: lane-per-env kernel (each lane streams its own SoA column)
    if (many_T > 0) slot_kernel = false;
    if (slot_kernel) {  // one workgroup per 64 envs
        const dim3 g((unsigned)((E + 63) / 64));
#define MACJD_SLOTS(JT, RT, NJW, NRW) \
    hipLaunchKernelGGL((macjd::env_step_slots_kernel<JT, RT, NJW, NRW>), g, dim3(64 * (NJW + NRW)), 0, stream, s->dev, *io)
        if (J == 3) MACJD_SLOTS(3, 4, 3, 4);
        else if (J == 6) MACJD_SLOTS(6, 8, 6, 8);
        else if (J == 12) MACJD_SLOTS(12, 16, 4, 8);
        else MACJD_SLOTS(2, 2, 2, 2);
#undef MACJD_SLOTS
    } else {
        // generic sizes (and the A/B hook): one lane per env
        // small batches: one wave per workgroup spreads the envs over more CUs (latency-bound regime);
        // large batches: 256-lane workgroups, grid-stride, tables staged once per workgroup.
        const int block = (E >= (1 << 16)) ? 256 : 64;
        int64_t grid = (E + block - 1) / block;
        // production variants: one env per lane, no grid-stride loop; many-step launches: grid = (envs / block, steps)
        if (many_T > 65535) return set_err(MACJD_EINVAL, "%s", "macjd_env_step_many: at most 65535 steps per launch");
        const dim3 gf((unsigned)(many_T > 0 ? (io->n_envs + block - 1) / block : grid), (unsigned)(many_T > 0 ? many_T : 1)), b(block);
        const int64_t cap = (block == 256) ? 256 * 8 : 256 * 16;
        if (grid > cap) grid = cap;
        const dim3 g((unsigned)grid);
        // production configuration (Philox uniforms, float32 actions / power arithmetic, no float64 diagnostics,
        // strides that fit int32)
        // ... and every byte offset of the launch below 2^32 (the production variant forms them in 32 bits)
        const int64_t En = io->n_envs;
        auto span_ok = [&](int64_t se, int64_t sx, int items, int64_t elem, int64_t extra = 0) {
            return se >= 0 && sx >= 0 && extra >= 0 &&
                   ((En - 1) * se + (int64_t)(items - 1) * sx + extra + 1) * elem < (int64_t)1 << 32;
        };
        const int64_t act_extra = many_T > 0 ? (int64_t)(many_T - 1) * t_stride : 0;
        const bool fast = !io->u && io->P32 && !(io->flags & MACJD_STEP_ARITH_F64) && !io->out64 && !io->pd64 &&
                          !io->snr64 && !io->prj64 && (E + block - 1) / block <= 0x7fffffff &&
                          span_ok(io->T_se, io->T_sx, J, 4, act_extra) && span_ok(io->P_se, io->P_sx, J, 4, act_extra) &&
                          span_ok(io->k_se, io->k_sx, R, 1) && (E * 3 + 3) * 4 < ((int64_t)1 << 32) && t_stride <= INT32_MAX &&
                          (!io->pd || span_ok(io->pd_se, io->pd_sx, R, 4)) &&
                          (!io->snr_with || span_ok(io->sw_se, io->sw_sx, R, 4));
        macjd::FastStepIO f{};
        if (fast) {
            f.n_envs = io->n_envs; f.env_offset = io->env_offset; f.seed = io->seed;   // (n_envs: real envs, not work items)
            f.T = io->T; f.P32 = io->P32; f.episode = io->episode; f.track = io->track; f.step = io->step;
            f.reward = io->reward; f.r_dpj = io->r_dpj; f.terminated = io->terminated; f.pd = io->pd;
            f.snr_with = io->snr_with; f.r_dpj_sum = io->r_dpj_sum; f.pe_tables = io->pe_tables;
            f.pe_flags = io->pe_flags; f.pe_stride = io->pe_stride; f.pe_tile = io->pe_tile;
            f.T_se = (int32_t)io->T_se; f.T_sx = (int32_t)io->T_sx; f.P_se = (int32_t)io->P_se; f.P_sx = (int32_t)io->P_sx;
            f.k_se = (int32_t)io->k_se; f.k_sx = (int32_t)io->k_sx; f.pd_se = (int32_t)io->pd_se; f.pd_sx = (int32_t)io->pd_sx;
            f.sw_se = (int32_t)io->sw_se; f.sw_sx = (int32_t)io->sw_sx;
            f.many_T = many_T; f.t_stride = (int32_t)t_stride;
        } else if (many_T > 0) {
            return set_err(MACJD_EUNSUPPORTED, "%s", "macjd_env_step_many: production configuration only (Philox uniforms, "
                           "float32 actions, no float64 diagnostics, offsets below 2^32 bytes)");
        }
        const bool no_reg = !macjd::env_options().regular;   // MACJD_ENV_REGULAR=0: keep IEEE divisions + all guards
        const bool pd32 = macjd::env_options().pd32;          // MACJD_ENV_PD32=0: all-float64 detection probabilities
#define MACJD_LAUNCH(JT, RT)                                                                                          \
    do {                                                                                                              \
        if (per_env && fast)                                                                                          \
            hipLaunchKernelGGL((macjd::env_step_kernel<JT, RT, true, true, macjd::FastStepIO>), gf, b, 0, stream, s->dev, f);      \
        else if (per_env)                                                                                             \
            hipLaunchKernelGGL((macjd::env_step_kernel<JT, RT, true, false, macjd_step_io>), g, b, 0, stream, s->dev, *io);        \
        else if (fast && s->host.regular && !no_reg && pd32)                                                          \
            hipLaunchKernelGGL((macjd::env_step_kernel<JT, RT, false, true, macjd::FastStepIO, true, true>), gf, b, 0, stream, s->dev, f); \
        else if (fast && s->host.regular && !no_reg)                                                                  \
            hipLaunchKernelGGL((macjd::env_step_kernel<JT, RT, false, true, macjd::FastStepIO, true>), gf, b, 0, stream, s->dev, f); \
        else if (fast)                                                                                                \
            hipLaunchKernelGGL((macjd::env_step_kernel<JT, RT, false, true, macjd::FastStepIO>), gf, b, 0, stream, s->dev, f);     \
        else                                                                                                          \
            hipLaunchKernelGGL((macjd::env_step_kernel<JT, RT, false, false, macjd_step_io>), g, b, 0, stream, s->dev, *io);       \
    } while (0)
        if (J == 3 && R == 4) MACJD_LAUNCH(3, 4);
        else if (J == 6 && R == 8) MACJD_LAUNCH(6, 8);
        else if (J == 12 && R == 16) MACJD_LAUNCH(12, 16);
        else if (J == 2 && R == 2) MACJD_LAUNCH(2, 2);
        else if (many_T > 0) {
            return set_err(MACJD_EUNSUPPORTED, "%s", "macjd_env_step_many: scenario size without a compiled production variant");
        } else {   // generic sizes: one (non-FAST) variant per table mode
            if (per_env) hipLaunchKernelGGL((macjd::env_step_kernel<0, 0, true, false, macjd_step_io>), g, b, 0, stream, s->dev, *io);
            else hipLaunchKernelGGL((macjd::env_step_kernel<0, 0, false, false, macjd_step_io>), g, b, 0, stream, s->dev, *io);
        }
#undef MACJD_LAUNCH
    }
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "macjd_env_step launch: %s", hipGetErrorString(err));
    return MACJD_OK;
}

int macjd_env_step(const macjd_scenario* s, const macjd_step_io* io, void* hip_stream) {
    int rc = validate_io(s, io);
    if (rc != MACJD_OK) return rc;
    if (io->n_envs == 0) return MACJD_OK;
    return launch_step(s, io, (hipStream_t)hip_stream);
}

int macjd_env_step_many(const macjd_scenario* s, const macjd_step_io* io, int32_t n_steps, int64_t t_stride, void* hip_stream) {
    int rc = validate_io(s, io);
    if (rc != MACJD_OK) return rc;
    if (n_steps < 1 || t_stride < 0) return set_err(MACJD_EINVAL, "%s", "macjd_env_step_many: bad n_steps / t_stride");
    if (io->pd || io->snr_with) return set_err(MACJD_EINVAL, "%s", "macjd_env_step_many: pd / snr_with are not written (pass NULL)");
    if (io->r_dpj_sum && !io->r_dpj) return set_err(MACJD_EINVAL, "%s", "macjd_env_step_many: r_dpj_sum needs the per-step r_dpj buffer");
    if (io->n_envs == 0) return MACJD_OK;
    rc = launch_step(s, io, (hipStream_t)hip_stream, n_steps, t_stride);
    if (rc != MACJD_OK) return rc;
    int64_t grid = (io->n_envs * 3 + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(macjd::env_advance_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)hip_stream, io->n_envs,
                       n_steps, io->step, io->r_dpj, io->r_dpj_sum);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "macjd_env_step_many launch: %s", hipGetErrorString(err));
    return MACJD_OK;
}

static int env_step_timed_impl(const macjd_scenario* s, const macjd_step_io* io, int iters, void* hip_stream,
                               float* ms_per_launch, int32_t many_T, int64_t t_stride);

int macjd_env_step_timed(const macjd_scenario* s, const macjd_step_io* io, int iters, void* hip_stream,
                         float* ms_per_launch) {
    return env_step_timed_impl(s, io, iters, hip_stream, ms_per_launch, 0, 0);
}

int macjd_env_step_many_timed(const macjd_scenario* s, const macjd_step_io* io, int32_t n_steps, int64_t t_stride, int iters,
                              void* hip_stream, float* ms_per_launch) {
    if (n_steps < 1 || t_stride < 0) return set_err(MACJD_EINVAL, "%s", "macjd_env_step_many_timed: bad n_steps / t_stride");
    return env_step_timed_impl(s, io, iters, hip_stream, ms_per_launch, n_steps, t_stride);
}

static int env_step_timed_impl(const macjd_scenario* s, const macjd_step_io* io, int iters, void* hip_stream,
                               float* ms_per_launch, int32_t many_T, int64_t t_stride) {
    int rc = validate_io(s, io);
    if (rc != MACJD_OK) return rc;
    if (iters < 1 || !ms_per_launch) return set_err(MACJD_EINVAL, "macjd_env_step_timed: bad iters / output");
    hipStream_t stream = (hipStream_t)hip_stream;
    hipEvent_t t0, t1;
    hipError_t err = hipEventCreate(&t0);
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "hipEventCreate: %s", hipGetErrorString(err));
    err = hipEventCreate(&t1);
    if (err != hipSuccess) { (void)hipEventDestroy(t0); return set_err(MACJD_EDEVICE, "hipEventCreate: %s", hipGetErrorString(err)); }
    // The launches are replayed from ONE HIP graph (as in the benchmark's rollout), so the figure is the kernel's
    // back-to-back launch duration on the GPU and not the host's enqueue rate (4 - 9 us per launch, box-dependent,
    // for a 4 us kernel).  If the capture is refused the launches are issued directly.
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t priv = nullptr;
    bool graphed = false;
    // capture + replay on a private stream (the caller's may be the null stream, which cannot be captured), after
    // the caller's earlier work has finished; the events are recorded on the stream the launches run on
    if (hipStreamSynchronize(stream) == hipSuccess && hipStreamCreateWithFlags(&priv, hipStreamNonBlocking) == hipSuccess) {
        if (hipStreamBeginCapture(priv, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            int crc = MACJD_OK;
            for (int i = 0; i < iters && crc == MACJD_OK; ++i) crc = launch_step(s, io, priv, many_T, t_stride);
            const hipError_t ce = hipStreamEndCapture(priv, &graph);
            if (ce == hipSuccess && crc == MACJD_OK && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess)
                graphed = hipGraphLaunch(exec, priv) == hipSuccess && hipStreamSynchronize(priv) == hipSuccess;   // warm
        }
        (void)hipGetLastError();
        if (graphed) stream = priv;
    }
    (void)hipEventRecord(t0, stream);
    if (graphed) {
        if (hipGraphLaunch(exec, stream) != hipSuccess) rc = set_err(MACJD_EDEVICE, "%s", "macjd_env_step_timed: graph launch failed");
    } else {
        for (int i = 0; i < iters && rc == MACJD_OK; ++i) rc = launch_step(s, io, stream, many_T, t_stride);
    }
    (void)hipEventRecord(t1, stream);
    err = hipEventSynchronize(t1);
    // the graph is destroyed with its stream drained (the runtime defers the release of a launched graph to the launch
    // stream's next synchronisation point; that stream is destroyed right below)
    if (priv) (void)hipStreamSynchronize(priv);
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    if (priv) (void)hipStreamDestroy(priv);
    float ms = 0.f;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, t0, t1);
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    if (rc != MACJD_OK) return rc;
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "macjd_env_step_timed: %s", hipGetErrorString(err));
    *ms_per_launch = ms / (float)iters;
    return MACJD_OK;
}

}  // extern "C"

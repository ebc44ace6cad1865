// macjd_episode.hip — the agent side of a whole episode batch as ONE launch (C-ABI: include/macjd_nets.h,
// macjd_agent_episode_io): for every step t, h_t = GRUCell(x, h_{t-1}); Q(h_t, a, P_a) for all actions (MP-DQN
// multi-pass Q-head); mask; epsilon-greedy; gather of the chosen power.  Reference: core/mac.py:59-166 driven by
// runners/episode_runner.py:27-181, one env step at a time; here T steps of E environments without returning to the
// host or to a launch boundary.
//
// Why this is legal: the observation of this environment is static (simulation/environment.py:479-522), so the GRU's
// input transform gi = W_ih ReLU(fc1 obs) + b_ih and the actor's output P are per-episode constants, and nothing the
// agent computes at step t depends on what the environment returned at step t-1.  Environments are independent of
// each other as well: a workgroup that owns 16 environments never needs anything from another workgroup.
//
// Mapping (J agents, H = 64 hidden units, A actions):
//   * workgroup = 16 consecutive environments = J MFMA row tiles (tile j = agent j, row = env within the group);
//     4 waves, one per SIMD; grid = E / 16 (256 workgroups at the benchmark's E = 4096: the whole chip);
//   * wave w owns hidden units U_w = [16w, 16w+16) of all three gates: its 3 x 4 W_hh fragments (k = 64 -> 4 quads)
//     and the 4 fragments of the Q-head's h-columns stay in VGPRs for the whole episode (64 registers);
//   * per step: (1) gh tiles (r, z, n) x J by v_mfma_f32_16x16x4_f32, A operand = h_{t-1} from LDS (ds_read_b128,
//     permuted-k quads as in macjd_mlp.hip); (2) gates on the accumulator tiles in registers (the wave holds r, z, n of
//     the SAME hidden units), h_t -> its own registers, the other LDS tile (ping-pong) and the staging row in HBM;
//     barrier; (3) base = W1[:, :H] h_t + b1 for the wave's 16 Q-head units -> LDS; barrier; (4) all-action Q-head:
//     wave j = agent tile j, lane = (row, quarter of the 64 units): 16 units x A actions per lane with the action /
//     power columns of W1 and w2 read from LDS (broadcast), quarter sums by two xor shuffles; (5) the lanes of
//     quarter 0 select: availability mask, first arg-max, epsilon-greedy with the Philox draw of (row, counter_base +
//     t + 1) — the keying of macjd_qhead_select — and write the chosen action / power of (t, env, agent).
//   Two barriers per step; LDS: 2 x J x 16 x 72 floats of h, J x 16 x 72 of base, (A + 2) x 64 of Q-head columns.
// Numerics: exact float32 everywhere; the gate expressions are those of gru_gates_kernel (IEEE division, expf), the
// Q-head expressions those of qhead_select_kernel; only the summation order of the two products differs from the
// step-by-step path (library GEMM + split over 8 waves), i.e. ~1e-7 relative on h and Q.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/macjd.h"
#include "../../include/macjd_nets.h"
#include "macjd_err.h"
#include "macjd_philox.h"

namespace macjd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int EP_H = 64;
constexpr int EP_LD = EP_H + 8;   // LDS pitch (= 8 mod 16 floats: conflict-free ds_read_b128 fragments)
constexpr int EP_KQ = EP_H / 16;  // quads of a K = 64 product

// FLAT (wide scenarios, e.g. 12 jammers): nothing above depends on WHICH agent a row belongs to — the networks are shared
// and every output is indexed by the flattened row n = env * n_agents + agent — so the workgroup simply owns J = 4 tiles
// of 16 consecutive rows n (64 rows: 36 KB of LDS instead of 110 KB for 12 agent tiles); same arithmetic per row.
template <int J, int A, bool FLAT = false>
__global__ void __launch_bounds__(256) agent_episode_kernel(const macjd_agent_episode_io io) {
    static_assert(J <= 8, "one wave per agent tile in the Q-head phase, two passes above 4");
    __shared__ __attribute__((aligned(16))) float Hl[2][J][16 * EP_LD];   // h_{t-1} / h_t, ping-pong
    __shared__ __attribute__((aligned(16))) float Bl[J][16 * EP_LD];      // Q-head base of the current step
    __shared__ float Wq[(A + 2) * EP_H];   // Q-head: action columns W1[u][H + a] at [a][u], power column at [A][u], w2 at [A+1][u]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int64_t e0 = (int64_t)blockIdx.x * 16;
    const int T = io.T;
    const int64_t E = io.n_envs;
    const int64_t NA = io.J;                            // agents per env (FLAT: run-time; otherwise == J)
    const int64_t N = E * NA;                           // rows
    const int64_t n0 = (int64_t)blockIdx.x * (16 * J);  // FLAT: first row of the workgroup
    // row of (tile j, row-in-tile r), clamped for the loads; `live`: the row exists
    auto row_of = [&](int j, int r, bool& live) -> int64_t {
        if (FLAT) {
            const int64_t n = n0 + 16 * j + r;
            live = n < N;
            return live ? n : N - 1;
        }
        const int64_t e = e0 + r;
        live = e < E;
        return (live ? e : E - 1) * J + j;
    };

    // ---- episode constants into registers / LDS -------------------------------------------------------------
    // weight fragments (B operands): rows gate * 64 + 16 wave + li of W_hh, row 16 wave + li of W1's h-columns
    f32x4 Bh[3][EP_KQ], Bq[EP_KQ];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int Q = 0; Q < EP_KQ; ++Q)
            Bh[c][Q] = *reinterpret_cast<const f32x4*>(io.w_hh + (int64_t)(c * EP_H + 16 * wave + li) * EP_H + 16 * Q + 4 * g);
#pragma unroll
    for (int Q = 0; Q < EP_KQ; ++Q)
        Bq[Q] = *reinterpret_cast<const f32x4_u*>(io.W1 + (int64_t)(16 * wave + li) * io.w1_ld + 16 * Q + 4 * g);
    const int u = 16 * wave + li;          // this lane's hidden unit (gates) / Q-head unit (base) in the C layout
    float bhh[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) bhh[c] = io.b_hh[c * EP_H + u];
    const float b1u = io.b1[u];
    // C layout of a 16 x 16 tile: lane holds rows 4g + r (r = 0..3) of column li.  gi of (tile j, row 4g + r), unit u:
    float giv[J][4][3], hreg[J][4];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bool live;
            const int64_t n = row_of(j, 4 * g + r, live);           // clamped: rows past the end are computed, never stored
#pragma unroll
            for (int c = 0; c < 3; ++c) giv[j][r][c] = io.gi[n * io.gi_ld + c * EP_H + u];
            hreg[j][r] = io.h0 ? io.h0[n * EP_H + u] : 0.0f;
            Hl[0][j][(4 * g + r) * EP_LD + u] = hreg[j][r];
        }
    for (int idx = threadIdx.x; idx < (A + 2) * EP_H; idx += 256) {
        const int a = idx / EP_H, uu = idx - a * EP_H;
        Wq[idx] = (a <= A) ? io.W1[(int64_t)uu * io.w1_ld + EP_H + a] : io.w2[uu];
    }
    const float b2 = io.b2[0];
    // Q-head phase roles: wave = agent tile (waves >= J idle there; J > 4: a second pass), lane = (row qr, quarter qq)
    const int qr = lane & 15, qq = lane >> 4;
    float pv[(J + 3) / 4][A];         // the row's actor output P[a] (static within the episode)
    uint64_t avail_bits[(J + 3) / 4];
    int n_avail[(J + 3) / 4];
#pragma unroll
    for (int pass = 0; pass < (J + 3) / 4; ++pass) {
        const int j = wave + 4 * pass;
        bool live_q;
        const int64_t n = row_of(j < J ? j : 0, qr, live_q);
        const int64_t ec = n / NA, jc = n - ec * NA;      // (env, agent) of the row, for the availability mask
        avail_bits[pass] = 0;
        n_avail[pass] = 0;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            pv[pass][a] = io.P_all[n * io.p_ld + a];
            bool av = true;
            if (io.avail) {
                const int64_t off = ec * io.av_se + jc * io.av_sj + (int64_t)a * io.av_sa;
                av = (io.avail_elem_size == 8) ? (((const int64_t*)io.avail)[off] != 0) : (((const int32_t*)io.avail)[off] != 0);
            }
            avail_bits[pass] |= av ? (1ull << a) : 0ull;
            n_avail[pass] += av ? 1 : 0;
        }
    }
    const uint64_t ctr_base = io.counter_base ? io.counter_base[0] : 0ull;
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        // ---- (1) gh = W_hh h_{t-1} for this wave's units, all gates, all agent tiles ----
        f32x4 acc[J][3];
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int Q = 0; Q < EP_KQ; ++Q) {
            f32x4 a[J];
#pragma unroll
            for (int j = 0; j < J; ++j) a[j] = *reinterpret_cast<const f32x4*>(&Hl[cur][j][li * EP_LD + 16 * Q + 4 * g]);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        acc[j][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][jj], Bh[c][Q][jj], acc[j][c], 0, 0, 0);
        }
        // ---- (2) gates (torch.nn.GRUCell, gate order r, z, n; expressions of gru_gates_kernel) ----
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float hr = acc[j][0][r] + bhh[0], hz = acc[j][1][r] + bhh[1], hn = acc[j][2][r] + bhh[2];
                const float rg = 1.0f / (1.0f + expf(-(giv[j][r][0] + hr)));
                const float zg = 1.0f / (1.0f + expf(-(giv[j][r][1] + hz)));
                const float nn = 1.0f - 2.0f / (expf(2.0f * (giv[j][r][2] + rg * hn)) + 1.0f);
                const float hnew = (hreg[j][r] - nn) * zg + nn;
                hreg[j][r] = hnew;
                const int row = 4 * g + r;
                Hl[nxt][j][row * EP_LD + u] = hnew;
                bool live;
                const int64_t n = row_of(j, row, live);
                if (live) io.hidden[((int64_t)t * N + n) * EP_H + u] = hnew;   // staging row t: post-update h_t
            }
        __syncthreads();
        // ---- (3) Q-head base = W1[:, :H] h_t + b1 for this wave's 16 units ----
        {
            f32x4 ab[J];
#pragma unroll
            for (int j = 0; j < J; ++j) ab[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int Q = 0; Q < EP_KQ; ++Q) {
                f32x4 a[J];
#pragma unroll
                for (int j = 0; j < J; ++j) a[j] = *reinterpret_cast<const f32x4*>(&Hl[nxt][j][li * EP_LD + 16 * Q + 4 * g]);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int j = 0; j < J; ++j) ab[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][jj], Bq[Q][jj], ab[j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) Bl[j][(4 * g + r) * EP_LD + u] = ab[j][r] + b1u;
        }
        __syncthreads();
        // ---- (4) all-action Q-head + (5) selection: wave = agent tile ----
        const float epsilon = io.greedy_only ? 0.0f : io.eps[t];
#pragma unroll
        for (int pass = 0; pass < (J + 3) / 4; ++pass) {
            const int j = wave + 4 * pass;   // wave-uniform
            if (j < J) {
                float q[A];
#pragma unroll
                for (int a = 0; a < A; ++a) q[a] = 0.0f;
                const float* brow = &Bl[j][qr * EP_LD + 16 * qq];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(brow + 4 * k4);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int uu = 16 * qq + 4 * k4 + kk;
                        const float wp = Wq[A * EP_H + uu], w2u = Wq[(A + 1) * EP_H + uu];
#pragma unroll
                        for (int a = 0; a < A; ++a) {
                            float v = b4[kk] + Wq[a * EP_H + uu];          // (W_h h + b1)[u] + W1[u, H + a]
                            v = fmaf(pv[pass][a], wp, v);                  // + W1[u, H + A] * P_a
                            v = fmaxf(v, 0.0f);                            // ReLU (networks.py:77)
                            q[a] = fmaf(v, w2u, q[a]);                     // second layer (networks.py:78)
                        }
                    }
                }
#pragma unroll
                for (int a = 0; a < A; ++a) {
                    q[a] += __shfl_xor(q[a], 16, 64);
                    q[a] += __shfl_xor(q[a], 32, 64);
                    q[a] += b2;
                }
                bool live;
                const int64_t n = row_of(j, qr, live);
                if (qq == 0 && live) {
                    // mask, first arg-max, epsilon-greedy (mac.py:142-146, action_selectors.py:34-62): as qhead_select_kernel
                    int best = 0;
                    float bestq = -INFINITY;
#pragma unroll
                    for (int a = 0; a < A; ++a) {
                        const float qa = ((avail_bits[pass] >> a) & 1ull) ? q[a] : -INFINITY;
                        if (qa > bestq) { bestq = qa; best = a; }
                    }
                    int chosen = best;
                    if (epsilon > 0.0f) {
                        const uint64_t counter = ctr_base + (uint64_t)(t + 1);
                        const Philox4 rr = philox4x32_10((uint32_t)n, (uint32_t)((uint64_t)n >> 32), (uint32_t)counter,
                                                         (uint32_t)(counter >> 32), (uint32_t)io.seed, (uint32_t)(io.seed >> 32));
                        const float u_pick = (float)(rr.v[0] >> 8) * (1.0f / 16777216.0f);
                        if (u_pick < epsilon) {
                            const int pool = n_avail[pass] > 0 ? n_avail[pass] : A;
                            int k = (int)(((uint64_t)rr.v[1] * (uint64_t)pool) >> 32);
                            chosen = 0;
#pragma unroll
                            for (int a = 0; a < A; ++a) {
                                const bool av = (n_avail[pass] > 0) ? ((avail_bits[pass] >> a) & 1ull) : true;
                                if (av) { if (k == 0) chosen = a; --k; }
                            }
                        }
                    }
                    float pc = 0.0f;
#pragma unroll
                    for (int a = 0; a < A; ++a) pc = (a == chosen) ? pv[pass][a] : pc;
                    const int64_t o = (int64_t)t * N + n;
                    io.T_out[o] = chosen;
                    io.P_out[o] = pc;
                }
            }
        }
        // no barrier here: the next step's (1) reads Hl[nxt] (complete since the barrier after (2)) and its (2) writes
        // Hl[cur], which nobody reads any more; its (3) overwrites Bl only after the barrier that follows its (2), which
        // every wave passes after finishing this step's (4)
    }
    if (io.h_final) {
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bool live;
                const int64_t n = row_of(j, 4 * g + r, live);
                if (live) io.h_final[n * EP_H + u] = hreg[j][r];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Double-DQN target values from hidden states (include/macjd_nets.h, macjd_doubleq_io): workgroup = 16 rows; wave w
// computes the Q-head base columns [16w, 16w+16) of BOTH networks (8 quads of MFMA each), then all four waves evaluate
// the heads for all actions — wave w: head (w & 1) (0 = eval, 1 = target), action half (w >> 1), lane = (row, quarter of
// the 64 units) as in agent_episode_kernel — and the first arg-max of the eval head (its two halves meet through LDS)
// picks the target head's value.  (Round 2 used two of the four waves for the all-action phase: 156 us at A = 33.)
template <int A>
struct DoubleQLds {
    alignas(16) float Hs[2][16 * EP_LD];    // h rows of the eval / target unroll
    alignas(16) float Bs[2][16 * EP_LD];    // base of the eval / target head
    float Wq[2][(A + 2) * EP_H];
    float amax_q[2][16];
    int amax_i[2][16];
};

// (the body is a device function of (arguments, workgroup index, LDS) so that macjd_qheads_pair can run it beside the
// taken-action Q-head in ONE launch)
template <int A>
__device__ __forceinline__ void qhead_double_q_body(const macjd_doubleq_io& io, const int blk, DoubleQLds<A>& L) {
    constexpr int AH = (A + 1) / 2;                                      // actions per half (the second half may hold one less)
    auto& Hs = L.Hs;
    auto& Bs = L.Bs;
    auto& Wq = L.Wq;
    auto& amax_q = L.amax_q;
    auto& amax_i = L.amax_i;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int64_t n0 = (int64_t)blk * 16;
    const bool same_h = (io.h_e == io.h_t) && (io.he_ld == io.ht_ld);
    // weight fragments + this lane's bias of both heads
    f32x4 Be[EP_KQ], Bt[EP_KQ];
#pragma unroll
    for (int Q = 0; Q < EP_KQ; ++Q) {
        Be[Q] = *reinterpret_cast<const f32x4_u*>(io.W1_e + (int64_t)(16 * wave + li) * io.w1e_ld + 16 * Q + 4 * g);
        Bt[Q] = *reinterpret_cast<const f32x4_u*>(io.W1_t + (int64_t)(16 * wave + li) * io.w1t_ld + 16 * Q + 4 * g);
    }
    const int u = 16 * wave + li;
    const float b1e = io.b1_e[u], b1t = io.b1_t[u];
    // all-action roles: head = wave & 1, actions [a0, a0 + na); lane = (row qr, quarter qq)
    const int head = wave & 1, a0 = (wave >> 1) * AH, na = (wave >> 1) ? A - AH : AH;
    const int qr = lane & 15, qq = lane >> 4;
    const int64_t nq = (n0 + qr < io.n_rows) ? n0 + qr : io.n_rows - 1;
    float pv[AH];
    {
        const int64_t np = io.p_group > 0 ? (nq / io.p_group) * io.p_inner + nq % io.p_inner : nq;   // one P row per sequence
        const float* prow = head ? io.P_t + np * io.pt_ld : io.P_e + np * io.pe_ld;
#pragma unroll
        for (int a = 0; a < AH; ++a) pv[a] = prow[a0 + (a < na ? a : na - 1)];
    }
    const float b2 = head ? io.b2_t[0] : io.b2_e[0];
    // h rows -> LDS (float4 pieces when aligned; rows past n: the last row, never stored)
    for (int idx = threadIdx.x; idx < 2 * 16 * (EP_H / 4); idx += 256) {
        const int which = idx / (16 * (EP_H / 4)), rem = idx - which * 16 * (EP_H / 4);
        const int row = rem / (EP_H / 4), c4 = rem - row * (EP_H / 4);
        if (which == 1 && same_h) continue;
        const int64_t nn = (n0 + row < io.n_rows) ? n0 + row : io.n_rows - 1;
        const float* src = (which ? io.h_t + nn * io.ht_ld : io.h_e + nn * io.he_ld) + 4 * c4;
        *reinterpret_cast<f32x4*>(&Hs[which][row * EP_LD + 4 * c4]) = *reinterpret_cast<const f32x4_u*>(src);
    }
    // (consecutive lanes walk along a ROW of W1 — its A + 1 action / power columns are contiguous — rather than down a
    // column: a column walk is one cache line per element, 2 x 64 x (A + 1) line requests per workgroup)
    for (int idx = threadIdx.x; idx < 2 * (A + 2) * EP_H; idx += 256) {
        const int which = idx / ((A + 2) * EP_H), rem = idx - which * (A + 2) * EP_H;
        const int uu = rem / (A + 2), a = rem - uu * (A + 2);
        const float* W1 = which ? io.W1_t : io.W1_e;
        const int64_t ld = which ? io.w1t_ld : io.w1e_ld;
        Wq[which][a * EP_H + uu] = (a <= A) ? W1[(int64_t)uu * ld + EP_H + a] : (which ? io.w2_t : io.w2_e)[uu];
    }
    __syncthreads();
    {
        f32x4 ae = f32x4{0.f, 0.f, 0.f, 0.f}, at = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* he = &Hs[0][li * EP_LD + 4 * g];
        const float* ht = &Hs[same_h ? 0 : 1][li * EP_LD + 4 * g];
#pragma unroll
        for (int Q = 0; Q < EP_KQ; ++Q) {
            const f32x4 xe = *reinterpret_cast<const f32x4*>(he + 16 * Q);
            const f32x4 xt = *reinterpret_cast<const f32x4*>(ht + 16 * Q);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                ae = __builtin_amdgcn_mfma_f32_16x16x4f32(xe[jj], Be[Q][jj], ae, 0, 0, 0);
                at = __builtin_amdgcn_mfma_f32_16x16x4f32(xt[jj], Bt[Q][jj], at, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Bs[0][(4 * g + r) * EP_LD + u] = ae[r] + b1e;
            Bs[1][(4 * g + r) * EP_LD + u] = at[r] + b1t;
        }
    }
    __syncthreads();
    float q[AH];
    {
        const float* wq = Wq[head];
#pragma unroll
        for (int a = 0; a < AH; ++a) q[a] = 0.0f;
        const float* brow = &Bs[head][qr * EP_LD + 16 * qq];
        // (33 actions: fully unrolled, the compiler hoists all 16 x 19 LDS reads of the lane's units in front of the
        // arithmetic — 256 VGPRs + spills, one workgroup per CU, 118 us for 38 784 rows; unit groups of four one after
        // the other keep ~80 values live)
        constexpr int K4_UNROLL = A > 17 ? 1 : 4;
#pragma unroll K4_UNROLL
        for (int k4 = 0; k4 < 4; ++k4) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(brow + 4 * k4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int uu = 16 * qq + 4 * k4 + kk;
                const float wp = wq[A * EP_H + uu], w2u = wq[(A + 1) * EP_H + uu];
#pragma unroll
                for (int a = 0; a < AH; ++a) {
                    float v = b4[kk] + wq[(a0 + (a < na ? a : na - 1)) * EP_H + uu];
                    v = fmaf(pv[a], wp, v);
                    v = fmaxf(v, 0.0f);
                    q[a] = fmaf(v, w2u, q[a]);
                }
            }
        }
#pragma unroll
        for (int a = 0; a < AH; ++a) {
            q[a] += __shfl_xor(q[a], 16, 64);
            q[a] += __shfl_xor(q[a], 32, 64);
            q[a] += b2;
        }
        if (head == 0 && qq == 0) {   // first maximum of this half's unmasked eval values (qmix.py:138-143)
            int am = 0;
            float aq = q[0];
#pragma unroll
            for (int a = 1; a < AH; ++a)
                if (a < na && q[a] > aq) { aq = q[a]; am = a; }
            amax_q[wave >> 1][qr] = aq;
            amax_i[wave >> 1][qr] = a0 + am;
        }
    }
    __syncthreads();
    // the first maximum over both halves: the second half wins only with a strictly larger value
    const int am = (amax_q[1][qr] > amax_q[0][qr]) ? amax_i[1][qr] : amax_i[0][qr];
    if (wave == 0 && qq == 0 && io.argmax_out && n0 + qr < io.n_rows) io.argmax_out[n0 + qr] = am;
    if (head == 1 && qq == 0 && n0 + qr < io.n_rows && am >= a0 && am < a0 + na) {   // qmix.py:147
        float gq = 0.0f;
#pragma unroll
        for (int a = 0; a < AH; ++a) gq = (a0 + a == am) ? q[a] : gq;
        io.out[n0 + qr] = gq;
    }
}

template <int A>
__global__ void __launch_bounds__(256) qhead_double_q_kernel(const macjd_doubleq_io io) {
    __shared__ DoubleQLds<A> L;
    qhead_double_q_body<A>(io, blockIdx.x, L);
}

// ---------------------------------------------------------------------------------------------
// Taken-action Q-values of the learner in one launch (include/macjd_nets.h, macjd_qtaken_io).
constexpr int QT_TILES = 2;       // 16-row MFMA tiles per workgroup: 32 rows (303 workgroups for the update's 9 696 rows)
constexpr int QT_AMAX = 64;

struct QtakenLds {
    alignas(16) float Hl[QT_TILES][16 * EP_LD];   // h rows of the workgroup
    float Wc[(QT_AMAX + 1) * EP_H];               // W1[u][H + a] at [a][u], a = A: power column
    float Qp[4][QT_TILES * 16];                   // per-wave partial sums of q
    int s_idx[QT_TILES * 16];
    float s_P[QT_TILES * 16];
};

__device__ __forceinline__ void qhead_taken_body(const macjd_qtaken_io& io, const int blk, QtakenLds& L) {
    auto& Hl = L.Hl;
    auto& Wc = L.Wc;
    auto& Qp = L.Qp;
    auto& s_idx = L.s_idx;
    auto& s_P = L.s_P;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int A = io.A, W = EP_H + A + 1;
    const int64_t r0 = (int64_t)blk * (16 * QT_TILES);
    // weight fragments of this wave's 16 units (B operands), bias, second-layer weight
    f32x4 Bq[EP_KQ];
#pragma unroll
    for (int Q = 0; Q < EP_KQ; ++Q)
        Bq[Q] = *reinterpret_cast<const f32x4_u*>(io.W1 + (int64_t)(16 * wave + li) * io.w1_ld + 16 * Q + 4 * g);
    const int u = 16 * wave + li;
    const float b1u = io.b1[u], w2u = io.w2[u];
    // ---- stage: h rows (float4), the action / power columns of W1, the rows' action index and power ----
    for (int i = threadIdx.x; i < QT_TILES * 16 * (EP_H / 4); i += 256) {
        const int row = i / (EP_H / 4), c4 = i - row * (EP_H / 4);
        const int64_t rc = (r0 + row < io.n_rows) ? r0 + row : io.n_rows - 1;   // clamped: rows past n are computed, never stored
        *reinterpret_cast<f32x4*>(&Hl[row >> 4][(row & 15) * EP_LD + 4 * c4]) =
            *reinterpret_cast<const f32x4*>(io.h + rc * io.h_ld + 4 * c4);
    }
    for (int i = threadIdx.x; i < (A + 1) * EP_H; i += 256) {   // (along W1's rows, as in the Double-DQN body)
        const int uu = i / (A + 1), a = i - uu * (A + 1);
        Wc[a * EP_H + uu] = io.W1[(int64_t)uu * io.w1_ld + EP_H + a];
    }
    if (threadIdx.x < QT_TILES * 16) {
        const int64_t rc = (r0 + threadIdx.x < io.n_rows) ? r0 + threadIdx.x : io.n_rows - 1;
        const int64_t a = (io.idx_elem_size == 8) ? ((const int64_t*)io.idx)[rc] : (int64_t)((const int32_t*)io.idx)[rc];
        s_idx[threadIdx.x] = (a >= 0 && a < A) ? (int)a : -1;
        s_P[threadIdx.x] = io.P[rc];
    }
    __syncthreads();
    // ---- the Q-head's input rows [h, onehot(a), P] (the backward's weight-gradient operand) ----
    if (io.x) {
        for (int i = threadIdx.x; i < QT_TILES * 16 * W; i += 256) {
            const int row = i / W, c = i - row * W;
            if (r0 + row < io.n_rows) {
                const float v = (c < EP_H) ? Hl[row >> 4][(row & 15) * EP_LD + c]
                                           : (c < EP_H + A) ? ((s_idx[row] == c - EP_H) ? 1.0f : 0.0f) : s_P[row];
                io.x[(r0 + row) * io.x_ld + c] = v;
            }
        }
    }
    // ---- base = W1[:, :H] h on MFMA, then the one-hot / power columns, bias, ReLU, and the second layer's dot ----
#pragma unroll
    for (int tile = 0; tile < QT_TILES; ++tile) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int Q = 0; Q < EP_KQ; ++Q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&Hl[tile][li * EP_LD + 16 * Q + 4 * g]);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj], Bq[Q][jj], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // C layout: this lane holds rows 4g + r of column (unit) li
            const int row = 16 * tile + 4 * g + r;
            const int a = s_idx[row];
            const float v = ((acc[r] + b1u) + (a >= 0 ? Wc[a * EP_H + u] : 0.0f)) + Wc[A * EP_H + u] * s_P[row];
            const float act = v > 0.0f ? v : 0.0f;
            if (r0 + row < io.n_rows) io.act[(r0 + row) * io.act_ld + u] = act;
            float qp = act * w2u;       // sum over the wave's 16 units: the 16 lanes li of this row group
            qp += __shfl_xor(qp, 1, 64);
            qp += __shfl_xor(qp, 2, 64);
            qp += __shfl_xor(qp, 4, 64);
            qp += __shfl_xor(qp, 8, 64);
            if (li == 0) Qp[wave][row] = qp;
        }
    }
    __syncthreads();
    if (threadIdx.x < QT_TILES * 16 && r0 + threadIdx.x < io.n_rows) {
        const int row = threadIdx.x;
        io.q[r0 + row] = ((Qp[0][row] + Qp[1][row]) + (Qp[2][row] + Qp[3][row])) + (io.b2 ? io.b2[0] : 0.0f);
    }
}

__global__ void __launch_bounds__(256) qhead_taken_kernel(const macjd_qtaken_io io) {
    __shared__ QtakenLds L;
    qhead_taken_body(io, blockIdx.x, L);
}

// Both Q-head launches of one learner update — the taken-action values of the eval head (autograd's forward) and the
// Double-DQN target values — as ONE grid: workgroups [0, n_taken) run the first body, the rest the second.  The two
// read the same hidden states and neither reads the other's output; as two launches they sat on two hardware queues and
// the update's serial chain paid a cross-queue hand-over (~10 us) to meet again.
template <int A>
__global__ void __launch_bounds__(256) qheads_pair_kernel(const macjd_qtaken_io tio, const macjd_doubleq_io dio, const int n_taken) {
    __shared__ union PairLds {
        QtakenLds t;
        DoubleQLds<A> d;
        __device__ PairLds() {}
    } L;
    if ((int)blockIdx.x < n_taken) qhead_taken_body(tio, blockIdx.x, L.t);
    else qhead_double_q_body<A>(dio, (int)blockIdx.x - n_taken, L.d);
}

}  // namespace macjd

extern "C" int macjd_qhead_taken_supported(int32_t H, int32_t A) {
    return (H == macjd::EP_H && A >= 1 && A <= macjd::QT_AMAX) ? 1 : 0;
}

static int qtaken_check(const macjd_qtaken_io* io) {
    using namespace macjd;
    if (!io) return set_err(MACJD_EINVAL, "%s", "macjd_qhead_taken: NULL io");
    if (!macjd_qhead_taken_supported(io->H, io->A)) return set_err(MACJD_EUNSUPPORTED, "%s", "macjd_qhead_taken: unsupported H / A");
    if (io->n_rows < 0 || !io->h || !io->idx || !io->P || !io->W1 || !io->b1 || !io->w2 || !io->act || !io->q)
        return set_err(MACJD_EINVAL, "%s", "macjd_qhead_taken: bad n_rows or NULL pointer");
    if ((io->idx_elem_size != 4 && io->idx_elem_size != 8) || io->h_ld < io->H || (io->h_ld & 3) || (((uintptr_t)io->h) & 15) ||
        io->w1_ld < io->H + io->A + 1 || io->act_ld < io->H || (io->x && io->x_ld < io->H + io->A + 1))
        return set_err(MACJD_EINVAL, "%s", "macjd_qhead_taken: bad stride / alignment / index size");
    return MACJD_OK;
}

extern "C" int macjd_qhead_taken(const macjd_qtaken_io* io, void* hip_stream) {
    using namespace macjd;
    const int rc = qtaken_check(io);
    if (rc != MACJD_OK) return rc;
    if (io->n_rows == 0) return MACJD_OK;
    const dim3 grid((unsigned)((io->n_rows + 16 * QT_TILES - 1) / (16 * QT_TILES)));
    hipLaunchKernelGGL(qhead_taken_kernel, grid, dim3(256), 0, (hipStream_t)hip_stream, *io);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "macjd_qhead_taken: %s", hipGetErrorString(err));
    return MACJD_OK;
}

extern "C" int macjd_qhead_double_q_supported(int32_t H, int32_t A) {
    return (H == macjd::EP_H) && (A == 5 || A == 9 || A == 17 || A == 33);
}

static int doubleq_check(const macjd_doubleq_io* io) {
    using namespace macjd;
    if (!io) return set_err(MACJD_EINVAL, "%s", "macjd_qhead_double_q: NULL io");
    if (!macjd_qhead_double_q_supported(io->H, io->A))
        return set_err(MACJD_EUNSUPPORTED, "%s", "macjd_qhead_double_q: unsupported H / A");
    if (io->n_rows < 0 || !io->h_e || !io->h_t || !io->P_e || !io->P_t || !io->W1_e || !io->b1_e || !io->w2_e || !io->b2_e ||
        !io->W1_t || !io->b1_t || !io->w2_t || !io->b2_t || !io->out)
        return set_err(MACJD_EINVAL, "%s", "macjd_qhead_double_q: bad n_rows or NULL pointer");
    if (io->he_ld < io->H || io->ht_ld < io->H || io->pe_ld < io->A || io->pt_ld < io->A || io->w1e_ld < io->H + io->A + 1 ||
        io->w1t_ld < io->H + io->A + 1)
        return set_err(MACJD_EINVAL, "%s", "macjd_qhead_double_q: row stride smaller than the row");
    return MACJD_OK;
}

extern "C" int macjd_qheads_pair(const macjd_qtaken_io* taken, const macjd_doubleq_io* dq, void* hip_stream) {
    using namespace macjd;
    int rc = qtaken_check(taken);
    if (rc != MACJD_OK) return rc;
    rc = doubleq_check(dq);
    if (rc != MACJD_OK) return rc;
    if (taken->A != dq->A) return set_err(MACJD_EINVAL, "%s", "macjd_qheads_pair: the two heads differ in A");
    const int64_t nt = (taken->n_rows + 16 * QT_TILES - 1) / (16 * QT_TILES), nd = (dq->n_rows + 15) / 16;
    if (nt + nd == 0) return MACJD_OK;
    if (nt + nd > 0x7fffffff) return set_err(MACJD_EINVAL, "%s", "macjd_qheads_pair: too many rows for one grid");
    const dim3 grid((unsigned)(nt + nd)), block(256);
    hipStream_t s = (hipStream_t)hip_stream;
    if (dq->A == 5) hipLaunchKernelGGL((qheads_pair_kernel<5>), grid, block, 0, s, *taken, *dq, (int)nt);
    else if (dq->A == 9) hipLaunchKernelGGL((qheads_pair_kernel<9>), grid, block, 0, s, *taken, *dq, (int)nt);
    else if (dq->A == 17) hipLaunchKernelGGL((qheads_pair_kernel<17>), grid, block, 0, s, *taken, *dq, (int)nt);
    else hipLaunchKernelGGL((qheads_pair_kernel<33>), grid, block, 0, s, *taken, *dq, (int)nt);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "macjd_qheads_pair: %s", hipGetErrorString(err));
    return MACJD_OK;
}

extern "C" int macjd_qhead_double_q(const macjd_doubleq_io* io, void* hip_stream) {
    using namespace macjd;
    const int rc = doubleq_check(io);
    if (rc != MACJD_OK) return rc;
    if (io->n_rows == 0) return MACJD_OK;
    const dim3 grid((unsigned)((io->n_rows + 15) / 16)), block(256);
    hipStream_t s = (hipStream_t)hip_stream;
    if (io->A == 5) hipLaunchKernelGGL((qhead_double_q_kernel<5>), grid, block, 0, s, *io);
    else if (io->A == 9) hipLaunchKernelGGL((qhead_double_q_kernel<9>), grid, block, 0, s, *io);
    else if (io->A == 17) hipLaunchKernelGGL((qhead_double_q_kernel<17>), grid, block, 0, s, *io);
    else hipLaunchKernelGGL((qhead_double_q_kernel<33>), grid, block, 0, s, *io);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "macjd_qhead_double_q: %s", hipGetErrorString(err));
    return MACJD_OK;
}

extern "C" int macjd_agent_episode_supported(int32_t J, int32_t H, int32_t A) {
    // J in {2, 3, 6}: one row tile per agent; any other agent count: four tiles of consecutive rows (FLAT)
    return (H == macjd::EP_H) && J >= 1 && (A == 5 || A == 9 || A == 17 || A == 33);
}

extern "C" int macjd_agent_episode(const macjd_agent_episode_io* io, void* hip_stream) {
    using namespace macjd;
    if (!io) return set_err(MACJD_EINVAL, "%s", "macjd_agent_episode: NULL io");
    if (!macjd_agent_episode_supported(io->J, io->H, io->A))
        return set_err(MACJD_EUNSUPPORTED, "%s", "macjd_agent_episode: unsupported J / H / A (see include/macjd_nets.h)");
    if (io->n_envs < 0 || io->T < 1 || !io->gi || !io->P_all || !io->w_hh || !io->b_hh || !io->W1 || !io->b1 || !io->w2 ||
        !io->b2 || !io->hidden || !io->T_out || !io->P_out || (!io->greedy_only && !io->eps))
        return set_err(MACJD_EINVAL, "%s", "macjd_agent_episode: bad n_envs / T or NULL pointer");
    if ((io->gi_ld != 0 && io->gi_ld < 3 * io->H) || (io->p_ld != 0 && io->p_ld < io->A) || io->w1_ld < io->H + io->A + 1)
        return set_err(MACJD_EINVAL, "%s", "macjd_agent_episode: row stride smaller than the row");
    if (io->avail && io->avail_elem_size != 4 && io->avail_elem_size != 8)
        return set_err(MACJD_EINVAL, "%s", "macjd_agent_episode: avail_elem_size must be 4 or 8");
    if (((uintptr_t)io->w_hh) & 15) return set_err(MACJD_EINVAL, "%s", "macjd_agent_episode: w_hh must be 16-byte aligned");
    if (io->n_envs == 0) return MACJD_OK;
    const bool flat = !(io->J == 2 || io->J == 3 || io->J == 6) || io->A == 33;
    const int64_t wgs = flat ? (io->n_envs * io->J + 63) / 64 : (io->n_envs + 15) / 16;
    if (wgs > 0x7fffffff) return set_err(MACJD_EINVAL, "%s", "macjd_agent_episode: too many rows for one launch");
    const dim3 grid((unsigned)wgs), block(256);
    hipStream_t s = (hipStream_t)hip_stream;
#define MACJD_EP(J_, A_) hipLaunchKernelGGL((agent_episode_kernel<J_, A_>), grid, block, 0, s, *io)
#define MACJD_EP_FLAT(A_) hipLaunchKernelGGL((agent_episode_kernel<4, A_, true>), grid, block, 0, s, *io)
#define MACJD_EP_J(J_)                     \
    do {                                   \
        if (io->A == 5) MACJD_EP(J_, 5);   \
        else if (io->A == 9) MACJD_EP(J_, 9); \
        else MACJD_EP(J_, 17);             \
    } while (0)
    if (flat) {
        if (io->A == 5) MACJD_EP_FLAT(5);
        else if (io->A == 9) MACJD_EP_FLAT(9);
        else if (io->A == 17) MACJD_EP_FLAT(17);
        else MACJD_EP_FLAT(33);
    } else if (io->J == 2) MACJD_EP_J(2);
    else if (io->J == 3) MACJD_EP_J(3);
    else MACJD_EP_J(6);
#undef MACJD_EP_J
#undef MACJD_EP_FLAT
#undef MACJD_EP
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "macjd_agent_episode: %s", hipGetErrorString(err));
    return MACJD_OK;
}

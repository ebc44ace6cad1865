// macjd_wgrad.hip — split-K weight / bias gradient of a Linear layer on exact-f32 MFMA.  C-ABI: include/macjd_nets.h
//
//   dW[M,N] = gout[K,M]^T x inp[K,N]      db[M] = sum_k gout[k,:]
//
// In the learner's backward these products have tiny outputs (hyper-network and Q-head weights, 64..384 x 46..128)
// and a long reduction (K = B (T+1) = 3232 mixer rows, or B (T+1) J = 9696 agent rows).  A library GEMM tiles the
// OUTPUT, i.e. runs them on 6..12 workgroups for 25..50 us each; here the REDUCTION is tiled:
//   * grid = (N tiles, M tiles, K chunks), 64 x 64 output tile x 64-row chunk (WG_KC) per workgroup (4 waves as 2 x 2,
//     32 x 32 per wave = 2 x 2 accumulators of v_mfma_f32_16x16x4_f32);
//   * both operand chunks are staged into LDS with coalesced 16-byte loads, [WG_KC][80] floats each (row pitch
//     80 = 16 mod 32: the fragment reads `chunk[4 kk + (lane >> 4)][16 t + (lane & 15)]` of both operands hit 32
//     distinct banks per 32-lane group);
//   * the chunk's partial tile goes to workspace[chunk][M][N]; the bias partial (column sums of the staged gout
//     chunk) is produced by the workgroups of N-tile 0;
//   * wgrad_reduce_kernel sums the chunks in index order (deterministic, no float atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/macjd.h"
#include "../../include/macjd_nets.h"
#include "macjd_err.h"

namespace macjd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef MACJD_WG_KC
#define MACJD_WG_KC 64    // rows of the reduction per work item: 41 KB of LDS, three workgroups per CU.  (128 rows = 82 KB = ONE
                          // workgroup per CU, whose load / multiply / store phases then never overlap another's: update 117.1 us
                          // vs 111.5; 32 rows: 115.9 — twice the partial tiles again.  With the one-lane-per-output reduce of
                          // round 2 the extra partials of 64-row items cost what the partial launch gained.)
#endif
#ifndef MACJD_WG_ABLATE
#define MACJD_WG_ABLATE 0   // timing-only builds (scripts/probe_wgrad.py): 1 no MFMA loop, 2 no global loads, 4 no partial store
#endif
#ifndef MACJD_WG_SUB
#define MACJD_WG_SUB 1      // WG_KC-row pieces per work item (2: piece s + 1 is loaded into registers while piece s is multiplied —
                            // measured slower on the update's problems: 18.3 vs 11.7 us, half the workgroups with twice the latency each)
#endif
constexpr int WG_BM = 64, WG_BN = 64, WG_KC = MACJD_WG_KC, WG_PITCH = 80, WG_SUB = MACJD_WG_SUB;
constexpr int WG_ROWS = WG_KC * WG_SUB;   // rows of the reduction per work item (= per partial tile in the workspace)

__host__ __device__ inline int64_t wg_pad(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// fields of one problem as the partial-products body needs them (selected with compile-time indices from a batch)
struct WgradProb {
    int64_t K, gout_ld, inp_ld;
    int M, N, Mp, Np, n_chunks;
    const float* gout;
    const float* inp;
    float* workspace;
    bool want_db;
    const float* outer_vec;   // see macjd_wgrad_io
    const float* outer_w;
};

__device__ __forceinline__ void wgrad_partial_body(const WgradProb& io, const int tn, const int tm, const int chunk,
                                                   float* __restrict__ sA, float* __restrict__ sB) {
    const int Mp = io.Mp, Np = io.Np;
    const int m0 = tm * WG_BM, n0 = tn * WG_BN;
    const int64_t k0 = (int64_t)chunk * WG_ROWS;
    const int tid = threadIdx.x;
    // ---- a WG_KC-row piece of both operands: 64 columns, 16 threads per row, 16-byte loads where aligned.  The piece is
    // loaded into registers (8 x 2 float4 per thread) and written to LDS in a second step, so that the loads of piece
    // s + 1 are in flight while piece s is multiplied (measured on the update's seven problems: loads ~10 us and MFMA
    // ~8 us of the launch were additive when each work item was load -> barrier -> multiply) ----
    // All loads are branch-free: rows past K and columns past M / N are read from CLAMPED (valid) addresses and zeroed by
    // selects afterwards.  A load under a divergent `if` makes hipcc wait for ALL outstanding loads at the join — the
    // first version guarded every element and so serialised its 8 x 2 row pieces, one memory latency after the other.
    // Width of a load is uniform per operand: 16 bytes (rows 16-byte aligned, tile inside the matrix), 8 bytes (even row
    // length: a column pair is inside or outside as a whole) or 4.
    constexpr int NIT = (WG_KC * 16) / 256;
    float4 va[NIT], vb[NIT];
    auto load_operand = [&](const float* __restrict__ base, const int64_t ld, const int cols, const int c0, const int piece,
                            float4* __restrict__ out) {
        const bool al16 = ((ld & 3) == 0) && ((((uintptr_t)base) & 15) == 0) && (c0 + 64 <= cols);
        const bool al8 = ((ld & 1) == 0) && ((cols & 1) == 0) && ((((uintptr_t)base) & 7) == 0) && cols >= 2;
        const int c4 = (tid & 15) * 4;
        int64_t roff[NIT];
        bool rok[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int64_t k = k0 + (int64_t)piece * WG_KC + ((it * 256 + tid) >> 4);
            rok[it] = k < io.K && !(MACJD_WG_ABLATE & 2);
            roff[it] = (k < io.K ? k : io.K - 1) * ld;
        }
        if (al16) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) out[it] = *reinterpret_cast<const float4*>(base + roff[it] + c0 + c4);
        } else if (al8) {
            const int ca = c0 + c4 < cols ? c0 + c4 : cols - 2, cb = c0 + c4 + 2 < cols ? c0 + c4 + 2 : cols - 2;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const float2 lo = *reinterpret_cast<const float2*>(base + roff[it] + ca);
                const float2 hi = *reinterpret_cast<const float2*>(base + roff[it] + cb);
                out[it] = float4{lo.x, lo.y, hi.x, hi.y};
            }
        } else {
            int cc[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) cc[e] = c0 + c4 + e < cols ? c0 + c4 + e : cols - 1;
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                out[it] = float4{base[roff[it] + cc[0]], base[roff[it] + cc[1]], base[roff[it] + cc[2]], base[roff[it] + cc[3]]};
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            out[it].x = (rok[it] && c0 + c4 + 0 < cols) ? out[it].x : 0.0f;
            out[it].y = (rok[it] && c0 + c4 + 1 < cols) ? out[it].y : 0.0f;
            out[it].z = (rok[it] && c0 + c4 + 2 < cols) ? out[it].z : 0.0f;
            out[it].w = (rok[it] && c0 + c4 + 3 < cols) ? out[it].w : 0.0f;
        }
    };
    auto load_piece = [&](const int piece) {
        load_operand(io.gout, io.gout_ld, io.M, m0, piece, va);
        load_operand(io.inp, io.inp_ld, io.N, n0, piece, vb);
        if (io.outer_vec) {   // (uniform) operand formed on the fly: (relu output > 0) ? vec[k] * w[m] : 0 — the very
            const int c4 = (tid & 15) * 4;   // expression of splitrelu_backward_kernel, so the products are bit-identical
            float w4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w4[e] = io.outer_w[m0 + c4 + e < io.M ? m0 + c4 + e : io.M - 1];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int64_t k = k0 + (int64_t)piece * WG_KC + ((it * 256 + tid) >> 4);
                const float gv = io.outer_vec[k < io.K ? k : io.K - 1];
                va[it].x = (va[it].x > 0.0f) ? gv * w4[0] : 0.0f;
                va[it].y = (va[it].y > 0.0f) ? gv * w4[1] : 0.0f;
                va[it].z = (va[it].z > 0.0f) ? gv * w4[2] : 0.0f;
                va[it].w = (va[it].w > 0.0f) ? gv * w4[3] : 0.0f;
            }
        }
    };
    auto store_piece = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * 256 + tid;
            const int r = idx >> 4, c4 = (idx & 15) * 4;
            *reinterpret_cast<float4*>(&sA[r * WG_PITCH + c4]) = va[it];
            *reinterpret_cast<float4*>(&sB[r * WG_PITCH + c4]) = vb[it];
        }
    };
    // ---- 32 x 32 per wave: acc[sm][sn] += A^T fragment x B fragment over the work item's rows ----
    const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* pa = sA + lk * WG_PITCH + wm + li;
    const float* pb = sB + lk * WG_PITCH + wn + li;
    const bool bias_lane = io.want_db && tn == 0 && tid < WG_BM;
    float bias_sum = 0.0f;
    const int n_pieces = (int)((((io.K - k0) < (int64_t)WG_ROWS ? (io.K - k0) : (int64_t)WG_ROWS) + WG_KC - 1) / WG_KC);
    load_piece(0);
    store_piece();
    __syncthreads();
    for (int piece = 0; piece < n_pieces; ++piece) {
        const bool more = piece + 1 < n_pieces;
        if (more) load_piece(piece + 1);       // in flight during the products below
        // Two fragment sets, ping-ponged and PINNED (as in macjd_mlp.hip): the four ds_reads of k-step kk + 1 are issued
        // right behind the first MFMA of k-step kk and consumed one step later, so their LDS latency hides behind three
        // MFMAs of 32 cycles.  Left alone the scheduler sinks every read next to its use: read -> lgkmcnt(0) -> 4 MFMA per
        // k-step, which ran the loop at a third of the MFMA rate (PMC, round 3, profiles/r03_wgrad_pmc.txt:
        // SQ_VALU_MFMA_BUSY_CYCLES = 4 096 per wave = the ideal 128 x 32, but 12 900 cycles of issue stall per wave).
        constexpr int KSTEPS = (MACJD_WG_ABLATE & 1) ? 2 : WG_KC / 4;
        float fa0 = pa[0], fa1 = pa[16], fb0 = pb[0], fb1 = pb[16];
        float ga0, ga1, gb0, gb1;
#pragma unroll 4
        for (int kk = 0; kk < KSTEPS; kk += 2) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0, fb0, acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            ga0 = pa[(kk + 1) * 4 * WG_PITCH]; ga1 = pa[(kk + 1) * 4 * WG_PITCH + 16];
            gb0 = pb[(kk + 1) * 4 * WG_PITCH]; gb1 = pb[(kk + 1) * 4 * WG_PITCH + 16];
            __builtin_amdgcn_sched_barrier(0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0, fb1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1, fb0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1, fb1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga0, gb0, acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int kn = (kk + 2 < KSTEPS) ? kk + 2 : KSTEPS - 1;   // past the end: re-read (unused)
            fa0 = pa[kn * 4 * WG_PITCH]; fa1 = pa[kn * 4 * WG_PITCH + 16];
            fb0 = pb[kn * 4 * WG_PITCH]; fb1 = pb[kn * 4 * WG_PITCH + 16];
            __builtin_amdgcn_sched_barrier(0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga0, gb1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga1, gb0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga1, gb1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // bias partial: column sums of the staged gout piece (N-tile 0 only), pieces added in order
        if (bias_lane) {
            float sp = 0.0f;
#pragma unroll 8
            for (int r = 0; r < WG_KC; ++r) sp += sA[r * WG_PITCH + tid];
            bias_sum += sp;
        }
        if (more) {
            __syncthreads();                   // every wave is done reading this piece
            store_piece();
            __syncthreads();
        }
    }
    float* ws = io.workspace + (int64_t)chunk * Mp * Np;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + a * 16 + lk * 4 + r;   // C layout: row = (lane >> 4) * 4 + reg
                const int n = n0 + wn + b * 16 + li;           //           col = lane & 15
                if (!(MACJD_WG_ABLATE & 4) || acc[a][b][r] == 12345.678f) ws[(int64_t)m * Np + n] = acc[a][b][r];
            }
    if (bias_lane) io.workspace[(int64_t)io.n_chunks * Mp * Np + (int64_t)chunk * Mp + m0 + tid] = bias_sum;
}

__global__ void __launch_bounds__(256) wgrad_partial_kernel(const macjd_wgrad_io io, const int Mp, const int Np) {
    __shared__ float sA[WG_KC * WG_PITCH];   // gout chunk  [k][m]
    __shared__ float sB[WG_KC * WG_PITCH];   // inp chunk   [k][n]
    WgradProb p;
    p.K = io.K; p.gout_ld = io.gout_ld; p.inp_ld = io.inp_ld; p.M = io.M; p.N = io.N; p.Mp = Mp; p.Np = Np;
    p.n_chunks = (int)gridDim.z; p.gout = io.gout; p.inp = io.inp; p.workspace = io.workspace; p.want_db = io.db != nullptr;
    p.outer_vec = io.outer_vec; p.outer_w = io.outer_w;
    wgrad_partial_body(p, blockIdx.x, blockIdx.y, blockIdx.z, sA, sB);
}

// Sum of the K-chunk partials, fixed order (deterministic).  A workgroup owns 64 consecutive output elements; consecutive
// lanes = consecutive n (every load instruction of a wave reads one contiguous row piece of a partial tile); the chunks
// are taken in rounds of eight and round r belongs to wave r mod 4, so an output's <= 76 loads are in flight from four
// waves at once instead of queueing behind one lane (one lane per output: 12.6 us for the update's problems, a
// latency chain of ten rounds).  Order of the sum: wave w adds chunk 8 r + s (r = w, w + 4, ...) into its accumulator s,
// combines ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)); the four waves' sums meet as (p0 + p1) + (p2 + p3).
__device__ __forceinline__ float wgrad_sum_rounds(const float* __restrict__ p, const int64_t stride, const int n_chunks, const int wave) {
    float a[8];
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) a[s8] = 0.0f;
    for (int c = 8 * wave; c < n_chunks; c += 32) {
        float v[8];
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {   // (no branch around a load: past the last chunk the last chunk is read and dropped)
            const int cc = c + s8 < n_chunks ? c + s8 : n_chunks - 1;
            v[s8] = p[(int64_t)cc * stride];
        }
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) a[s8] += (c + s8 < n_chunks) ? v[s8] : 0.0f;
    }
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

constexpr int WG_RED = 64;   // outputs per reduce workgroup

// the four waves' sums of output `lane` -> the value (valid in wave 0)
__device__ __forceinline__ float wgrad_meet(float (*part)[WG_RED], const float mine, const int wave, const int lane) {
    part[wave][lane] = mine;
    __syncthreads();
    const float out = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    __syncthreads();   // (the caller's loop writes `part` again)
    return out;
}

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const macjd_wgrad_io io, const int Mp, const int Np,
                                                           const int n_chunks) {
    __shared__ float part[4][WG_RED];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t total = (int64_t)io.M * io.N, all = total + (io.db ? io.M : 0);
    for (int64_t i0 = (int64_t)blockIdx.x * WG_RED; i0 < all; i0 += (int64_t)gridDim.x * WG_RED) {
        const int64_t i = i0 + lane < all ? i0 + lane : all - 1;   // (lanes past the end repeat the last output, never store)
        const float* src;
        int64_t stride;
        float* dst;
        if (i < total) {
            const int m = (int)(i / io.N), n = (int)(i - (int64_t)m * io.N);
            src = io.workspace + (int64_t)m * Np + n; stride = (int64_t)Mp * Np; dst = io.dW + (int64_t)m * io.dw_ld + n;
        } else {
            const int m = (int)(i - total);
            src = io.workspace + (int64_t)n_chunks * Mp * Np + m; stride = Mp; dst = io.db + m;
        }
        const float v = wgrad_meet(part, wgrad_sum_rounds(src, stride, n_chunks, wave), wave, lane);
        if (wave == 0 && i0 + lane < all) *dst = v;
    }
}

// ---- several problems in one launch pair (the backward pass of the learner defers its weight gradients) ----
struct WgradBatch {
    macjd_wgrad_io io[MACJD_WGRAD_MAX_BATCH];
    int Mp[MACJD_WGRAD_MAX_BATCH], Np[MACJD_WGRAD_MAX_BATCH], chunks[MACJD_WGRAD_MAX_BATCH];
    int tiles_n[MACJD_WGRAD_MAX_BATCH], tiles_m[MACJD_WGRAD_MAX_BATCH];
    int64_t wg_start[MACJD_WGRAD_MAX_BATCH + 1];    // prefix of workgroups (tile x chunk work items) per problem
    int64_t out_start[MACJD_WGRAD_MAX_BATCH + 1];   // prefix of reduce work items (M N + M outputs) per problem
    int n;
};

// workgroup w works on problem p with wg_start[p] <= w < wg_start[p + 1]; the problem's fields are picked with
// compile-time indices only (a run-time index into the by-value table would copy the ~1 KB struct to scratch)
__global__ void __launch_bounds__(256) wgrad_partial_many_kernel(const WgradBatch b) {
    __shared__ float sA[WG_KC * WG_PITCH];
    __shared__ float sB[WG_KC * WG_PITCH];
    const int64_t w = blockIdx.x;
    WgradProb p;
    int tiles_n = b.tiles_n[0], tiles_m = b.tiles_m[0];
    int64_t start = 0;
#define MACJD_PICK(q)                                                                                               \
    p.K = b.io[q].K; p.gout_ld = b.io[q].gout_ld; p.inp_ld = b.io[q].inp_ld; p.M = b.io[q].M; p.N = b.io[q].N;       \
    p.Mp = b.Mp[q]; p.Np = b.Np[q]; p.n_chunks = b.chunks[q]; p.gout = b.io[q].gout; p.inp = b.io[q].inp;            \
    p.workspace = b.io[q].workspace; p.want_db = b.io[q].db != nullptr; tiles_n = b.tiles_n[q]; tiles_m = b.tiles_m[q]; \
    p.outer_vec = b.io[q].outer_vec; p.outer_w = b.io[q].outer_w;                                                    \
    start = b.wg_start[q];
    MACJD_PICK(0)
#pragma unroll
    for (int q = 1; q < MACJD_WGRAD_MAX_BATCH; ++q) {
        if (q < b.n && w >= b.wg_start[q]) { MACJD_PICK(q) }
    }
#undef MACJD_PICK
    const int local = (int)(w - start);
    const int tn = local % tiles_n, rest = local / tiles_n;
    const int tm = rest % tiles_m, chunk = rest / tiles_m;
    wgrad_partial_body(p, tn, tm, chunk, sA, sB);
}

__global__ void __launch_bounds__(256) wgrad_reduce_many_kernel(const WgradBatch b) {
    __shared__ float part[4][WG_RED];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t total_all = b.out_start[b.n];
    for (int64_t g0 = (int64_t)blockIdx.x * WG_RED; g0 < total_all; g0 += (int64_t)gridDim.x * WG_RED) {
        const int64_t g = g0 + lane < total_all ? g0 + lane : total_all - 1;
        int M = b.io[0].M, N = b.io[0].N, Mp = b.Mp[0], Np = b.Np[0], n_chunks = b.chunks[0];
        int64_t start = 0, dw_ld = b.io[0].dw_ld;
        const float* workspace = b.io[0].workspace;
        float* dW = b.io[0].dW;
        float* db = b.io[0].db;
#pragma unroll
        for (int q = 1; q < MACJD_WGRAD_MAX_BATCH; ++q) {
            if (q < b.n && g >= b.out_start[q]) {
                M = b.io[q].M; N = b.io[q].N; Mp = b.Mp[q]; Np = b.Np[q]; n_chunks = b.chunks[q];
                start = b.out_start[q]; dw_ld = b.io[q].dw_ld; workspace = b.io[q].workspace; dW = b.io[q].dW; db = b.io[q].db;
            }
        }
        const int64_t i = g - start;
        const int64_t total = (int64_t)M * N;
        // (a problem without a bias output still owns M items behind its M N: they are summed and dropped)
        const bool is_w = i < total;
        const int m = is_w ? (int)(i / N) : (int)(i - total);
        const int n = is_w ? (int)(i - (int64_t)m * N) : 0;
        const float* src = is_w ? workspace + (int64_t)m * Np + n : workspace + (int64_t)n_chunks * Mp * Np + m;
        const float v = wgrad_meet(part, wgrad_sum_rounds(src, is_w ? (int64_t)Mp * Np : (int64_t)Mp, n_chunks, wave), wave, lane);
        if (wave == 0 && g0 + lane < total_all) {
            if (is_w) dW[(int64_t)m * dw_ld + n] = v;
            else if (db) db[m] = v;
        }
    }
}

}  // namespace macjd

extern "C" int64_t macjd_linear_wgrad_workspace_floats(int64_t K, int32_t M, int32_t N) {
    using namespace macjd;
    if (K < 1 || M < 1 || N < 1) return -1;
    const int64_t chunks = (K + WG_ROWS - 1) / WG_ROWS, Mp = wg_pad(M, WG_BM), Np = wg_pad(N, WG_BN);
    return chunks * Mp * Np + chunks * Mp;
}

extern "C" int macjd_linear_wgrad(const macjd_wgrad_io* io, void* hip_stream) {
    using namespace macjd;
    if (!io || io->K < 1 || io->M < 1 || io->N < 1 || !io->gout || !io->inp || !io->dW || !io->workspace)
        return set_err(MACJD_EINVAL, "%s", "macjd_linear_wgrad: bad argument");
    if (io->gout_ld < io->M || io->inp_ld < io->N || io->dw_ld < io->N)
        return set_err(MACJD_EINVAL, "%s", "macjd_linear_wgrad: bad leading dimension");
    if ((io->outer_vec == nullptr) != (io->outer_w == nullptr))
        return set_err(MACJD_EINVAL, "%s", "macjd_linear_wgrad: outer_vec and outer_w go together");
    const int Mp = (int)wg_pad(io->M, WG_BM), Np = (int)wg_pad(io->N, WG_BN);
    const int64_t chunks = (io->K + WG_ROWS - 1) / WG_ROWS;
    if (chunks > 65535) return set_err(MACJD_EUNSUPPORTED, "%s", "macjd_linear_wgrad: K too large");
    hipStream_t s = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(wgrad_partial_kernel, dim3(Np / WG_BN, Mp / WG_BM, (unsigned)chunks), dim3(256), 0, s, *io, Mp, Np);
    const int64_t total = (int64_t)io->M * io->N + io->M;   // WG_RED output elements per workgroup
    const unsigned rblocks = (unsigned)((total + WG_RED - 1) / WG_RED < 8192 ? (total + WG_RED - 1) / WG_RED : 8192);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rblocks), dim3(256), 0, s, *io, Mp, Np, (int)chunks);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "macjd_linear_wgrad: %s", hipGetErrorString(err));
    return MACJD_OK;
}

extern "C" int macjd_linear_wgrad_many(const macjd_wgrad_io* ios, int32_t n, void* hip_stream) {
    using namespace macjd;
    if (!ios || n < 1 || n > MACJD_WGRAD_MAX_BATCH)
        return set_err(MACJD_EINVAL, "%s", "macjd_linear_wgrad_many: 1..MACJD_WGRAD_MAX_BATCH problems");
    WgradBatch b{};
    b.n = n;
    for (int p = 0; p < n; ++p) {
        const macjd_wgrad_io* io = &ios[p];
        if (io->K < 1 || io->M < 1 || io->N < 1 || !io->gout || !io->inp || !io->dW || !io->workspace)
            return set_err(MACJD_EINVAL, "%s", "macjd_linear_wgrad_many: bad argument");
        if (io->gout_ld < io->M || io->inp_ld < io->N || io->dw_ld < io->N)
            return set_err(MACJD_EINVAL, "%s", "macjd_linear_wgrad_many: bad leading dimension");
        if ((io->outer_vec == nullptr) != (io->outer_w == nullptr))
            return set_err(MACJD_EINVAL, "%s", "macjd_linear_wgrad_many: outer_vec and outer_w go together");
        b.io[p] = *io;
        b.Mp[p] = (int)wg_pad(io->M, WG_BM);
        b.Np[p] = (int)wg_pad(io->N, WG_BN);
        b.chunks[p] = (int)((io->K + WG_ROWS - 1) / WG_ROWS);
        b.tiles_n[p] = b.Np[p] / WG_BN;
        b.tiles_m[p] = b.Mp[p] / WG_BM;
        b.wg_start[p + 1] = b.wg_start[p] + (int64_t)b.tiles_n[p] * b.tiles_m[p] * b.chunks[p];
        b.out_start[p + 1] = b.out_start[p] + (int64_t)io->M * io->N + io->M;
    }
    for (int p = n; p < MACJD_WGRAD_MAX_BATCH; ++p) {
        b.wg_start[p + 1] = b.wg_start[n];
        b.out_start[p + 1] = b.out_start[n];
        b.tiles_n[p] = b.tiles_m[p] = 1;
    }
    if (b.wg_start[n] > 0x7fffffff) return set_err(MACJD_EUNSUPPORTED, "%s", "macjd_linear_wgrad_many: too many work items");
    hipStream_t s = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(wgrad_partial_many_kernel, dim3((unsigned)b.wg_start[n]), dim3(256), 0, s, b);
    const int64_t total = b.out_start[n];
    const unsigned rblocks = (unsigned)((total + WG_RED - 1) / WG_RED < 8192 ? (total + WG_RED - 1) / WG_RED : 8192);
    hipLaunchKernelGGL(wgrad_reduce_many_kernel, dim3(rblocks), dim3(256), 0, s, b);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "macjd_linear_wgrad_many: %s", hipGetErrorString(err));
    return MACJD_OK;
}

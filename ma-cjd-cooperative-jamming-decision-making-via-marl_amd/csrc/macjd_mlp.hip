// macjd_mlp.hip — fused dense-layer chain on the MI355X matrix cores, exact float32.
//
// y = act_n(W_n ... act_1(W_1 x + b_1) ... + b_n) for up to three layers in ONE launch, using
// v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate: an exact-f32 fmaf chain, so the path's 1e-5 tolerance on
// Q-values / hidden states holds; bf16 MFMA would not).  C-ABI: include/macjd_nets.h.
//
// A workgroup is 4 waves = 64 rows; each wave owns 16 rows for the whole chain; workgroups are persistent over
// 64-row tiles.
//  * Weights are used in torch's own [out, in] row-major layout — no re-packing pass, no workspace.  Row o of W_l
//    goes global -> LDS by LDS-DMA (global_load_lds, no VGPR round trip) to LDS row o of pitch ldw; each wave keeps
//    its 16 x width activation tile in a private LDS strip [16][lda]; the input rows arrive by LDS-DMA as well (one
//    instruction per row; everything a tile needs is in flight under ONE memory latency).
//  * Operand reads are ds_read_b128.  An MFMA sums over its four k lane-groups, and WHICH k a lane-group supplies is
//    free as long as A and B agree, so a "quad" of four consecutive MFMAs covers k in [16 Q, 16 Q + 16) with lane
//    group g = lane >> 4 supplying k = 16 Q + 4 g + j in MFMA j: lane l reads ONE float4 of A
//    (act[l & 15][16 Q + 4 g ..]) and one float4 of B per output tile (W[16 t + (l & 15)][16 Q + 4 g ..]) and feeds
//    four MFMAs from each — 4x fewer LDS instructions than ds_read_b32 fragments, which at one wave per SIMD are
//    issue-bound well below the MFMA rate.  Both pitches are = 8 (mod 16) floats: rows stay 16-B aligned and, for
//    the b128 lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32), the sixteen 16-B slots of a group are
//    distinct mod 256 B — conflict-free (brute-forced over pitches; = 2 (mod 4) slots is the condition).
//    Pad rows (out -> multiple of 16) and pad columns (in -> multiple of 16) are zero-filled with plain LDS stores.
//    Per output element the sum runs over k in the order (Q, j, g) instead of 0..K-1: still an exact-f32 fmaf
//    chain, a different association than a k-ordered loop (differences ~1e-7 relative, far inside the 1e-5 bar).
//  * A layer reads its whole input (k-outer loop, all output tiles accumulate in registers) before it overwrites the
//    strip with its output: no double buffer, no barrier inside the chain.  The k-loop is software-pipelined over two
//    register fragment sets with the issue order pinned (see mlp_layer).
//  * The C tile (col = lane & 15, row = (lane >> 4) * 4 + reg) gets bias + activation in registers and goes to the
//    strip (inner layers) or straight to global memory (last layer).
// When all layers' images fit in LDS together they are staged once per workgroup; otherwise (wide scenarios) each
// layer is staged just before use.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/macjd.h"
#include "../../include/macjd_nets.h"
#include "macjd_err.h"

namespace macjd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MLP_MAX_HIDDEN = 128;
constexpr int MLP_MAX_IN = 256;
constexpr int MLP_MAX_OUT = 384;
constexpr int MLP_WAVES = 4;
constexpr int MLP_LDS_BYTES = 160 * 1024;
// Phase ablation for scripts/ablate_mlp.cpp (timing only, results are wrong when non-zero): 1 = no weight DMA,
// 2 = no input staging, 4 = no MFMA loops, 16 = no epilogue stores.  Always 0 in the product build.
#ifndef MACJD_MLP_ABLATE
#define MACJD_MLP_ABLATE 0
#endif
constexpr int MLP_ABL = MACJD_MLP_ABLATE;

// smallest pitch >= width with pitch = 8 (mod 16) floats (see the header: conflict-free ds_read_b128 fragments)
__host__ __device__ inline int mlp_pitch(int width) { return ((width + 7) / 16) * 16 + 8; }
__host__ __device__ inline int mlp_k16(int k) { return (k + 15) & ~15; }
__host__ __device__ inline int mlp_n16(int n) { return (n + 15) & ~15; }
// LDS image of one layer: [N16 rows x ldw] weights, then N16 bias values
__host__ __device__ inline int mlp_image_floats(int k_in, int n_out) {
    return mlp_n16(n_out) * mlp_pitch(mlp_k16(k_in)) + mlp_n16(n_out);
}

struct MlpGeom {
    int ldw[3];     // LDS row pitch of layer l's weight image (floats)
    int off[3];     // float offset of layer l's image from the image region (all 0 when not resident)
    int vec16[4];   // [l]: W_l rows can move 16 B per lane; [3]: the input rows can
    int lda;        // strip pitch
    int resident;   // all images staged once per workgroup
};

template <int ACT>
__device__ __forceinline__ float mlp_act(float v) {
    if (ACT == MACJD_ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == MACJD_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

// Rows [first, n) step `step` of a row-major global matrix -> LDS rows of pitch lds_pitch, by LDS-DMA: lane l's SZ
// bytes land at the (wave-uniform) LDS row + l * SZ, so a row is a whole number of instructions and never spills
// into the next (padded) row.  The lane mask is hoisted out of the row loop: at one wave per SIMD every instruction
// around the DMA issue costs ~4 cycles, so a row must be a handful of instructions (address bumps + m0 + glds).
template <int SZ>
__device__ __forceinline__ void mlp_glds_rows(float* __restrict__ lds0, int lds_pitch, const float* __restrict__ g0,
                                              int64_t g_pitch, int first, int n, int step, int K, int lane) {
    constexpr int F = SZ / 4;             // floats per lane
    const int units = K / F;              // SZ-byte units per row (K % F == 0 by the caller's choice of SZ)
    for (int c = 0; c < units; c += 64) {  // one pass unless the row is wider than 64 units
        if (c + lane < units) {
            const float* g = g0 + (int64_t)first * g_pitch + (c + lane) * F;
            float* l = lds0 + first * lds_pitch + c * F;
            for (int r = first; r < n; r += step) {
                if constexpr (SZ == 16)   // the builtin wants a literal size
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
                else
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)l, 4, 0, 0);
                g += (int64_t)step * g_pitch;
                l += step * lds_pitch;
            }
        }
    }
}
__device__ __forceinline__ void mlp_glds_rows(float* lds0, int lds_pitch, const float* g0, int64_t g_pitch, int first,
                                              int n, int step, int K, int vec16, int lane) {
    if (vec16) mlp_glds_rows<16>(lds0, lds_pitch, g0, g_pitch, first, n, step, K, lane);
    else mlp_glds_rows<4>(lds0, lds_pitch, g0, g_pitch, first, n, step, K, lane);
}
__device__ __forceinline__ void mlp_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Whole workgroup: issue the DMA of layer (W [N,K], b [N]) into its LDS image and zero the pads.  The caller waits
// (mlp_dma_wait) and barriers before the first read.
__device__ __forceinline__ void mlp_stage_layer(float* __restrict__ img, int ldw, const float* __restrict__ W,
                                                const float* __restrict__ b, int K, int N, int vec16) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, tid = threadIdx.x;
    const int K4 = mlp_k16(K), N16 = mlp_n16(N);   // K4: the padded row width
    float* bias = img + N16 * ldw;
    if (!(MLP_ABL & 1)) {
        mlp_glds_rows(img, ldw, W, K, wave, N, MLP_WAVES, K, vec16, lane);
        if (wave == 0) mlp_glds_rows<4>(bias, 0, b, 0, 0, 1, 1, N, lane);
    }
    const int kp = K4 - K;   // pad columns of the real rows
    for (int idx = tid; idx < N * kp; idx += 64 * MLP_WAVES) {
        const int o = idx / kp;
        img[o * ldw + K + (idx - o * kp)] = 0.0f;
    }
    for (int idx = tid; idx < (N16 - N) * K4; idx += 64 * MLP_WAVES) {   // pad rows
        const int r = idx / K4;
        img[(N + r) * ldw + (idx - r * K4)] = 0.0f;
    }
    for (int idx = N + tid; idx < N16; idx += 64 * MLP_WAVES) bias[idx] = 0.0f;
}

// Operand fragments of G consecutive quads (one "block" = 4 G k-steps): G float4 of A + G x NT float4 of B.
template <int NT, int G>
struct MlpFrags {
    f32x4 a[G], b[G][NT];
    __device__ __forceinline__ void load(const float* __restrict__ a_ptr, const float* __restrict__ b_ptr, int tile_stride,
                                         int block) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            a[i] = *(const f32x4*)(a_ptr + (block * G + i) * 16);
#pragma unroll
            for (int t = 0; t < NT; ++t) b[i][t] = *(const f32x4*)(b_ptr + t * tile_stride + (block * G + i) * 16);
        }
    }
    __device__ __forceinline__ void mfma_first(f32x4* acc) const {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][0], b[0][0][0], acc[0], 0, 0, 0);
    }
    __device__ __forceinline__ void mfma_rest(f32x4* acc) const {
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (i + jj + t > 0) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][jj], b[i][t][jj], acc[t], 0, 0, 0);
    }
};

template <int NT, int ACT, bool LAST>
__device__ __forceinline__ void mlp_epilogue(const f32x4* acc, float* __restrict__ strip, int lda,
                                             const float* __restrict__ bias_lds, int N, float* __restrict__ y,
                                             int64_t y_ld, int64_t row0, int64_t n_rows) {
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int o = t * 16 + li;
        const float bv = bias_lds[o];   // zero in the padded columns
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = lk * 4 + r;
            const float v = mlp_act<ACT>(acc[t][r] + bv);
            if (MLP_ABL & 16) {
                if (v == 12345.678f) y[0] = v;   // keeps the values live
            } else if (LAST) {
                if (o < N && row0 + row < n_rows) y[(row0 + row) * y_ld + o] = v;
            } else {
                // padded columns hold act(0): they meet zero weight columns in the next layer and are never stored
                strip[row * lda + o] = v;
            }
        }
    }
}

// one dense layer for this wave's 16 rows; NT = number of 16-column output tiles (compile time)
template <int NT>
__device__ __forceinline__ void mlp_layer(float* __restrict__ strip, int lda, const float* __restrict__ img, int ldw,
                                          int K, int N, int act, bool last, float* __restrict__ y, int64_t y_ld,
                                          int64_t row0, int64_t n_rows) {
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int quads = (K + 15) / 16;
    const float* a_ptr = strip + li * lda + 4 * lk;
    const float* b_ptr = img + li * ldw + 4 * lk;
    const int ts = 16 * ldw;
    // Software pipeline over blocks of G quads (>= 8 MFMAs = >= 256 matrix-core cycles per block), two fragment sets
    // ping-ponged: inside block j the loads of block j+1 are issued right after the block's FIRST MFMA and consumed
    // by the next block, so (a) the wait the compiler puts before that first MFMA only drains loads issued a whole
    // block earlier and (b) the LDS latency of the new loads hides behind the remaining MFMAs.  The scheduling
    // barriers pin that order: left alone, the scheduler sinks every ds_read next to its use.
    constexpr int G = (NT >= 2) ? 1 : 2;
    const int blocks = quads / G;
    MlpFrags<NT, G> f0, f1;
    int j = 0;
    if (blocks > 0 && !(MLP_ABL & 4)) {
        f0.load(a_ptr, b_ptr, ts, 0);
        for (; j + 2 <= blocks; j += 2) {
            f0.mfma_first(acc);
            __builtin_amdgcn_sched_barrier(0);
            f1.load(a_ptr, b_ptr, ts, j + 1);
            __builtin_amdgcn_sched_barrier(0);
            f0.mfma_rest(acc);
            __builtin_amdgcn_sched_barrier(0);
            f1.mfma_first(acc);
            __builtin_amdgcn_sched_barrier(0);
            f0.load(a_ptr, b_ptr, ts, (j + 2 < blocks) ? j + 2 : blocks - 1);   // past the end: re-read (unused)
            __builtin_amdgcn_sched_barrier(0);
            f1.mfma_rest(acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (j < blocks) {   // odd number of blocks: set 0 holds the last one
            f0.mfma_first(acc);
            f0.mfma_rest(acc);
        }
    }
    for (int q = blocks * G; q < quads && !(MLP_ABL & 4); ++q) {   // a quad that does not fill a block (G = 2 only)
        MlpFrags<NT, 1> f;
        f.load(a_ptr, b_ptr, ts, q);
        f.mfma_first(acc);
        f.mfma_rest(acc);
    }
    const float* bias = img + NT * 16 * ldw;
    // activation / last are wave-uniform run-time values; the epilogue body is compiled per combination
    if (last) {
        if (act == MACJD_ACT_SIGMOID) mlp_epilogue<NT, MACJD_ACT_SIGMOID, true>(acc, strip, lda, bias, N, y, y_ld, row0, n_rows);
        else if (act == MACJD_ACT_RELU) mlp_epilogue<NT, MACJD_ACT_RELU, true>(acc, strip, lda, bias, N, y, y_ld, row0, n_rows);
        else mlp_epilogue<NT, MACJD_ACT_NONE, true>(acc, strip, lda, bias, N, y, y_ld, row0, n_rows);
    } else {
        if (act == MACJD_ACT_RELU) mlp_epilogue<NT, MACJD_ACT_RELU, false>(acc, strip, lda, bias, N, y, y_ld, row0, n_rows);
        else if (act == MACJD_ACT_SIGMOID) mlp_epilogue<NT, MACJD_ACT_SIGMOID, false>(acc, strip, lda, bias, N, y, y_ld, row0, n_rows);
        else mlp_epilogue<NT, MACJD_ACT_NONE, false>(acc, strip, lda, bias, N, y, y_ld, row0, n_rows);
    }
}

__device__ __forceinline__ void mlp_layer_dispatch(int nt, float* strip, int lda, const float* img, int ldw, int K, int N,
                                                   int act, bool last, float* y, int64_t y_ld, int64_t row0,
                                                   int64_t n_rows) {
    switch (nt) {  // wave-uniform
        case 1: mlp_layer<1>(strip, lda, img, ldw, K, N, act, last, y, y_ld, row0, n_rows); break;
        case 2: mlp_layer<2>(strip, lda, img, ldw, K, N, act, last, y, y_ld, row0, n_rows); break;
        case 3: mlp_layer<3>(strip, lda, img, ldw, K, N, act, last, y, y_ld, row0, n_rows); break;
        case 4: mlp_layer<4>(strip, lda, img, ldw, K, N, act, last, y, y_ld, row0, n_rows); break;
        case 8: mlp_layer<8>(strip, lda, img, ldw, K, N, act, last, y, y_ld, row0, n_rows); break;
        case 12: mlp_layer<12>(strip, lda, img, ldw, K, N, act, last, y, y_ld, row0, n_rows); break;
        case 24: mlp_layer<24>(strip, lda, img, ldw, K, N, act, last, y, y_ld, row0, n_rows); break;
        default: break;  // host rejects other tile counts
    }
}

// the whole chain for the 64-row tiles block, block + n_blocks, ... of one problem (one copy of the code: the
// single-problem kernel and the two-problem kernel both call it)
__device__ __forceinline__ void mlp_forward_body(const macjd_mlp_io& io, const MlpGeom& g, const int block, const int n_blocks) {
    extern __shared__ float lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave: scalar
    const int lda = g.lda;
    float* wbase = lds + MLP_WAVES * 16 * lda;          // weight-image region
    float* strip = lds + wave * 16 * lda;               // this wave's activation strip [16][lda]
    const int L = io.n_layers;
    if (g.resident) {                                   // all layers' images: issued now, waited for after the
#pragma unroll                                          // first tile's inputs have been issued as well
        for (int l = 0; l < 3; ++l)
            if (l < L) mlp_stage_layer(wbase + g.off[l], g.ldw[l], io.W[l], io.b[l], io.dims[l], io.dims[l + 1], g.vec16[l]);
    }
    const int K0 = io.dims[0], K0p = mlp_k16(K0);
    const int64_t n_tiles = (io.n_rows + 63) / 64;
    for (int64_t tile = block; tile < n_tiles; tile += n_blocks) {
        const int64_t row0 = tile * 64 + wave * 16;
        // this wave's 16 input rows: LDS-DMA straight into the strip; missing rows and pad columns are zeroed
        if (!(MLP_ABL & 2)) {
            const int64_t left = io.n_rows - row0;
            const int n_valid = left >= 16 ? 16 : (left > 0 ? (int)left : 0);
            mlp_glds_rows(strip, lda, io.x + row0 * io.x_ld, io.x_ld, 0, n_valid, 1, K0, g.vec16[3], lane);
            for (int r = n_valid; r < 16; ++r)
                for (int k = lane; k < K0; k += 64) strip[r * lda + k] = 0.0f;
            const int kp = K0p - K0;
            for (int idx = lane; idx < 16 * kp; idx += 64) {
                const int r = idx / kp;
                strip[r * lda + K0 + (idx - r * kp)] = 0.0f;
            }
        }
        // LDS-DMA data is ordered for a ds_read by the issuing wave's vmcnt wait followed by a barrier the reader
        // has passed: one wait + barrier per tile covers the strip and (first tile) the resident weight images
        mlp_dma_wait();
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            if (l < L) {
                const int K = io.dims[l], N = io.dims[l + 1];
                const float* img = wbase + g.off[l];
                if (!g.resident) {
                    if (l > 0) __syncthreads();   // everyone is done with the previous layer's image
                    mlp_stage_layer(wbase, g.ldw[l], io.W[l], io.b[l], K, N, g.vec16[l]);
                    mlp_dma_wait();
                    __syncthreads();
                }
                mlp_layer_dispatch((N + 15) / 16, strip, lda, img, g.ldw[l], K, N, io.act[l], l == L - 1, io.y, io.y_ld,
                                   row0, io.n_rows);
            }
        }
        if (!g.resident) __syncthreads();   // the next tile restages layer 0 over the last layer's image
    }
    mlp_dma_wait();   // a workgroup without tiles must still drain its DMA before exit
}

__global__ void __launch_bounds__(64 * MLP_WAVES) mlp_forward_kernel(const macjd_mlp_io io, const MlpGeom g) {
    mlp_forward_body(io, g, blockIdx.x, gridDim.x);
}

// two independent chains in one launch: workgroups [0, split) run problem 0, the rest problem 1 (e.g. the actor chain
// and the fc1 -> W_ih chain of the rollout step over the same rows, or the eval and target actors of the learner)
__global__ void __launch_bounds__(64 * MLP_WAVES) mlp_forward_pair_kernel(const macjd_mlp_io io0, const MlpGeom g0,
                                                                          const macjd_mlp_io io1, const MlpGeom g1,
                                                                          const int split) {
    if ((int)blockIdx.x < split) mlp_forward_body(io0, g0, blockIdx.x, split);
    else mlp_forward_body(io1, g1, blockIdx.x - split, gridDim.x - split);
}

}  // namespace macjd

// validates one problem and fills its LDS geometry; *lds_floats = strips + weight images
static int mlp_prepare(const macjd_mlp_io* io, macjd::MlpGeom* gp, size_t* lds_floats) {
    using namespace macjd;
    if (!io) return set_err(MACJD_EINVAL, "%s", "macjd_mlp_forward: NULL io");
    const int L = io->n_layers;
    if (L < 1 || L > 3 || io->n_rows < 0 || !io->x || !io->y)
        return set_err(MACJD_EINVAL, "%s", "macjd_mlp_forward: bad n_layers / n_rows / pointers");
    MlpGeom& g = *gp;
    g = MlpGeom{};
    int total = 0, biggest = 0, widest = 0;
    for (int l = 0; l < L; ++l) {
        const int K = io->dims[l], N = io->dims[l + 1];
        if (!io->W[l] || !io->b[l] || K < 1 || N < 1) return set_err(MACJD_EINVAL, "%s", "macjd_mlp_forward: bad layer");
        if (io->act[l] < 0 || io->act[l] > 2) return set_err(MACJD_EINVAL, "%s", "macjd_mlp_forward: bad activation");
        const bool last = (l == L - 1);
        if ((l == 0 && K > MLP_MAX_IN) || (l > 0 && K > MLP_MAX_HIDDEN) || (!last && N > MLP_MAX_HIDDEN) ||
            (last && N > MLP_MAX_OUT))
            return set_err(MACJD_EUNSUPPORTED, "%s", "macjd_mlp_forward: layer width outside the supported range");
        const int nt = (N + 15) / 16;
        if (!(nt == 1 || nt == 2 || nt == 3 || nt == 4 || nt == 8 || nt == 12 || nt == 24))
            return set_err(MACJD_EUNSUPPORTED, "%s", "macjd_mlp_forward: unsupported output tile count");
        g.ldw[l] = mlp_pitch(mlp_k16(K));
        g.off[l] = total;
        g.vec16[l] = (K % 4 == 0) && (((uintptr_t)io->W[l]) % 16 == 0);
        const int img = mlp_image_floats(K, N);
        total += img;
        biggest = img > biggest ? img : biggest;
        widest = K > widest ? K : widest;   // the strip holds the inputs of every layer
    }
    // x_ld == 0 is a broadcast input (every row the same vector, e.g. the static observation): fine to read
    if ((io->x_ld != 0 && io->x_ld < io->dims[0]) || io->y_ld < io->dims[L])
        return set_err(MACJD_EINVAL, "%s", "macjd_mlp_forward: row stride smaller than the row");
    g.vec16[3] = (io->dims[0] % 4 == 0) && (io->x_ld % 4 == 0) && (((uintptr_t)io->x) % 16 == 0);
    g.lda = mlp_pitch(mlp_k16(widest));
    const int strips_f = MLP_WAVES * 16 * g.lda;
    const int budget_f = MLP_LDS_BYTES / 4 - strips_f;
    g.resident = total <= budget_f;
    if (!g.resident) {
        if (biggest > budget_f)
            return set_err(MACJD_EUNSUPPORTED, "%s", "macjd_mlp_forward: one layer's weights exceed the LDS budget");
        for (int l = 0; l < L; ++l) g.off[l] = 0;
    }
    *lds_floats = (size_t)strips_f + (size_t)(g.resident ? total : biggest);
    return MACJD_OK;
}

extern "C" int macjd_mlp_forward_pair(const macjd_mlp_io* io0, const macjd_mlp_io* io1, void* hip_stream) {
    using namespace macjd;
    MlpGeom g0, g1;
    size_t f0 = 0, f1 = 0;
    int rc = mlp_prepare(io0, &g0, &f0);
    if (rc != MACJD_OK) return rc;
    rc = mlp_prepare(io1, &g1, &f1);
    if (rc != MACJD_OK) return rc;
    if (io0->n_rows == 0 || io1->n_rows == 0)
        return set_err(MACJD_EINVAL, "%s", "macjd_mlp_forward_pair: both problems need rows (use macjd_mlp_forward)");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)mlp_forward_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS_BYTES);
        attr_set = true;
    }
    const int64_t t0 = (io0->n_rows + 63) / 64, t1 = (io1->n_rows + 63) / 64;
    const int b0 = (int)(t0 < 256 ? t0 : 256), b1 = (int)(t1 < 256 ? t1 : 256);
    const size_t lds_bytes = (f0 > f1 ? f0 : f1) * 4;
    hipLaunchKernelGGL(mlp_forward_pair_kernel, dim3((unsigned)(b0 + b1)), dim3(64 * MLP_WAVES), lds_bytes,
                       (hipStream_t)hip_stream, *io0, g0, *io1, g1, b0);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "macjd_mlp_forward_pair: %s", hipGetErrorString(err));
    return MACJD_OK;
}

extern "C" int macjd_mlp_forward(const macjd_mlp_io* io, void* hip_stream) {
    using namespace macjd;
    MlpGeom g;
    size_t lds_floats = 0;
    const int rc = mlp_prepare(io, &g, &lds_floats);
    if (rc != MACJD_OK) return rc;
    if (io->n_rows == 0) return MACJD_OK;
    const size_t lds_bytes = lds_floats * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)mlp_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS_BYTES);
        attr_set = true;
    }
    hipStream_t stream = (hipStream_t)hip_stream;
    const int64_t n_tiles = (io->n_rows + 63) / 64;
    const unsigned grid = (unsigned)(n_tiles < 256 ? n_tiles : 256);
    hipLaunchKernelGGL(mlp_forward_kernel, dim3(grid), dim3(64 * MLP_WAVES), lds_bytes, stream, *io, g);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "macjd_mlp_forward: %s", hipGetErrorString(err));
    return MACJD_OK;
}

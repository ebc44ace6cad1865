// macjd_mlp.hip — fused dense-layer chain on the MI355X matrix cores, exact float32.
//
// y = act_n(W_n ... act_1(W_1 x + b_1) ... + b_n) for up to three layers in ONE forward launch, using
// v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate: bit-for-bit a k-ordered fmaf chain, so the path's 1e-5
// tolerance on Q-values / hidden states holds; bf16 MFMA would not).  C-ABI: include/macjd_nets.h.
//
// Two kernels per call:
//  1. mlp_pack_kernel re-lays every layer's torch [out, in] weight as MFMA B-operand fragments (+ its bias),
//     packed[(kstep * tiles + tile) * 64 + lane] = W[16 tile + (lane & 15)][4 kstep + (lane >> 4)] (zero padded),
//     into a caller-provided workspace.  (Weights change between calls during training, so the pack is part
//     of the call — a few microseconds — rather than a cache that a replayed HIP graph could leave stale.)
//  2. mlp_forward_kernel: a workgroup is 4 waves = 64 rows, each wave owns 16 rows for the whole chain.
//     * staging a layer = LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPR round trip, no
//       index arithmetic) of its packed image, overlapped with the staging of the first input tile; the
//       B-operand read of (kstep, tile) is then `w[(kstep * tiles + tile) * 64 + lane]`: one 256-B row, 64
//       distinct banks, conflict-free by construction, and the tiles of one k-step are immediate offsets;
//     * each wave keeps its 16 x width activation tile in a private LDS strip [16][lda], lda = 32 m + 2, so the
//       A-operand read (lane l -> act[l & 15][4 kstep + (l >> 4)]) also hits 32 distinct banks per 32-lane
//       group; a layer reads its whole input (k-outer loop, all output tiles accumulate in registers)
//       before it overwrites the strip with its output: no double buffer, no barrier inside the chain;
//     * the C tile (col = lane & 15, row = (lane >> 4) * 4 + reg) gets bias + activation in registers and goes
//       to the strip (inner layers) or straight to global memory (last layer).
//     Workgroups are persistent over 64-row tiles.  When all layers' packed weights fit in LDS together they
//     are staged once per workgroup; otherwise (wide scenarios) each layer is staged just before use.
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

// Packed image of one layer: [B fragments, padded to 256 floats][bias, padded to 256 floats].  256 floats = 1 KiB =
// one wave-wide LDS-DMA instruction (64 lanes x 16 B), so an image is a whole number of DMA pieces.
__host__ __device__ inline int mlp_wfrag_floats(int k_in, int n_out) { return ((n_out + 15) / 16) * ((k_in + 3) / 4) * 64; }
__host__ __device__ inline int mlp_wpad_floats(int k_in, int n_out) { return (mlp_wfrag_floats(k_in, n_out) + 255) & ~255; }
__host__ __device__ inline int mlp_bpad_floats(int n_out) { return (((n_out + 15) / 16) * 16 + 255) & ~255; }
__host__ __device__ inline int mlp_packed_floats(int k_in, int n_out) { return mlp_wpad_floats(k_in, n_out) + mlp_bpad_floats(n_out); }

__device__ __forceinline__ float mlp_act(float v, int act) {
    if (act == MACJD_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == MACJD_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

struct MlpPackArgs {
    const float* W[3];
    const float* b[3];
    int K[3], N[3], off[3];  // off = float offset of the layer's packed image in the workspace
    int n_layers, total;     // total packed floats
};

__global__ void __launch_bounds__(256) mlp_pack_kernel(const MlpPackArgs a, float* __restrict__ packed) {
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < a.total; idx += gridDim.x * blockDim.x) {
        int l = 0;
        if (a.n_layers > 1 && idx >= a.off[1]) l = 1;
        if (a.n_layers > 2 && idx >= a.off[2]) l = 2;
        const int K = (l == 0) ? a.K[0] : (l == 1 ? a.K[1] : a.K[2]);
        const int N = (l == 0) ? a.N[0] : (l == 1 ? a.N[1] : a.N[2]);
        const float* W = (l == 0) ? a.W[0] : (l == 1 ? a.W[1] : a.W[2]);
        const float* b = (l == 0) ? a.b[0] : (l == 1 ? a.b[1] : a.b[2]);
        const int local = idx - ((l == 0) ? a.off[0] : (l == 1 ? a.off[1] : a.off[2]));
        const int wfrag = mlp_wfrag_floats(K, N), wpad = mlp_wpad_floats(K, N);
        float v = 0.0f;
        if (local < wfrag) {
            const int lane = local & 63, frag = local >> 6;
            const int tiles = (N + 15) / 16;
            const int ks = frag / tiles, t = frag - ks * tiles;
            const int o = t * 16 + (lane & 15), k = ks * 4 + (lane >> 4);
            v = (o < N && k < K) ? W[(int64_t)o * K + k] : 0.0f;
        } else if (local >= wpad && local - wpad < N) {
            v = b[local - wpad];
        }
        packed[idx] = v;
    }
}

// Whole-workgroup asynchronous copy global -> LDS with LDS-DMA (global_load_lds_dwordx4): each wave
// instruction moves 1 KiB (lane l's 16 bytes land at the wave-uniform LDS base + 16 l), no VGPR round trip.
// The caller waits (s_waitcnt vmcnt(0)) and barriers before the first read.
__device__ __forceinline__ void mlp_stage_dma(float* __restrict__ dst, const float* __restrict__ src, int n_floats) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_pieces = n_floats >> 8;  // images are whole numbers of 256-float pieces
    for (int c = wave; c < n_pieces; c += MLP_WAVES) {
        const float* g = src + c * 256 + lane * 4;
        float* l = dst + c * 256;  // wave-uniform
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
}
__device__ __forceinline__ void mlp_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// one dense layer for this wave's 16 rows; NT = number of 16-column output tiles (compile time)
template <int NT>
__device__ __forceinline__ void mlp_layer(float* __restrict__ strip, int lda, const float* __restrict__ w_lds,
                                          const float* __restrict__ bias_lds, int K, int N, int act, bool last,
                                          float* __restrict__ y, int64_t y_ld, int64_t row0, int64_t n_rows) {
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ksteps = (K + 3) / 4;
    const float* a_ptr = strip + li * lda + lk;
    const float* b_ptr = w_lds + lane;
    // register double buffer: the A value and the NT B fragments of k-step ks+1 are in flight while the NT
    // MFMAs of k-step ks issue (a load-wait-MFMA chain per fragment would expose the LDS latency NT times)
    float a_cur = a_ptr[0], b_cur[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) b_cur[t] = b_ptr[t * 64];
    for (int ks = 0; ks < ksteps; ++ks) {
        float a_nxt = 0.0f, b_nxt[NT];
        const int kn = (ks + 1 < ksteps) ? ks + 1 : ks;   // last iteration re-reads its own fragments (harmless)
        a_nxt = a_ptr[kn * 4];
#pragma unroll
        for (int t = 0; t < NT; ++t) b_nxt[t] = b_ptr[(kn * NT + t) * 64];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur, b_cur[t], acc[t], 0, 0, 0);
        a_cur = a_nxt;
#pragma unroll
        for (int t = 0; t < NT; ++t) b_cur[t] = b_nxt[t];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int o = t * 16 + li;
        const float bv = bias_lds[o];   // zero in the padded columns
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = lk * 4 + r;
            const float v = (o < N) ? mlp_act(acc[t][r] + bv, act) : 0.0f;   // padded columns stay zero
            if (last) {
                if (o < N && row0 + row < n_rows) y[(row0 + row) * y_ld + o] = v;
            } else {
                strip[row * lda + o] = v;
            }
        }
    }
}

__device__ __forceinline__ void mlp_layer_dispatch(int nt, float* strip, int lda, const float* w_lds, const float* bias,
                                                   int K, int N, int act, bool last, float* y, int64_t y_ld,
                                                   int64_t row0, int64_t n_rows) {
    switch (nt) {  // wave-uniform
        case 1: mlp_layer<1>(strip, lda, w_lds, bias, K, N, act, last, y, y_ld, row0, n_rows); break;
        case 2: mlp_layer<2>(strip, lda, w_lds, bias, K, N, act, last, y, y_ld, row0, n_rows); break;
        case 3: mlp_layer<3>(strip, lda, w_lds, bias, K, N, act, last, y, y_ld, row0, n_rows); break;
        case 4: mlp_layer<4>(strip, lda, w_lds, bias, K, N, act, last, y, y_ld, row0, n_rows); break;
        case 8: mlp_layer<8>(strip, lda, w_lds, bias, K, N, act, last, y, y_ld, row0, n_rows); break;
        case 12: mlp_layer<12>(strip, lda, w_lds, bias, K, N, act, last, y, y_ld, row0, n_rows); break;
        case 24: mlp_layer<24>(strip, lda, w_lds, bias, K, N, act, last, y, y_ld, row0, n_rows); break;
        default: break;  // host rejects other tile counts
    }
}

__global__ void __launch_bounds__(64 * MLP_WAVES) mlp_forward_kernel(const macjd_mlp_io io, const float* __restrict__ packed,
                                                                     const int resident, const int lda, const int w_off1,
                                                                     const int w_off2) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* wbase = lds + MLP_WAVES * 16 * lda;          // packed-image region (1 KiB aligned: 64*lda*4 bytes)
    float* strip = lds + wave * 16 * lda;               // this wave's activation strip [16][lda]
    const int L = io.n_layers;
    if (resident) {                                     // all layers' images: issued now, waited for after the
        int total = 0;                                  // first tile's inputs have been staged
#pragma unroll
        for (int l = 0; l < 3; ++l) total += (l < L) ? mlp_packed_floats(io.dims[l], io.dims[l + 1]) : 0;
        mlp_stage_dma(wbase, packed, total);
    }
    bool weights_pending = resident;
    const int64_t n_tiles = (io.n_rows + 63) / 64;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t row0 = tile * 64 + wave * 16;
        // stage this wave's 16 input rows (zero-padded to a multiple of 4 columns / missing rows)
        const int K0 = io.dims[0], K0p = (K0 + 3) & ~3;
#pragma unroll 4
        for (int idx = lane; idx < 16 * K0p; idx += 64) {
            const int r = idx / K0p, k = idx - r * K0p;
            const int64_t row = row0 + r;
            strip[r * lda + k] = (k < K0 && row < io.n_rows) ? io.x[row * io.x_ld + k] : 0.0f;
        }
        if (weights_pending) {
            mlp_dma_wait();
            __syncthreads();
            weights_pending = false;
        }
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            if (l < L) {
                const int K = io.dims[l], N = io.dims[l + 1];
                const int woff = (l == 0) ? 0 : (l == 1 ? w_off1 : w_off2);
                const float* w_lds = wbase + (resident ? woff : 0);
                if (!resident) {
                    __syncthreads();   // everyone is done with the previous layer's image
                    mlp_stage_dma(wbase, packed + woff, mlp_packed_floats(K, N));
                    mlp_dma_wait();
                    __syncthreads();
                }
                mlp_layer_dispatch((N + 15) / 16, strip, lda, w_lds, w_lds + mlp_wpad_floats(K, N), K, N, io.act[l],
                                   l == L - 1, io.y, io.y_ld, row0, io.n_rows);
            }
        }
    }
    if (weights_pending) mlp_dma_wait();   // a workgroup without tiles must still drain its DMA before exit
}

}  // namespace macjd

extern "C" int64_t macjd_mlp_workspace_floats(const macjd_mlp_io* io) {
    if (!io || io->n_layers < 1 || io->n_layers > 3) return -1;
    int64_t total = 0;
    for (int l = 0; l < io->n_layers; ++l) total += macjd::mlp_packed_floats(io->dims[l], io->dims[l + 1]);
    return total;
}

extern "C" int macjd_mlp_forward(const macjd_mlp_io* io, float* workspace, void* hip_stream) {
    using namespace macjd;
    if (!io || !workspace) return set_err(MACJD_EINVAL, "%s", "macjd_mlp_forward: NULL io / workspace");
    const int L = io->n_layers;
    if (L < 1 || L > 3 || io->n_rows < 0 || !io->x || !io->y)
        return set_err(MACJD_EINVAL, "%s", "macjd_mlp_forward: bad n_layers / n_rows / pointers");
    MlpPackArgs pa{};
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
        pa.W[l] = io->W[l]; pa.b[l] = io->b[l]; pa.K[l] = K; pa.N[l] = N; pa.off[l] = total;
        const int pf = mlp_packed_floats(K, N);
        total += pf;
        biggest = pf > biggest ? pf : biggest;
        widest = K > widest ? K : widest;   // strip holds the inputs of every layer
    }
    pa.n_layers = L; pa.total = total;
    if (io->n_rows == 0) return MACJD_OK;
    const int lda = ((widest + 31) / 32) * 32 + 2;
    const int strips_f = MLP_WAVES * 16 * lda;
    const int budget_f = MLP_LDS_BYTES / 4 - strips_f;
    const int resident = total <= budget_f;
    if (!resident && biggest > budget_f)
        return set_err(MACJD_EUNSUPPORTED, "%s", "macjd_mlp_forward: one layer's weights exceed the LDS budget");
    const size_t lds_bytes = (size_t)(strips_f + (resident ? total : biggest)) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)mlp_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS_BYTES);
        attr_set = true;
    }
    hipStream_t stream = (hipStream_t)hip_stream;
    const unsigned pack_grid = (unsigned)((total + 255) / 256 < 512 ? (total + 255) / 256 : 512);
    hipLaunchKernelGGL(mlp_pack_kernel, dim3(pack_grid), dim3(256), 0, stream, pa, workspace);
    const int64_t n_tiles = (io->n_rows + 63) / 64;
    const unsigned grid = (unsigned)(n_tiles < 256 ? n_tiles : 256);
    hipLaunchKernelGGL(mlp_forward_kernel, dim3(grid), dim3(64 * MLP_WAVES), lds_bytes, stream, *io,
                       (const float*)workspace, resident, lda, pa.off[1], pa.off[2]);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "macjd_mlp_forward: %s", hipGetErrorString(err));
    return MACJD_OK;
}

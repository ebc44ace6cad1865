// macjd_mlp.hip — fused dense-layer chain on the MI355X matrix cores, exact float32.
//
// y = act_n(W_n ... act_1(W_1 x + b_1) ... + b_n) for up to three layers in ONE launch, using
// v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate: bit-for-bit a k-ordered fmaf chain, so the path's 1e-5
// tolerance on Q-values / hidden states holds; bf16 MFMA would not).  C-ABI: include/macjd_nets.h.
//
// Tiling (wave64): a workgroup is 4 waves = 64 rows, each wave owns 16 rows for the whole chain.
//   * layer weights live in LDS as [out_pad16][ldw], ldw = roundup32(in) + 2 floats: the B-operand read
//     (lane l -> W[o = tile + (l & 15)][k = step*4 + (l >> 4)]) then hits 32 distinct banks per 32-lane group;
//   * each wave keeps its 16 x width activation tile in a private LDS strip [16][lda], lda = 130: the A-operand
//     read (lane l -> act[l & 15][step*4 + (l >> 4)]) has the same conflict-free shape; a layer reads its
//     whole input (k-outer loop, all output tiles accumulate in registers) before it overwrites the strip
//     with its output, so no double buffer and no barrier inside the chain;
//   * the C tile (col = lane & 15, row = (lane >> 4)*4 + reg) gets bias + activation in registers and goes
//     to the strip (inner layers) or straight to global memory (last layer).
// Workgroups are persistent over 64-row tiles.  When all layers' weights fit in LDS together they are
// staged once per workgroup; otherwise (wide scenarios) each layer is staged just before it is used.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/macjd.h"
#include "../../include/macjd_nets.h"
#include "macjd_err.h"

namespace macjd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MLP_LDA = 130;            // activation strip row stride (floats): 128 + 2
constexpr int MLP_MAX_HIDDEN = 128;
constexpr int MLP_MAX_IN = 256;
constexpr int MLP_MAX_OUT = 384;
constexpr int MLP_WAVES = 4;
constexpr int MLP_LDS_BYTES = 160 * 1024;

__host__ __device__ inline int mlp_ldw(int k_in) { return ((k_in + 31) / 32) * 32 + 2; }
__host__ __device__ inline int mlp_wfloats(int k_in, int n_out) { return ((n_out + 15) / 16) * 16 * mlp_ldw(k_in); }

__device__ __forceinline__ float mlp_act(float v, int act) {
    if (act == MACJD_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == MACJD_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

// cooperative (whole workgroup) staging of one layer's weights into LDS, zero-padded
__device__ __forceinline__ void mlp_stage_weights(float* __restrict__ dst, const float* __restrict__ W, int K, int N) {
    const int ldw = mlp_ldw(K);
    const int rows = ((N + 15) / 16) * 16;
    for (int idx = threadIdx.x; idx < rows * ldw; idx += blockDim.x) {
        const int o = idx / ldw, k = idx - o * ldw;
        dst[idx] = (o < N && k < K) ? W[(int64_t)o * K + k] : 0.0f;
    }
}

// one dense layer for this wave's 16 rows; NT = number of 16-column output tiles (compile time)
template <int NT>
__device__ __forceinline__ void mlp_layer(float* __restrict__ strip, const float* __restrict__ w_lds,
                                          const float* __restrict__ bias, int K, int N, int act, bool last,
                                          float* __restrict__ y, int64_t y_ld, int64_t row0, int64_t n_rows) {
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int ldw = mlp_ldw(K);
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ksteps = (K + 3) / 4;
    const float* a_ptr = strip + li * MLP_LDA + lk;
    const float* b_ptr = w_lds + li * ldw + lk;
    for (int ks = 0; ks < ksteps; ++ks) {
        const float a = a_ptr[ks * 4];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float b = b_ptr[t * 16 * ldw + ks * 4];
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int o = t * 16 + li;
        const float bv = (o < N) ? bias[o] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = lk * 4 + r;
            const float v = (o < N) ? mlp_act(acc[t][r] + bv, act) : 0.0f;   // padded columns stay zero
            if (last) {
                if (o < N && row0 + row < n_rows) y[(row0 + row) * y_ld + o] = v;
            } else {
                strip[row * MLP_LDA + o] = v;
            }
        }
    }
}

__device__ __forceinline__ void mlp_layer_dispatch(int nt, float* strip, const float* w_lds, const float* bias, int K,
                                                   int N, int act, bool last, float* y, int64_t y_ld, int64_t row0,
                                                   int64_t n_rows) {
    switch (nt) {  // wave-uniform
        case 1: mlp_layer<1>(strip, w_lds, bias, K, N, act, last, y, y_ld, row0, n_rows); break;
        case 2: mlp_layer<2>(strip, w_lds, bias, K, N, act, last, y, y_ld, row0, n_rows); break;
        case 3: mlp_layer<3>(strip, w_lds, bias, K, N, act, last, y, y_ld, row0, n_rows); break;
        case 4: mlp_layer<4>(strip, w_lds, bias, K, N, act, last, y, y_ld, row0, n_rows); break;
        case 8: mlp_layer<8>(strip, w_lds, bias, K, N, act, last, y, y_ld, row0, n_rows); break;
        case 12: mlp_layer<12>(strip, w_lds, bias, K, N, act, last, y, y_ld, row0, n_rows); break;
        case 24: mlp_layer<24>(strip, w_lds, bias, K, N, act, last, y, y_ld, row0, n_rows); break;
        default: break;  // host rejects other tile counts
    }
}

__global__ void __launch_bounds__(64 * MLP_WAVES) mlp_forward_kernel(const macjd_mlp_io io, const int resident,
                                                                     const int w_off1, const int w_off2) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* strips = lds;                                        // [MLP_WAVES][16][MLP_LDA]
    float* wbase = lds + MLP_WAVES * 16 * MLP_LDA;              // weights region
    float* strip = strips + wave * 16 * MLP_LDA;
    const int L = io.n_layers;
    const int w_off[3] = {0, w_off1, w_off2};
    if (resident) {
        for (int l = 0; l < L; ++l) mlp_stage_weights(wbase + w_off[l], io.W[l], io.dims[l], io.dims[l + 1]);
        __syncthreads();
    }
    const int64_t n_tiles = (io.n_rows + 63) / 64;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t row0 = tile * 64 + wave * 16;
        // stage this wave's 16 input rows (zero-padded to a multiple of 4 columns / missing rows)
        const int K0 = io.dims[0], K0p = (K0 + 3) & ~3;
        for (int idx = lane; idx < 16 * K0p; idx += 64) {
            const int r = idx / K0p, k = idx - r * K0p;
            const int64_t row = row0 + r;
            strip[r * MLP_LDA + k] = (k < K0 && row < io.n_rows) ? io.x[row * io.x_ld + k] : 0.0f;
        }
        for (int l = 0; l < L; ++l) {
            const int K = io.dims[l], N = io.dims[l + 1];
            const float* w_lds = wbase + (resident ? w_off[l] : 0);
            if (!resident) {
                __syncthreads();   // everyone is done with the previous layer's weights
                mlp_stage_weights(wbase, io.W[l], K, N);
                __syncthreads();
            }
            mlp_layer_dispatch((N + 15) / 16, strip, w_lds, io.b[l], K, N, io.act[l], l == L - 1, io.y, io.y_ld, row0,
                               io.n_rows);
        }
    }
}

}  // namespace macjd

extern "C" int macjd_mlp_forward(const macjd_mlp_io* io, void* hip_stream) {
    using namespace macjd;
    if (!io) return set_err(MACJD_EINVAL, "%s", "macjd_mlp_forward: NULL io");
    const int L = io->n_layers;
    if (L < 1 || L > 3 || io->n_rows < 0 || !io->x || !io->y)
        return set_err(MACJD_EINVAL, "%s", "macjd_mlp_forward: bad n_layers / n_rows / pointers");
    int total = 0, biggest = 0, off[3] = {0, 0, 0};
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
        if (l == 0 && K > MLP_LDA - 2) return set_err(MACJD_EUNSUPPORTED, "%s", "macjd_mlp_forward: input wider than the strip");
        off[l] = total;
        const int wf = mlp_wfloats(K, N);
        total += wf;
        biggest = wf > biggest ? wf : biggest;
    }
    if (io->n_rows == 0) return MACJD_OK;
    const int strips_f = MLP_WAVES * 16 * MLP_LDA;
    const int budget_f = MLP_LDS_BYTES / 4 - strips_f;
    int resident = total <= budget_f;
    if (!resident && biggest > budget_f)
        return set_err(MACJD_EUNSUPPORTED, "%s", "macjd_mlp_forward: one layer's weights exceed the LDS budget");
    const size_t lds_bytes = (size_t)(strips_f + (resident ? total : biggest)) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)mlp_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS_BYTES);
        attr_set = true;
    }
    const int64_t n_tiles = (io->n_rows + 63) / 64;
    const unsigned grid = (unsigned)(n_tiles < 256 ? n_tiles : 256);
    hipLaunchKernelGGL(mlp_forward_kernel, dim3(grid), dim3(64 * MLP_WAVES), lds_bytes, (hipStream_t)hip_stream, *io,
                       resident, off[1], off[2]);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return set_err(MACJD_EDEVICE, "macjd_mlp_forward: %s", hipGetErrorString(err));
    return MACJD_OK;
}

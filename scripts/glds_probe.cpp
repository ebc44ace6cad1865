// LDS-DMA issue-cost probe (timing only): how fast can one wave per SIMD stream L2-resident data into LDS with
// global_load_lds, as a function of how the destination base (M0) and the instruction width are used?
//   mode 0: 1 KiB dwordx4 pieces, LDS base changes every instruction (s_mov m0 per piece)
//   mode 1: 1 KiB dwordx4 pieces, one LDS base per 4 pieces + immediate offsets 0/1024/2048/3072
//   mode 2: padded rows — 512 B dwordx4 per instruction (32 active lanes), base changes every instruction
//   mode 3: padded rows — 184 B dword per instruction (46 active lanes), base changes every instruction
//   mode 4: like mode 2 but through registers: global_load_dwordx4 -> ds_write_b128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define GLDS(g, l, sz, off) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g), (__attribute__((address_space(3))) void*)(l), sz, off, 0)

template <int MODE>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ src, float* __restrict__ out, int reps) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* mine = lds + wave * 8192;   // 32 KiB per wave
    for (int rep = 0; rep < reps; ++rep) {
        if (MODE == 0) {
#pragma unroll 1
            for (int c = 0; c < 24; ++c) GLDS(src + (wave * 24 + c) * 256 + lane * 4, mine + c * 256, 16, 0);
        } else if (MODE == 1) {
#pragma unroll 1
            for (int c = 0; c < 24; c += 4) {
                const float* g = src + (wave * 24 + c) * 256 + lane * 4;
                float* l = mine + c * 256;
                GLDS(g, l, 16, 0); GLDS(g, l, 16, 1024); GLDS(g, l, 16, 2048); GLDS(g, l, 16, 3072);
            }
        } else if (MODE == 2) {
#pragma unroll 1
            for (int r = 0; r < 48; ++r)
                if (lane < 32) GLDS(src + (wave * 48 + r) * 128 + lane * 4, mine + r * 130, 16, 0);
        } else if (MODE == 3) {
#pragma unroll 1
            for (int r = 0; r < 48; ++r)
                if (lane < 46) GLDS(src + (wave * 48 + r) * 46 + lane, mine + r * 66, 4, 0);
        } else {
            float4 v[8];
#pragma unroll 1
            for (int r0 = 0; r0 < 48; r0 += 16) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {   // 2 rows per instruction: lanes 0-31 row r, 32-63 row r+1
                    const int r = r0 + 2 * i + (lane >> 5);
                    v[i] = *(const float4*)(src + (wave * 48 + r) * 128 + (lane & 31) * 4);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = r0 + 2 * i + (lane >> 5);
                    float* d = mine + r * 132 + (lane & 31) * 4;
                    d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (lds[threadIdx.x] == 12345.0f) out[0] = 1.0f;
}

template <int MODE>
void run(const float* src, float* out, const char* what, double kib_per_wave) {
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 64;
    probe<MODE><<<256, 256, 128 * 1024>>>(src, out, reps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<256, 256, 128 * 1024>>>(src, out, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d  %-58s %.3f us per staging round (%.1f KiB per wave)\n", MODE, what, ms * 1000.0 / reps, kib_per_wave);
}

int main() {
    float *src, *out;
    hipMalloc(&src, 1 << 22); hipMemset(src, 0, 1 << 22); hipMalloc(&out, 64);
    run<0>(src, out, "24 x 1 KiB dwordx4, m0 per instruction", 24.0);
    run<1>(src, out, "24 x 1 KiB dwordx4, m0 per 4 + immediate offsets", 24.0);
    run<2>(src, out, "48 x 512 B dwordx4 (32 lanes), m0 per instruction", 24.0);
    run<3>(src, out, "48 x 184 B dword (46 lanes), m0 per instruction", 8.6);
    run<4>(src, out, "48 x 512 B rows via registers (dwordx4 + 4 ds_write_b32)", 24.0);
    return 0;
}

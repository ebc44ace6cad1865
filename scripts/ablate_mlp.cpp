// Timing-only phase ablation of the fused dense-chain kernel (csrc/macjd_mlp.hip).  Build one binary per
// MACJD_MLP_ABLATE value (see scripts/ablate_mlp.sh); each prints the average device time of one macjd_mlp_forward
// call for the actor shape (46-128-128-9, 12288 rows) measured with HIP events over back-to-back launches.
#include <cstdio>
#include <cstdlib>
#include "../ma-cjd-cooperative-jamming-decision-making-via-marl_amd/csrc/macjd_mlp.hip"

namespace macjd { int set_err(int code, const char* fmt, const char* a) { fprintf(stderr, fmt, a); fprintf(stderr, "\n"); return code; } }
int main(int argc, char** argv) {
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 12288;
    const int dims[4] = {46, 128, 128, 9};
    macjd_mlp_io io{};
    io.n_layers = 3; io.n_rows = rows;
    float *x, *y;
    hipMalloc(&x, rows * 184 * 4); hipMemset(x, 0, rows * 184 * 4);
    hipMalloc(&y, rows * 16 * 4);
    io.x = x; io.x_ld = 46; io.y = y; io.y_ld = 9;
    for (int l = 0; l < 3; ++l) {
        float *W, *b;
        hipMalloc(&W, dims[l] * dims[l + 1] * 4); hipMemset(W, 0, dims[l] * dims[l + 1] * 4);
        hipMalloc(&b, dims[l + 1] * 4); hipMemset(b, 0, dims[l + 1] * 4);
        io.W[l] = W; io.b[l] = b; io.act[l] = l < 2 ? MACJD_ACT_RELU : MACJD_ACT_SIGMOID;
    }
    for (int l = 0; l < 4; ++l) io.dims[l] = dims[l];
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) if (macjd_mlp_forward(&io, st)) { printf("error\n"); return 1; }
    hipStreamSynchronize(st);
    const int n = 300;
    hipEventRecord(e0, st);
    for (int i = 0; i < n; ++i) macjd_mlp_forward(&io, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("ablate=%d rows=%lld us_per_call=%.2f\n", MACJD_MLP_ABLATE, (long long)rows, ms * 1000.0f / n);
    return 0;
}

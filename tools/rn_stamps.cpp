// Diagnostic build only: where does residual_norm_kernel spend its time at the verify shape (5 rows, H = 5120, S slabs)?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/rn_stamps.cpp -o /tmp/rn_stamps
#define SD_RN_STAMPS 1
#include "../llmspeculativesampling_amd/csrc/model_kernels.h"
#include <cstdio>
#include <vector>
void sd_set_error(const char *fmt, ...) {}
int main(int argc, char **argv) {
    const int H = 5120, rows = 5, S = argc > 1 ? atoi(argv[1]) : 5, Mpad = 16, threads = 640;
    bf16_t *x, *h, *w; float *part, *junk;
    hipMalloc(&x, (size_t)16 * H * 2); hipMalloc(&h, (size_t)16 * H * 2); hipMalloc(&w, (size_t)H * 2);
    hipMalloc(&part, (size_t)S * Mpad * H * 4); hipMalloc(&junk, (size_t)512 << 20);
    hipMemset(x, 0, (size_t)16 * H * 2); hipMemset(w, 0x3f, (size_t)H * 2); hipMemset(part, 0, (size_t)S * Mpad * H * 4);
    for (int it = 0; it < 5; ++it) {
        hipMemset(junk, it, (size_t)512 << 20);                    // evict: the slabs come from memory, as after a real GEMM
        hipDeviceSynchronize();
        hipLaunchKernelGGL((residual_norm_kernel<bf16_t>), dim3(rows), dim3(threads), 0, 0, x, part, S, (size_t)Mpad * H, H,
                           (const bf16_t *)nullptr, (const bf16_t *)w, (const bf16_t *)nullptr, 1e-5f, NORM_RMS, RES_PRE, h);
        hipDeviceSynchronize();
        long long st[8];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_rn_stamps), sizeof(st));
        printf("S=%d: loads issued+landed %.2f  residual+store x %.2f  block_sum %.2f  norm+store h %.2f | total %.2f us\n", S,
               (st[1] - st[0]) / 100.0, (st[2] - st[1]) / 100.0, (st[3] - st[2]) / 100.0, (st[4] - st[3]) / 100.0,
               (st[4] - st[0]) / 100.0);
    }
    return 0;
}

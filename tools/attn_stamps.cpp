// Diagnostic build only: where does attn_kernel spend its time at the verify shape (5 rows, 196 keys, D = 128, 40 heads)?
// wall_clock64 ticks at 100 MHz.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/attn_stamps.cpp -o gpurun_out/attn_stamps
#define SD_ATT_STAMPS 1
#include "../llmspeculativesampling_amd/csrc/model_kernels.h"
#include <cstdio>
#include <vector>
void sd_set_error(const char *fmt, ...) {}
int main(int argc, char **argv) {
    const int Hq = 40, D = 128, max_seq = 512, rows = argc > 1 ? atoi(argv[1]) : 5, ctx = argc > 2 ? atoi(argv[2]) : 191;
    const size_t kv_elems = (size_t)2 * Hq * max_seq * D;
    bf16_t *kv, *q, *out;
    hipMalloc(&kv, kv_elems * 2); hipMalloc(&q, (size_t)16 * Hq * D * 2); hipMalloc(&out, (size_t)16 * Hq * D * 2);
    std::vector<unsigned short> h(kv_elems);
    for (size_t i = 0; i < kv_elems; ++i) h[i] = 0x3c00 + (i * 2654435761u >> 22) % 512;       // bf16 values around 0.01
    hipMemcpy(kv, h.data(), kv_elems * 2, hipMemcpyHostToDevice);
    hipMemcpy(q, h.data(), (size_t)16 * Hq * D * 2, hipMemcpyHostToDevice);
    RowTab tab = {};
    tab.n_rows = rows; tab.n_streams = 1; tab.n_groups = 1;
    tab.kv_base[0] = kv; tab.max_seq[0] = max_seq;
    for (int i = 0; i < rows; ++i) { tab.row_pos[i] = ctx + i; tab.row_stream[i] = 0; }
    tab.grp_row0[0] = 0; tab.grp_n[0] = rows; tab.grp_pos[0] = ctx; tab.grp_stream[0] = 0;
    const int s_max = ctx + rows, s_cap = (s_max + 63) / 64 * 64;
    const size_t lds = sizeof(float) * ((size_t)ATT_TQ * D + (size_t)(256 / (D / 8)) * ATT_TQ * D + (size_t)ATT_TQ * s_cap);
    hipFuncSetAttribute(reinterpret_cast<const void *>(attn_kernel<bf16_t, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((attn_kernel<bf16_t, 128>), dim3(Hq, 1, 1), dim3(256), lds, 0, q, tab, 0, out, Hq, Hq, 0, 0.0883883f, s_cap, 1, (float *)nullptr);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long st[16];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_att_stamps), sizeof(st));
        printf("rows %d keys %d  event %.2f us | in-kernel:", rows, s_max, ms * 1e3);
        const char *names[] = {"issue V/q/K loads + QK^T", "barrier", "softmax", "barrier", "P.V", "barrier", "fold + store"};
        for (int i = 1; i <= 7; ++i) printf("  %s %.2f", names[i - 1], (st[i] - st[i - 1]) / 100.0);
        printf("  | total %.2f us\n", (st[7] - st[0]) / 100.0);
    }
    return 0;
}

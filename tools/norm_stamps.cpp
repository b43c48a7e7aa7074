// Diagnostic build only: where does norm_probs_kernel spend its time?  (wall_clock64 = 100 MHz)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/norm_stamps.cpp -o tools/norm_stamps_bin
#define SD_STAMPS 1
#include "../llmspeculativesampling_amd/csrc/sampling.hip"
#include <vector>
#include <random>
void sd_set_error(const char *fmt, ...) {}
static void report(const char *tag) {
    long long st[32];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
    printf("%-22s", tag);
    for (int i = 1; i <= 6; ++i) printf("  [%d-%d] %6.2f", i - 1, i, (st[i] - st[i - 1]) / 100.0);
    printf("   total %.2f us\n", (st[6] - st[0]) / 100.0);
}
int main() {
    const int V = 32000;
    std::vector<float> h(V), tmax(V / 16);
    std::mt19937 g(1); std::normal_distribution<float> nd(0.f, 4.f);
    for (auto &v : h) v = nd(g);
    for (int t = 0; t < V / 16; ++t) { float m = -1e30f; for (int c = 0; c < 16; ++c) m = fmaxf(m, h[t * 16 + c]); tmax[t] = m; }
    float *x, *o, *tm; int *tok, *err; void *ws;
    hipMalloc(&x, V * 4); hipMalloc(&o, V * 4); hipMalloc(&tm, V / 16 * 4); hipMalloc(&tok, 4); hipMalloc(&err, 16);
    hipMalloc(&ws, sd_norm_workspace_bytes(1));
    hipMemcpy(x, h.data(), V * 4, hipMemcpyHostToDevice);
    hipMemcpy(tm, tmax.data(), V / 16 * 4, hipMemcpyHostToDevice);
    for (int use_ws = 0; use_ws < 2; ++use_ws) {
        for (int it = 0; it < 3; ++it) {
            sd_norm_sample(x, V, 1.0f, 20, 0.9f, 0, o, err, nullptr, 1, 2, tok, err + 1, use_ws ? ws : nullptr, nullptr);
            hipDeviceSynchronize();
        }
        report(use_ws ? "ws (norm_cand) 1024" : "single kernel 1024");
    }
    int ref; hipMemcpy(&ref, tok, 4, hipMemcpyDeviceToHost);
    for (const char *thr : {"1024", "512", "256", "128"}) {
        g_norm_tile_threads = atoi(thr);
        for (int it = 0; it < 3; ++it) {
            hipMemset(o, 0, V * 4);
            sd_norm_rows_with_tiles(x, 1, V, V, 1.0f, 20, 0.9f, 0, o, V, err, 1, 2, tok, err + 1, nullptr, tm, nullptr, nullptr);
            hipDeviceSynchronize();
        }
        char tag[64]; int t2; hipMemcpy(&t2, tok, 4, hipMemcpyDeviceToHost);
        snprintf(tag, sizeof tag, "tiles %s thr (%s)", thr, t2 == ref ? "tok ok" : "TOK DIFF");
        report(tag);
    }
    return 0;
}

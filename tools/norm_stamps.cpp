// Diagnostic build only: where does norm_probs_kernel spend its time?  (wall_clock64 = 100 MHz)
#define SD_STAMPS 1
#include "../llmspeculativesampling_amd/csrc/sampling.hip"
#include <vector>
#include <random>
void sd_set_error(const char *fmt, ...) {}
int main() {
    const int V = 32000;
    std::vector<float> h(V);
    std::mt19937 g(1); std::normal_distribution<float> nd(0.f, 4.f);
    for (auto &v : h) v = nd(g);
    float *x, *o; int *tok, *err; void *ws;
    hipMalloc(&x, V * 4); hipMalloc(&o, V * 4); hipMalloc(&tok, 4); hipMalloc(&err, 16); hipMalloc(&ws, sd_norm_workspace_bytes(1));
    hipMemcpy(x, h.data(), V * 4, hipMemcpyHostToDevice);
    for (int use_ws = 0; use_ws < 2; ++use_ws) {
        for (int it = 0; it < 3; ++it) {
            sd_norm_sample(x, V, 1.0f, 20, 0.9f, 0, o, err, nullptr, 1, 2, tok, err + 1, use_ws ? ws : nullptr, nullptr);
            hipDeviceSynchronize();
        }
        long long st[32];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
        printf("ws=%d:", use_ws);
        for (int i = 1; i <= 6; ++i) printf("  [%d-%d] %.2f us", i - 1, i, (st[i] - st[i - 1]) / 100.0);
        printf("   total %.2f us\n", (st[6] - st[0]) / 100.0);
    }
    return 0;
}

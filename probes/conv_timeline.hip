// Probe: the timeline of ONE block of conv_igemm_dma_kernel on layer3's launches (round 6).  Shader-clock stamps of thread 0 of every block (ABL bit 5):
//   0 entry | 1 prologue fills issued | 2 first stage landed + barrier | 3 k-loop issued | 4 last MFMA done | 5 epilogue tile stored (issued) | 6 stores acknowledged |
//   7 statistics folded, atomics acknowledged
// next to the launch's duration by HIP events, and the spread of the block entries (the launch ramp).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I../open-solution-mapping-challenge_amd/csrc -I../include conv_timeline.hip -o conv_timeline
#include "../open-solution-mapping-challenge_amd/csrc/api.hip"
#include "../open-solution-mapping-challenge_amd/csrc/igemm.hip"
#include <algorithm>
#include <vector>

namespace msc_conv {      // the other translation units' kernels are not part of this probe
bool conv1x1_cfg_ok(const ConvK&, int) { return false; }
int conv1x1_launch(const ConvK&, int, hipStream_t) { return -1; }
int halo32_conv_launch(const ConvK&, int, hipStream_t) { return -1; }
int halo32_deconv_launch(const ConvK&, int, hipStream_t) { return -1; }
int halo32_stem_launch(const ConvK&, int, hipStream_t) { return -1; }
int halo32_down_launch(const ConvK&, int, hipStream_t) { return -1; }
}

struct Shape { const char* name; int N, H, W, Cin, Cout, K; bool stats; };

template <int TP, int TC, int WP, int WC, int KB, int NST>
void run_cfg(const char* cfgname, const Shape& s, ConvK k) {
    if (k.Cout % TC || ((long)k.Cin * 2) % KB) return;
    const int blocks = ((k.M + TP - 1) / TP) * (k.Cout / TC);
    unsigned long long* dbg;
    hipMalloc(&dbg, (size_t)blocks * 8 * sizeof(unsigned long long));
    hipMemset(dbg, 0, (size_t)blocks * 8 * sizeof(unsigned long long));
    ConvK kd = k;
    kd.kws = reinterpret_cast<float*>(dbg);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f, best32 = 1e9f;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(a); launch_dma<bf16_t, TP, TC, WP, WC, KB, NST, 0>(k, 0, 0); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = std::min(best, ms * 1e3f);
        hipEventRecord(a); launch_dma<bf16_t, TP, TC, WP, WC, KB, NST, 32>(kd, 0, 0); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b); best32 = std::min(best32, ms * 1e3f);
    }
    std::vector<unsigned long long> h((size_t)blocks * 8);
    hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0min = ~0ull, t0max = 0, tend = 0;
    for (int i = 0; i < blocks; ++i) { t0min = std::min(t0min, h[i * 8]); t0max = std::max(t0max, h[i * 8]); tend = std::max(tend, h[i * 8 + 7]); }
    printf("%-22s %-20s %4d blocks  product %6.1f us  stamped %6.1f us | entry spread %6llu clk, first entry -> last exit %7llu clk | median clk since entry:", s.name, cfgname, blocks, best, best32,
           t0max - t0min, tend - t0min);
    for (int j = 1; j < 8; ++j) {
        std::vector<unsigned long long> v(blocks);
        for (int i = 0; i < blocks; ++i) v[i] = h[i * 8 + j] - h[i * 8];
        std::sort(v.begin(), v.end());
        printf(" %d:%llu", j, v[blocks / 2]);
    }
    printf("\n");
    hipFree(dbg);
}

int main() {
    std::vector<Shape> shapes = {
        {"layer3 1x1 1024->256 bn", 32, 16, 16, 1024, 256, 1, true},
        {"layer3 1x1 256->1024 bn", 32, 16, 16, 256, 1024, 1, true},
        {"layer3 1x1 1024->256", 32, 16, 16, 1024, 256, 1, false},
        {"layer2 1x1 512->128 bn", 32, 32, 32, 512, 128, 1, true},
    };
    for (const Shape& s : shapes) {
        const size_t in_n = (size_t)s.N * s.H * s.W * s.Cin, wt_n = (size_t)s.Cout * s.K * s.K * s.Cin, out_n = (size_t)s.N * s.H * s.W * s.Cout;
        bf16_t *in, *wt, *out;
        double* slots;
        hipMalloc(&in, in_n * 2); hipMalloc(&wt, wt_n * 2); hipMalloc(&out, out_n * 2); hipMalloc(&slots, (size_t)MSC_BN_SLOTS * s.Cout * 2 * 8);
        hipMemset(slots, 0, (size_t)MSC_BN_SLOTS * s.Cout * 2 * 8);
        std::vector<bf16_t> h(in_n > wt_n ? in_n : wt_n);
        unsigned x = 12345u;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (bf16_t)(0x3c00u + ((x >> 16) & 0x3ffu) | ((x >> 5) & 0x8000u)); }
        hipMemcpy(in, h.data(), in_n * 2, hipMemcpyHostToDevice);
        hipMemcpy(wt, h.data(), wt_n * 2, hipMemcpyHostToDevice);
        msc_conv_desc d = {};
        d.in = in; d.wt = wt; d.out = out; d.in_ld = s.Cin; d.out_ld = s.Cout; d.dtype = MSC_BF16; d.mode = 0;
        d.N = s.N; d.Hi = d.Ho = s.H; d.Wi = d.Wo = s.W; d.Cin = s.Cin; d.Cout = s.Cout; d.KH = d.KW = s.K; d.stride = 1; d.pad = s.K / 2;
        if (s.stats) d.stats = slots;
        else d.relu = 1;
        ConvK k;
        if (conv_fill(&d, &k) != MSC_OK) { printf("%s: %s\n", s.name, msc_last_error()); return 1; }
        run_cfg<128, 64, 4, 2, 128, 3>("128x64 8w KB128 x3", s, k);
        run_cfg<128, 64, 4, 2, 256, 3>("128x64 8w KB256 x3", s, k);
        run_cfg<256, 128, 4, 2, 128, 3>("256x128 8w KB128 x3", s, k);
        hipFree(in); hipFree(wt); hipFree(out); hipFree(slots);
    }
    return 0;
}

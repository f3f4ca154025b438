// Probe: where does the time of conv_igemm_dma_kernel go?  The product kernel (ABL = 0) next to ablated instantiations
// of the same template -- no MFMA / no DMA after the prologue / DMA + barriers only / no barrier -- on the layer shapes
// that dominate the ResNet101-U-Net train step.  Timing only: ablated variants compute garbage.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../open-solution-mapping-challenge_amd/csrc -I../include conv_ablate.hip -o conv_ablate
#include "../open-solution-mapping-challenge_amd/csrc/api.hip"
#include "../open-solution-mapping-challenge_amd/csrc/igemm.hip"
#include <vector>

struct Shape { const char* name; int N, H, W, Cin, Cout, K; };

template <int TP, int TC, int WP, int WC, int KB, int NST, int ABL>
float time_one(const ConvK& k, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch_dma<bf16_t, TP, TC, WP, WC, KB, NST, ABL>(k, 0, 0);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) launch_dma<bf16_t, TP, TC, WP, WC, KB, NST, ABL>(k, 0, 0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 1e3f * ms / reps;
}

template <int TP, int TC, int WP, int WC, int KB, int NST>
void run_cfg(const char* cfgname, const Shape& s, const ConvK& k) {
    if (k.Cout % TC || ((long)k.Cin * 2) % KB) return;
    const int reps = 20;
    const float full = time_one<TP, TC, WP, WC, KB, NST, 0>(k, reps);
    const float nomma = time_one<TP, TC, WP, WC, KB, NST, 1>(k, reps);
    const float nodma = time_one<TP, TC, WP, WC, KB, NST, 2>(k, reps);
    const float dmaonly = time_one<TP, TC, WP, WC, KB, NST, 4>(k, reps);
    const float nobar = time_one<TP, TC, WP, WC, KB, NST, 8>(k, reps);
    const float loop = time_one<TP, TC, WP, WC, KB, NST, 6>(k, reps);
    const float pinned = time_one<TP, TC, WP, WC, KB, NST, 16>(k, reps);
    const double gf = 2.0 * (double)k.M * k.Cout * k.Cin * k.KH * k.KW / 1e9;
    printf("%-22s %-26s full %7.1f us (%6.0f TF) | no-MFMA %7.1f | no-DMA %7.1f | DMA+barrier only %7.1f | no-barrier %7.1f | loop+epilogue %7.1f | pinned pieces %7.1f\n",
           s.name, cfgname, full, gf / full * 1e3, nomma, nodma, dmaonly, nobar, loop, pinned);
}

int main() {
    std::vector<Shape> shapes = {
        {"dec1 3x3 128->128", 32, 128, 128, 128, 128, 3},
        {"dec2 3x3 320->128", 32, 64, 64, 320, 128, 3},
        {"dec3 3x3 768->256", 32, 32, 32, 768, 256, 3},
        {"layer3 3x3 256->256", 32, 16, 16, 256, 256, 3},
        {"layer3 1x1 1024->256", 32, 16, 16, 1024, 256, 1},
        {"layer3 1x1 256->1024", 32, 16, 16, 256, 1024, 1},
        {"layer1 1x1 64->256", 32, 64, 64, 64, 256, 1},
        {"layer2 1x1 128->512", 32, 32, 32, 128, 512, 1},
    };
    for (const Shape& s : shapes) {
        const size_t in_n = (size_t)s.N * s.H * s.W * s.Cin, wt_n = (size_t)s.Cout * s.K * s.K * s.Cin, out_n = (size_t)s.N * s.H * s.W * s.Cout;
        bf16_t *in, *wt, *out;
        hipMalloc(&in, in_n * 2); hipMalloc(&wt, wt_n * 2); hipMalloc(&out, out_n * 2);
        std::vector<bf16_t> h(in_n > wt_n ? in_n : wt_n);
        unsigned x = 12345u;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (bf16_t)(0x3c00u + ((x >> 16) & 0x3ffu) | ((x >> 5) & 0x8000u)); }   // random bf16, |v| in [0.0078, 0.03]
        hipMemcpy(in, h.data(), in_n * 2, hipMemcpyHostToDevice);
        hipMemcpy(wt, h.data(), wt_n * 2, hipMemcpyHostToDevice);
        msc_conv_desc d = {};
        d.in = in; d.wt = wt; d.out = out; d.in_ld = s.Cin; d.out_ld = s.Cout; d.dtype = MSC_BF16; d.mode = 0;
        d.N = s.N; d.Hi = d.Ho = s.H; d.Wi = d.Wo = s.W; d.Cin = s.Cin; d.Cout = s.Cout; d.KH = d.KW = s.K; d.stride = 1; d.pad = s.K / 2; d.relu = 1;
        ConvK k;
        if (conv_fill(&d, &k) != MSC_OK) { printf("%s: %s\n", s.name, msc_last_error()); return 1; }
        run_cfg<256, 128, 4, 2, 128, 3>("256x128 8w KB128 x3", s, k);
        run_cfg<256, 256, 2, 4, 128, 2>("256x256 8w KB128 x2", s, k);
        run_cfg<128, 128, 2, 2, 64, 4>("128x128 4w KB64 x4", s, k);
        run_cfg<64, 128, 2, 4, 256, 3>("64x128 8w KB256 x3", s, k);
        run_cfg<64, 128, 2, 4, 128, 3>("64x128 8w KB128 x3", s, k);
        run_cfg<64, 64, 2, 2, 128, 4>("64x64 4w KB128 x4", s, k);
        hipFree(in); hipFree(wt); hipFree(out);
    }
    return 0;
}

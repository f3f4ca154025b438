// Probe: where does the time of the weight-gradient kernel (conv_wgrad_dma_kernel) go?  Product kernel next to ablated
// instantiations (no MFMA / no DMA after the prologue / DMA + barriers only / no atomics) on the dominant layer shapes, with
// the split-K policy of the grouped launch (64 k-steps per block).  Timing only.
#include "../open-solution-mapping-challenge_amd/csrc/api.hip"
#include "../open-solution-mapping-challenge_amd/csrc/igemm.hip"
#include <vector>

struct Shape { const char* name; int N, H, W, A, B, K; };

template <int TA, int TB, int ABL>
float time_one(const WgK& k, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((conv_wgrad_dma_kernel<bf16_t, TA, TB, 4, ABL>), dim3(k.nblocks), dim3(256), 0, 0, k);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((conv_wgrad_dma_kernel<bf16_t, TA, TB, 4, ABL>), dim3(k.nblocks), dim3(256), 0, 0, k);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 1e3f * ms / reps;
}

template <int TA, int TB>
void run_tile(const Shape& s, const msc_wgrad_desc& d, int steps) {
    WgPlan pl;
    if (wgrad_plan(&d, steps, TA, &pl) != MSC_OK) { printf("%s: %s\n", s.name, msc_last_error()); return; }
    if (pl.ta != TA || pl.tb != TB) return;
    const int reps = 10;
    const float full = time_one<TA, TB, 0>(pl.k, reps), nomma = time_one<TA, TB, 1>(pl.k, reps), nodma = time_one<TA, TB, 2>(pl.k, reps),
                dmaonly = time_one<TA, TB, 4>(pl.k, reps), noatom = time_one<TA, TB, 32>(pl.k, reps), skel = time_one<TA, TB, 6 | 32>(pl.k, reps);
    const double gf = 2.0 * (double)pl.k.M * s.A * s.B * s.K * s.K / 1e9;
    printf("%-24s tile %3dx%-3d steps/block %3d blocks %6d | full %7.1f us (%5.0f TF) | no-MFMA %7.1f | no-DMA %7.1f | DMA only %7.1f | no-atomics %7.1f | loop only %7.1f\n",
           s.name, TA, TB, steps, pl.k.nblocks, full, gf / full * 1e3, nomma, nodma, dmaonly, noatom, skel);
}

int main() {
    std::vector<Shape> shapes = {
        {"layer3 3x3 256x256", 32, 16, 16, 256, 256, 3}, {"layer3 1x1 1024->256", 32, 16, 16, 256, 1024, 1}, {"dec1 3x3 128x128", 32, 128, 128, 128, 128, 3},
        {"dec3 3x3 768->256", 32, 32, 32, 256, 768, 3},   {"layer2 3x3 128x128", 32, 32, 32, 128, 128, 3},   {"layer1 3x3 64x64", 32, 64, 64, 64, 64, 3},
    };
    for (const Shape& s : shapes) {
        const size_t pn = (size_t)s.N * s.H * s.W * s.A, qn = (size_t)s.N * s.H * s.W * s.B, wn = (size_t)s.A * s.K * s.K * s.B;
        bf16_t *P, *Q; float* dw;
        hipMalloc(&P, pn * 2); hipMalloc(&Q, qn * 2); hipMalloc(&dw, wn * 4); hipMemset(dw, 0, wn * 4);
        std::vector<bf16_t> h(pn > qn ? pn : qn);
        unsigned x = 777u;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (bf16_t)((0x3c00u + ((x >> 16) & 0x3ffu)) | ((x >> 5) & 0x8000u)); }
        hipMemcpy(P, h.data(), pn * 2, hipMemcpyHostToDevice); hipMemcpy(Q, h.data(), qn * 2, hipMemcpyHostToDevice);
        msc_wgrad_desc d = {};
        d.p = P; d.q = Q; d.dw = dw; d.p_ld = s.A; d.q_ld = s.B; d.dtype = MSC_BF16;
        d.N = s.N; d.Hp = d.Hq = s.H; d.Wp = d.Wq = s.W; d.A = s.A; d.B = s.B; d.KH = d.KW = s.K; d.stride = 1; d.pad = s.K / 2;
        for (int steps : {64, 256}) { run_tile<128, 128>(s, d, steps); run_tile<64, 64>(s, d, steps); }
        hipFree(P); hipFree(Q); hipFree(dw);
    }
    return 0;
}

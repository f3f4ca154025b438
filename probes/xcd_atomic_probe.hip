// Probe: per-XCD accumulation slots.  Every workgroup adds into slot[xcc_id][channel]; all contributions to one address
// come from ONE XCD, so the atomic may run in that XCD's L2 (workgroup-scope encoding, no sc1) instead of at the device
// coherence point.  Checks exactness of the sums after the kernel boundary and times agent scope vs workgroup scope vs
// one shared slot (the cross-XCD same-address case that serialises).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7; }   // HW_REG_XCC_ID, bits 0..3
template <int SCOPE, int SHARED>
__global__ __launch_bounds__(256) void add_kernel(float* slots, int C, int reps, int* xcc_seen) {
    const int x = SHARED ? 0 : xcc_id();
    if (threadIdx.x == 0) xcc_seen[blockIdx.x] = xcc_id();
    for (int r = 0; r < reps; ++r)
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float* a = slots + ((long)x * C + c) * 2;
            if (SCOPE == 0) { __hip_atomic_fetch_add(a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_fetch_add(a + 1, 2.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            else { __hip_atomic_fetch_add(a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_add(a + 1, 2.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        }
}
__global__ void read_kernel(const float* slots, int C, float* out) {     // the consumer: a later kernel sums the 8 slots
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        float s1 = 0.f, s2 = 0.f;
        for (int x = 0; x < 8; ++x) { s1 += slots[((long)x * C + c) * 2]; s2 += slots[((long)x * C + c) * 2 + 1]; }
        out[2 * c] = s1; out[2 * c + 1] = s2;
    }
}
template <int SCOPE, int SHARED>
void run(const char* name, int blocks, int C, int reps) {
    float *slots, *out; int* seen;
    hipMalloc(&slots, 8L * C * 2 * 4); hipMalloc(&out, C * 2 * 4); hipMalloc(&seen, blocks * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f; bool ok = true;
    for (int it = 0; it < 5; ++it) {
        hipMemsetAsync(slots, 0, 8L * C * 2 * 4, 0);
        hipEventRecord(a);
        hipLaunchKernelGGL((add_kernel<SCOPE, SHARED>), dim3(blocks), dim3(256), 0, 0, slots, C, reps, seen);
        hipEventRecord(b);
        hipLaunchKernelGGL(read_kernel, dim3(64), dim3(256), 0, 0, slots, C, out);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        std::vector<float> h(2 * C); hipMemcpy(h.data(), out, C * 8, hipMemcpyDeviceToHost);
        for (int c = 0; c < C; ++c) if (h[2 * c] != (float)blocks * reps || h[2 * c + 1] != 2.0f * blocks * reps) ok = false;
    }
    std::vector<int> hs(blocks); hipMemcpy(hs.data(), seen, blocks * 4, hipMemcpyDeviceToHost);
    int hist[8] = {0}; int match = 0;
    for (int i = 0; i < blocks; ++i) { hist[hs[i] & 7]++; match += (hs[i] == (i & 7)); }
    printf("%-34s blocks %5d C %5d reps %3d: %8.1f us  sums %s  | xcc hist %d %d %d %d %d %d %d %d, block%%8==xcc for %d/%d\n", name, blocks, C, reps, best * 1e3f,
           ok ? "exact" : "WRONG", hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], match, blocks);
    hipFree(slots); hipFree(out); hipFree(seen);
}
int main() {
    for (int blocks : {256, 2048}) for (int C : {256, 1024}) {
        run<0, 0>("agent scope, per-XCD slots", blocks, C, 4);
        run<1, 0>("workgroup scope, per-XCD slots", blocks, C, 4);
        run<0, 1>("agent scope, ONE slot", blocks, C, 4);
    }
    return 0;
}

// Float atomic adds from every workgroup of the chip into one small gradient-like array (the pattern of the backward
// kernels' dW / small-parameter adds): device scope into ONE array, against workgroup scope into one replica PER XCD (the
// XCD id read from the hardware register), summed afterwards.  Reports time and checks the sums.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

__device__ __forceinline__ int xcc_id() { return (int)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)); }  // HW_REG_XCC_ID[3:0]

template <int MODE>  // 0: agent scope, one array; 1: workgroup scope, per-XCD replica; 2: agent scope, per-XCD replica
__global__ __launch_bounds__(512) void k(float* __restrict__ g, int n_addr, int per_wg, int* __restrict__ xcc_seen) {
    const int xcc = xcc_id();
    if (threadIdx.x == 0) atomicAdd(&xcc_seen[xcc & 15], 1);
    float* dst = MODE == 0 ? g : g + (size_t)xcc * n_addr;
    for (int i = threadIdx.x; i < per_wg; i += blockDim.x) {
        const int a = (int)(((unsigned)i * 2654435761u + blockIdx.x * 97u) % (unsigned)n_addr);
        if (MODE == 1) __hip_atomic_fetch_add(&dst[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(&dst[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main() {
    const int n_addr = 60000, per_wg = 60000, wgs = 3643;
    float* g; int* seen;
    hipMalloc(&g, sizeof(float) * n_addr * 16);
    hipMalloc(&seen, sizeof(int) * 16);
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(g, 0, sizeof(float) * n_addr * 16);
        hipMemset(seen, 0, sizeof(int) * 16);
        hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
        hipEventRecord(s);
        if (mode == 0) k<0><<<wgs, 512>>>(g, n_addr, per_wg, seen);
        if (mode == 1) k<1><<<wgs, 512>>>(g, n_addr, per_wg, seen);
        if (mode == 2) k<2><<<wgs, 512>>>(g, n_addr, per_wg, seen);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        std::vector<float> h((size_t)n_addr * 16);
        std::vector<int> hs(16);
        hipMemcpy(h.data(), g, sizeof(float) * n_addr * 16, hipMemcpyDeviceToHost);
        hipMemcpy(hs.data(), seen, sizeof(int) * 16, hipMemcpyDeviceToHost);
        double tot = 0;
        for (size_t i = 0; i < h.size(); ++i) tot += h[i];
        const double want = (double)wgs * per_wg;
        printf("mode %d: %.3f ms, %.1f G atomics/s, sum %.0f (want %.0f) %s | WGs per XCC:", mode, ms, want / ms / 1e6, tot, want,
               std::fabs(tot - want) < 0.5 ? "OK" : "WRONG");
        for (int i = 0; i < 16; ++i) if (hs[i]) printf(" %d:%d", i, hs[i]);
        printf("\n");
    }
    return 0;
}

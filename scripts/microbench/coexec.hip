// Does a matrix instruction co-execute with vector ALU work on gfx950?  Times N MFMAs alone, M v_fma alone, and both
// interleaved in one wave (2 waves per SIMD), for the f32 16x16x4 and the bf16 16x16x32 MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int MODE, bool BF16, int NF>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
    bf8 pa, pb;
    for (int i = 0; i < 8; ++i) { pa[i] = (__bf16)(a + i); pb[i] = (__bf16)(b - i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE & 1) {
                if (BF16) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, pb, acc[u], 0, 0, 0);
                else acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u], 0, 0, 0);
            }
            if (MODE & 2) {
#pragma unroll
                for (int i = 0; i < NF; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a));  // NF independent v_fma per MFMA (asm: no SLP packing)
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE, bool BF16, int NF>
float run(float* d, int iters, int threads) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    k<MODE, BF16, NF><<<256, threads>>>(d, iters);
    hipEventRecord(s);
    k<MODE, BF16, NF><<<256, threads>>>(d, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms;
}
template <int NF>
void report(float* d, int it, int threads) {
    printf("%d waves/SIMD, %d fma per mfma | f32 16x16x4: mfma %.3f valu %.3f both %.3f | bf16 16x16x32: mfma %.3f valu %.3f both %.3f ms\n",
           threads / 256, NF, run<1, false, NF>(d, it, threads), run<2, false, NF>(d, it, threads), run<3, false, NF>(d, it, threads),
           run<1, true, NF>(d, it, threads), run<2, true, NF>(d, it, threads), run<3, true, NF>(d, it, threads));
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int it = 20000;
    for (int threads : {256, 512}) { report<1>(d, it, threads); report<2>(d, it, threads); report<4>(d, it, threads); report<8>(d, it, threads); }
    return 0;
}

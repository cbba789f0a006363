// Development microbenchmark (not product): does gfx950 interlock a 16-deep MFMA that accumulates onto the result of a 32-deep one?
// hipcc 7.2 issues `v_mfma_f32_16x16x16_bf16 D, a, b, D` straight behind `v_mfma_f32_16x16x32_bf16 D, ..` with no wait state
// (same vDst as SrcC: it treats the pair like a same-opcode accumulation chain).  Each kernel below runs the chain
// k32, k32, <N wait states>, k16 on the same accumulator and compares it, bit for bit, with the same chain behind 2 x s_nop 15.
//   hipcc -O3 --offload-arch=gfx950 mfma_chain_hazard.hip -o mfma_chain_hazard && ./mfma_chain_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CHAIN(NAME, K32, K16, T8, T4, WAIT)                                                                        \
    __device__ __forceinline__ f4 NAME(T8 a0, T8 b0, T8 a1, T8 b1, T4 a2, T4 b2) {                                \
        f4 acc;                                                                                                    \
        asm volatile(K32 " %0, %1, %2, 0\n\t" K32 " %0, %3, %4, %0\n\t" WAIT K16 " %0, %5, %6, %0\n\t"            \
                     "s_nop 15\n\ts_nop 15"                                                                        \
                     : "=&v"(acc) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2));                         \
        return acc;                                                                                                \
    }
#define BF32 "v_mfma_f32_16x16x32_bf16"
#define BF16 "v_mfma_f32_16x16x16_bf16"
#define HF32 "v_mfma_f32_16x16x32_f16"
#define HF16 "v_mfma_f32_16x16x16_f16"
CHAIN(bf_safe, BF32, BF16, bf8, bf4, "s_nop 15\n\ts_nop 15\n\t")
CHAIN(bf_w0, BF32, BF16, bf8, bf4, "")
CHAIN(bf_w1, BF32, BF16, bf8, bf4, "s_nop 0\n\t")
CHAIN(bf_w2, BF32, BF16, bf8, bf4, "s_nop 1\n\t")
CHAIN(bf_w4, BF32, BF16, bf8, bf4, "s_nop 3\n\t")
CHAIN(bf_w8, BF32, BF16, bf8, bf4, "s_nop 7\n\t")
CHAIN(hf_safe, HF32, HF16, h8, h4, "s_nop 15\n\ts_nop 15\n\t")
CHAIN(hf_w0, HF32, HF16, h8, h4, "")
CHAIN(hf_w2, HF32, HF16, h8, h4, "s_nop 1\n\t")
CHAIN(hf_w4, HF32, HF16, h8, h4, "s_nop 3\n\t")
CHAIN(hf_w8, HF32, HF16, h8, h4, "s_nop 7\n\t")

// 10 counters: mismatching lanes of bf w0, w1, w2, w4, w8, hf w0, w2, w4, w8, and the number of lanes compared
__global__ __launch_bounds__(512) void probe(const float* __restrict__ src, unsigned long long* __restrict__ bad, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long cnt[10] = {};
    for (int it = 0; it < iters; ++it) {
        const float* p = src + ((size_t)(tid * 7 + it * 131) % 4096) * 8;
        bf8 a0, b0, a1, b1; bf4 a2, b2; h8 c0, d0, c1, d1; h4 c2, d2;
        for (int e = 0; e < 8; ++e) {
            a0[e] = (__bf16)p[e]; b0[e] = (__bf16)p[8 + e]; a1[e] = (__bf16)p[16 + e]; b1[e] = (__bf16)p[24 + e];
            c0[e] = (_Float16)p[e]; d0[e] = (_Float16)p[8 + e]; c1[e] = (_Float16)p[16 + e]; d1[e] = (_Float16)p[24 + e];
        }
        for (int e = 0; e < 4; ++e) { a2[e] = (__bf16)p[32 + e]; b2[e] = (__bf16)p[36 + e]; c2[e] = (_Float16)p[32 + e]; d2[e] = (_Float16)p[36 + e]; }
        const f4 rs = bf_safe(a0, b0, a1, b1, a2, b2), hs = hf_safe(c0, d0, c1, d1, c2, d2);
        const f4 r[5] = {bf_w0(a0, b0, a1, b1, a2, b2), bf_w1(a0, b0, a1, b1, a2, b2), bf_w2(a0, b0, a1, b1, a2, b2), bf_w4(a0, b0, a1, b1, a2, b2), bf_w8(a0, b0, a1, b1, a2, b2)};
        const f4 h[4] = {hf_w0(c0, d0, c1, d1, c2, d2), hf_w2(c0, d0, c1, d1, c2, d2), hf_w4(c0, d0, c1, d1, c2, d2), hf_w8(c0, d0, c1, d1, c2, d2)};
        for (int k = 0; k < 5; ++k)
            for (int j = 0; j < 4; ++j) cnt[k] += __builtin_bit_cast(unsigned, r[k][j]) != __builtin_bit_cast(unsigned, rs[j]);
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 4; ++j) cnt[5 + k] += __builtin_bit_cast(unsigned, h[k][j]) != __builtin_bit_cast(unsigned, hs[j]);
        cnt[9] += 4;
    }
    for (int k = 0; k < 10; ++k)
        if (cnt[k]) atomicAdd(bad + k, cnt[k]);
}

int main() {
    std::vector<float> host(4096 * 8 + 64);
    srand(5);
    for (auto& v : host) v = (float)(rand() % 2001 - 1000) / 500.f;
    float* src; unsigned long long* bad;
    hipMalloc(&src, host.size() * sizeof(float)); hipMalloc(&bad, 10 * sizeof(unsigned long long));
    hipMemcpy(src, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
    const char* names[10] = {"bf16 k32->k16, 0 wait states", "bf16 1", "bf16 2", "bf16 4", "bf16 8", "f16 k32->k16, 0 wait states", "f16 2", "f16 4", "f16 8", "values compared"};
    for (int threads : {64, 512}) {  // one wave per workgroup / eight (two per SIMD)
        hipMemset(bad, 0, 10 * sizeof(unsigned long long));
        hipLaunchKernelGGL(probe, dim3(2048), dim3(threads), 0, 0, src, bad, 64);
        unsigned long long out[10];
        hipMemcpy(out, bad, sizeof(out), hipMemcpyDeviceToHost);
        printf("workgroups of %d threads:\n", threads);
        for (int k = 0; k < 10; ++k) printf("  %-32s %llu\n", names[k], out[k]);
    }
    return 0;
}

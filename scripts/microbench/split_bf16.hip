// Can three bf16 pieces per operand and six bf16 MFMAs replace an exact-fp32 MFMA layer on gfx950, and what does it buy?
// One 64 x 64 layer applied to a 16-read tile per wave, `iters` times: (a) 64 x v_mfma_f32_16x16x4_f32, (b) 2 K-blocks x 4 out
// tiles x 6 x v_mfma_f32_16x16x32_bf16 plus the on-the-fly split of the activations (the weights are split beforehand).
// Prints the time of both and their error against a double-precision product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float wval(int m, int k) { return __sinf(0.37f * m + 1.3f * k) * 0.2f + 0.013f * ((m * 7 + k * 3) % 11); }
__device__ __forceinline__ float xval(int k, int n, int it) { return __cosf(0.11f * k + 0.7f * n + it) * 1.7f + 0.001f * k; }

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// fragments: fp32 [kt][mt][lane][4], bf16 [kb][mt][3][lane][8]
__global__ void pack(float* wf, __bf16* wb) {
    for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < 4 * 4 * 64 * 4; i += blockDim.x * gridDim.x) {
        const int j = i & 3, lane = (i >> 2) & 63, mt = (i >> 8) & 3, kt = i >> 10;
        wf[i] = wval(16 * mt + (lane & 15), 16 * kt + 4 * j + (lane >> 4));
    }
    for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < 2 * 4 * 64 * 8; i += blockDim.x * gridDim.x) {
        const int e = i & 7, lane = (i >> 3) & 63, mt = (i >> 9) & 3, kb = i >> 11;
        __bf16 h, m, l;
        split3(wval(16 * mt + (lane & 15), 32 * kb + 8 * (lane >> 4) + e), h, m, l);
        const size_t base = ((size_t)(kb * 4 + mt) * 3) * 512 + lane * 8 + e;
        wb[base] = h; wb[base + 512] = m; wb[base + 1024] = l;
    }
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void layer(const float* __restrict__ wf, const __bf16* __restrict__ wb, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63, n = lane & 15, kg = lane >> 4;
    f4 acc[4];
    float carry = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = f4{0.f, 0.f, 0.f, 0.f};
        if (MODE == 0) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                f4 b;
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = xval(16 * kt + 4 * j + kg, n, it & 3) + carry;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const f4 a = reinterpret_cast<const f4*>(wf)[(kt * 4 + mt) * 64 + lane];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc[mt], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                bf8 bh, bm, bl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    __bf16 h, m, l;
                    split3(xval(32 * kb + 8 * kg + e, n, it & 3) + carry, h, m, l);
                    bh[e] = h; bm[e] = m; bl[e] = l;
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const bf8* p = reinterpret_cast<const bf8*>(wb) + (size_t)(kb * 4 + mt) * 3 * 64 + lane;
                    const bf8 ah = p[0], am = p[64], al = p[128];
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[mt], 0, 0, 0);
                }
            }
        }
        carry = (acc[0][0] + acc[3][3]) * 1e-30f;  // a dependency between the iterations that does not change the values
    }
    if (blockIdx.x == 0 && threadIdx.x < 64)
        for (int mt = 0; mt < 4; ++mt)
            for (int J = 0; J < 4; ++J) out[(16 * mt + 4 * kg + J) * 16 + n] = acc[mt][J];
}

int main() {
    float *wf, *out; __bf16* wb;
    hipMalloc(&wf, 4 * 4 * 64 * 4 * 4); hipMalloc(&wb, 2 * 4 * 3 * 512 * 2); hipMalloc(&out, 64 * 16 * 4);
    pack<<<8, 256>>>(wf, wb);
    const int iters = 2000;
    std::vector<float> res[2];
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(s);
            if (mode == 0) layer<0><<<512, 512>>>(wf, wb, out, iters); else layer<1><<<512, 512>>>(wf, wb, out, iters);
            hipEventRecord(e); hipEventSynchronize(e);
        }
        float ms; hipEventElapsedTime(&ms, s, e);
        res[mode].resize(64 * 16);
        hipMemcpy(res[mode].data(), out, 64 * 16 * 4, hipMemcpyDeviceToHost);
        printf("%s: %.3f ms for %d layers x 4096 waves  (%.1f ns per wave-layer)\n", mode == 0 ? "fp32 mfma 16x16x4 " : "bf16 x 6 16x16x32", ms,
               iters, ms * 1e6 / iters / 4096 * 1024 / 1024);
    }
    // reference in double on the host: last iteration has it = iters - 1
    double e0 = 0, e1 = 0, scale = 0;
    const int it = (iters - 1) & 3;
    for (int m = 0; m < 64; ++m)
        for (int n = 0; n < 16; ++n) {
            double ref = 0;
            for (int k = 0; k < 64; ++k) {
                const float w = sinf(0.37f * m + 1.3f * k) * 0.2f + 0.013f * ((m * 7 + k * 3) % 11);
                const float x = cosf(0.11f * k + 0.7f * n + it) * 1.7f + 0.001f * k;
                ref += (double)w * (double)x;
            }
            e0 = fmax(e0, fabs(res[0][m * 16 + n] - ref)); e1 = fmax(e1, fabs(res[1][m * 16 + n] - ref)); scale = fmax(scale, fabs(ref));
        }
    printf("max |error| vs double: fp32 mfma %.3e, bf16 x 6 %.3e  (values up to %.2f; host sinf/cosf differ from the device's in the last bits)\n", e0, e1, scale);
    return 0;
}

// Can an fp32 product run on the f16 matrix pipe as TWO f16 pieces per operand (three MFMAs) instead of three bf16 pieces (six)?
//   numerics: y = W x (60 x 64 weights ~ N(0, 0.13), 16 reads) against fp64 for
//     a) three bf16 pieces, six MFMAs (what the forward kernels do today)
//     b) two f16 pieces, three MFMAs, weights as they are (their low pieces are f16 denormals)
//     c) two f16 pieces, three MFMAs, weights scaled by 2^10 before the split (low pieces normal), result scaled back
//     d) as c) with the fourth product (lo x lo)
//   at three activation magnitudes; then the FP16_OVFL mode bit (overflow clamps to 65504 instead of inf); then timing of
//   split + MFMAs for a) and c) at 4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 f16x2.hip -o f16x2 && ./f16x2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
#define DEV __device__ __forceinline__

DEV void split_bf3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x; const float r = x - (float)h; m = (__bf16)r; l = (__bf16)(r - (float)m);
}
DEV void split_h2(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x; l = (_Float16)(x - (float)h);
}

// A operand: lane (m = lane & 15, kg = lane >> 4) holds A[m][8 kg + e]; B operand: lane (n, kg) holds B[8 kg + e][n]
template <int MODE>
__global__ void numerics(const float* __restrict__ W, const float* __restrict__ X, float* __restrict__ Y, int K, float wscale) {
    const int lane = threadIdx.x & 63, m = lane & 15, kg = lane >> 4;
    for (int mt = 0; mt < 4; ++mt) {
        f4 acc = {0, 0, 0, 0};
        for (int kb = 0; kb < K / 32; ++kb) {
            float a[8], b[8];
            for (int e = 0; e < 8; ++e) {
                a[e] = W[(16 * mt + m) * K + 32 * kb + 8 * kg + e] * wscale;
                b[e] = X[(32 * kb + 8 * kg + e) * 16 + m];
            }
            if (MODE == 0) {
                bf8 ah, am, al, bh, bm, bl;
                for (int e = 0; e < 8; ++e) {
                    __bf16 h, mm, l;
                    split_bf3(a[e], h, mm, l); ah[e] = h; am[e] = mm; al[e] = l;
                    split_bf3(b[e], h, mm, l); bh[e] = h; bm[e] = mm; bl[e] = l;
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
            } else {
                h8 ah, al, bh, bl;
                for (int e = 0; e < 8; ++e) {
                    _Float16 h, l;
                    split_h2(a[e], h, l); ah[e] = h; al[e] = l;
                    split_h2(b[e], h, l); bh[e] = h; bl[e] = l;
                }
                if (MODE == 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
            }
        }
        for (int j = 0; j < 4; ++j) Y[(16 * mt + 4 * kg + j) * 16 + m] = acc[j] / wscale;  // C: col = lane & 15, row = 4 (lane >> 4) + j
    }
}

__global__ void ovfl(float* out) {
    // FP16_OVFL = MODE bit 23: an overflowing f16 result is clamped to +-65504 instead of +-inf
    float big = 1.0e6f + threadIdx.x, neg = -3.0e5f;
    _Float16 a = (_Float16)big;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    _Float16 b, c;
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(b) : "v"(big));
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(c) : "v"(neg));
    h2 p;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p) : "v"(big), "v"(neg));
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0");
    if (threadIdx.x == 0) { out[0] = (float)a; out[1] = (float)b; out[2] = (float)c; out[3] = (float)p[0]; out[4] = (float)p[1]; }
}

// timing: per iteration one "k block step" for two read tiles: split 2 x 8 activations, NTO output tiles
template <int MODE, int NTO>
__global__ __launch_bounds__(512, 4) void timing(float* out, const float* __restrict__ frag, int iters) {
    f4 acc[2][NTO];
    for (int r = 0; r < 2; ++r) for (int t = 0; t < NTO; ++t) acc[r][t] = f4{0, 0, 0, 0};
    float x[2][8];
    for (int r = 0; r < 2; ++r) for (int e = 0; e < 8; ++e) x[r][e] = threadIdx.x * 0.001f + e + r;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            bf8 bh[2], bm[2], bl[2];
            for (int r = 0; r < 2; ++r)
                for (int e = 0; e < 8; e += 2) {
                    const bf2 h = {(__bf16)x[r][e], (__bf16)x[r][e + 1]};
                    const float ra = x[r][e] - (float)h[0], rb = x[r][e + 1] - (float)h[1];
                    const bf2 mm = {(__bf16)ra, (__bf16)rb};
                    const bf2 l = {(__bf16)(ra - (float)mm[0]), (__bf16)(rb - (float)mm[1])};
                    bh[r][e] = h[0]; bh[r][e + 1] = h[1]; bm[r][e] = mm[0]; bm[r][e + 1] = mm[1]; bl[r][e] = l[0]; bl[r][e + 1] = l[1];
                }
            for (int t = 0; t < NTO; ++t) {
                const bf8* fp = reinterpret_cast<const bf8*>(frag) + ((it * NTO + t) % 64) * 192 + lane;
                const bf8 ah = fp[0], am = fp[64], al = fp[128];
                for (int r = 0; r < 2; ++r) {
                    acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[r], acc[r][t], 0, 0, 0);
                    acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[r], acc[r][t], 0, 0, 0);
                    acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm[r], acc[r][t], 0, 0, 0);
                    acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh[r], acc[r][t], 0, 0, 0);
                    acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[r], acc[r][t], 0, 0, 0);
                    acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[r], acc[r][t], 0, 0, 0);
                }
            }
        } else {
            h8 bh[2], bl[2];
            for (int r = 0; r < 2; ++r)
                for (int e = 0; e < 8; e += 2) {
                    const h2 h = {(_Float16)x[r][e], (_Float16)x[r][e + 1]};
                    const h2 l = {(_Float16)(x[r][e] - (float)h[0]), (_Float16)(x[r][e + 1] - (float)h[1])};
                    bh[r][e] = h[0]; bh[r][e + 1] = h[1]; bl[r][e] = l[0]; bl[r][e + 1] = l[1];
                }
            for (int t = 0; t < NTO; ++t) {
                const h8* fp = reinterpret_cast<const h8*>(frag) + ((it * NTO + t) % 64) * 128 + lane;
                const h8 ah = fp[0], al = fp[64];
                for (int r = 0; r < 2; ++r) {
                    acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[r], acc[r][t], 0, 0, 0);
                    acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[r], acc[r][t], 0, 0, 0);
                    acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[r], acc[r][t], 0, 0, 0);
                }
            }
        }
        for (int r = 0; r < 2; ++r) for (int e = 0; e < 8; ++e) x[r][e] = x[r][e] * 0.999f + acc[r][0][e & 3] * 1e-9f;
    }
    float s = 0;
    for (int r = 0; r < 2; ++r) for (int t = 0; t < NTO; ++t) s += acc[r][t][0] + acc[r][t][1] + acc[r][t][2] + acc[r][t][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int NTO>
float time_it(float* d, const float* frag, int iters) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    timing<MODE, NTO><<<2048, 512>>>(d, frag, iters);
    hipEventRecord(s);
    timing<MODE, NTO><<<2048, 512>>>(d, frag, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms;
}

int main() {
    const int K = 64, M = 64, N = 16;
    std::vector<float> W(M * K), X(K * N), Y(M * N);
    float *dW, *dX, *dY;
    hipMalloc(&dW, W.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&dY, Y.size() * 4);
    srand(1);
    auto gauss = [] { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307 * v); };
    for (float xs : {1.0f, 1e-3f, 300.0f}) {
        for (auto& w : W) w = (float)(0.13 * gauss());
        for (auto& x : X) x = (float)(xs * gauss());
        hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
        std::vector<double> ref(M * N), refabs(M * N);
        std::vector<float> f32(M * N);
        for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
            double s = 0, sa = 0; float sf = 0;
            for (int k = 0; k < K; ++k) { s += (double)W[m * K + k] * X[k * N + n]; sa += fabs((double)W[m * K + k] * X[k * N + n]); sf = fmaf(W[m * K + k], X[k * N + n], sf); }
            ref[m * N + n] = s; refabs[m * N + n] = sa; f32[m * N + n] = sf;
        }
        auto report = [&](const char* name) {
            hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
            double rms = 0, mx = 0, rms32 = 0;
            for (int i = 0; i < M * N; ++i) {  // error relative to the sum of |terms| (the scale fp32 accumulation errors live on)
                const double e = fabs(Y[i] - ref[i]) / refabs[i], e32 = fabs(f32[i] - ref[i]) / refabs[i];
                rms += e * e; rms32 += e32 * e32; mx = e > mx ? e : mx;
            }
            printf("  x scale %-6g %-44s rms %.3e max %.3e   (sequential fp32 fma: rms %.3e)\n", xs, name, sqrt(rms / (M * N)), mx, sqrt(rms32 / (M * N)));
        };
        numerics<0><<<1, 64>>>(dW, dX, dY, K, 1.0f); report("bf16 x 3 pieces, 6 MFMAs");
        numerics<1><<<1, 64>>>(dW, dX, dY, K, 1.0f); report("f16 x 2 pieces, 3 MFMAs, W as is");
        numerics<1><<<1, 64>>>(dW, dX, dY, K, 1024.0f); report("f16 x 2 pieces, 3 MFMAs, W * 2^10");
        numerics<2><<<1, 64>>>(dW, dX, dY, K, 1024.0f); report("f16 x 2 pieces, 4 MFMAs, W * 2^10");
    }
    ovfl<<<1, 64>>>(dY);
    hipMemcpy(Y.data(), dY, 5 * 4, hipMemcpyDeviceToHost);
    printf("f16 overflow: default cvt(1e6) = %g; with FP16_OVFL: cvt(1e6) = %g, cvt(-3e5) = %g, cvt_pk = (%g, %g)\n", Y[0], Y[1], Y[2], Y[3], Y[4]);

    float *dout, *dfrag;
    hipMalloc(&dout, 2048 * 512 * 4); hipMalloc(&dfrag, 64 * 192 * 64 * 16); hipMemset(dfrag, 0, 64 * 192 * 64 * 16);
    const int it = 4000;
    printf("timing, 2048 x 512 threads, %d steps of (split 2 tiles x 8 values, NTO x 2 tiles MFMA groups):\n", it);
    printf("  NTO 2: bf16x3 %.3f ms   f16x2 %.3f ms\n", time_it<0, 2>(dout, dfrag, it), time_it<1, 2>(dout, dfrag, it));
    printf("  NTO 4: bf16x3 %.3f ms   f16x2 %.3f ms\n", time_it<0, 4>(dout, dfrag, it), time_it<1, 4>(dout, dfrag, it));
    return 0;
}

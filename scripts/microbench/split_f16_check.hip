// Development aid: the two forms of the f16 two-piece split (pmt_device.hpp: split_pair_f16, inline asm against plain C++), value by
// value over random floats of every magnitude, with MODE.FP16_OVFL set as in the kernels.  hipcc --offload-arch=gfx950 -O3 -o split_check split_f16_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
__device__ void split_asm(float a, float b, unsigned& h, unsigned& l, float k4096) {
    float ra, rb;
    asm("v_cvt_pk_f16_f32 %0, %3, %4\n\t"
        "v_fma_mix_f32 %1, %0, -1.0, %3 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %2, %0, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(h), "=&v"(ra), "=&v"(rb) : "v"(a), "v"(b));
    asm("v_fma_mixlo_f16 %0, %1, %3, 0\n\t"
        "v_fma_mixhi_f16 %0, %2, %3, 0"
        : "=&v"(l) : "v"(ra), "v"(rb), "v"(k4096));
}
__device__ void split_c(float a, float b, unsigned& h, unsigned& l) {
    const h2v hh = {(_Float16)a, (_Float16)b};
    const h2v ll = {(_Float16)((a - (float)hh[0]) * 4096.f), (_Float16)((b - (float)hh[1]) * 4096.f)};
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, ll);
}
__global__ void k(const float* x, int n, unsigned* out, int ovfl) {
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    float k4096 = 4096.f;
    asm volatile("" : "+v"(k4096));
    unsigned h1, l1, h2, l2;
    split_asm(x[2 * i], x[2 * i + 1], h1, l1, k4096);
    split_c(x[2 * i], x[2 * i + 1], h2, l2);
    out[4 * i] = h1; out[4 * i + 1] = l1; out[4 * i + 2] = h2; out[4 * i + 3] = l2;
}
static float h2f(unsigned short u) { _Float16 h; std::memcpy((void*)&h, (const void*)&u, 2); return (float)h; }
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const int e = rand() % 60 - 40;  // 2^-40 .. 2^19
        const float m = 1.f + (rand() / (float)RAND_MAX);
        x[i] = ((rand() & 1) ? -1.f : 1.f) * ldexpf(m, e);
    }
    x[0] = 0.f; x[1] = -0.f; x[2] = 65504.f; x[3] = 65520.f; x[4] = 1e6f; x[5] = -1e6f; x[6] = 6e-5f; x[7] = 5.9e-8f;
    float* dx; unsigned* dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 2 * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    for (int ovfl = 0; ovfl < 2; ++ovfl) {
        hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, dout, ovfl);
        std::vector<unsigned> out(2 * n);
        hipMemcpy(out.data(), dout, n * 2 * 4, hipMemcpyDeviceToHost);
        long diff_bits = 0; double worst_asm = 0, worst_c = 0; int shown = 0;
        for (int i = 0; i < n / 2; ++i) {
            for (int half = 0; half < 2; ++half) {
                const float v = x[2 * i + half];
                const unsigned short ha = out[4 * i] >> (16 * half), la = out[4 * i + 1] >> (16 * half), hc = out[4 * i + 2] >> (16 * half), lc = out[4 * i + 3] >> (16 * half);
                const double ra = (double)h2f(ha) + (double)h2f(la) / 4096.0, rc = (double)h2f(hc) + (double)h2f(lc) / 4096.0;
                if (ha != hc || la != lc) {
                    ++diff_bits;
                    if (shown < 8) { printf("  ovfl %d x = %.9g: asm (%04x, %04x) -> %.9g   c++ (%04x, %04x) -> %.9g\n", ovfl, v, ha, la, ra, hc, lc, rc); ++shown; }
                }
                if (fabsf(v) < 65504.f && fabsf(v) > 1e-7f) {
                    worst_asm = fmax(worst_asm, fabs(ra - v) / fabs(v));
                    worst_c = fmax(worst_c, fabs(rc - v) / fabs(v));
                }
            }
        }
        printf("FP16_OVFL %d: %ld of %d values split differently; worst relative reconstruction error asm %.3e, c++ %.3e\n", ovfl, diff_bits, n, worst_asm, worst_c);
    }
    return 0;
}

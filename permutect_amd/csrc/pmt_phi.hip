// Materialised parametrizations ("phi") of the artifact model, forward and backward, one launch each.
//
// The reference registers torch parametrizations on a handful of small tensors (reference architecture/
// feature_clustering.py:41-75, exponentially_modified_gaussian.py:66-75, gated_mlp.py:196-201,
// euclidean_transformation.py:14-16): exp, bounded sigmoid, unit rows, log_softmax and torch's `orthogonal`
// (matrix_exp map with trivialization).  Evaluated through torch autograd they cost ~150 tiny launches per training step
// (the matrix exponential and its adjoint alone are ~100), i.e. ~2.5 ms of launch gaps next to an 8 ms step.  Here one
// workgroup per tensor evaluates its parametrization (forward) or adds J^T d(phi) into the flat gradient buffer
// (backward).  The matrix exponential runs in fp64 (scaling and squaring, Taylor order 18), so it agrees with torch's
// fp32 result to fp32 rounding; its adjoint is the upper-right block of expm([[A^T, G], [0, A^T]]), the identity torch
// itself differentiates matrix_exp with.
#include <hip/hip_runtime.h>
#include <math.h>

#include "permutect_amd.h"

#define PHI_THREADS 256
#define PHI_MAXN (2 * PMT_MAX_ORTHO_DIM)  // the adjoint works on a 2n x 2n block matrix

__device__ double block_max(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = PHI_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}
__device__ double block_sum(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = PHI_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// C = A * B * scale  (N x N, row-major, LDS)
__device__ void matmul(double* __restrict__ Cm, const double* __restrict__ A, const double* __restrict__ B, int N, double scale) {
    for (int idx = threadIdx.x; idx < N * N; idx += PHI_THREADS) {
        const int i = idx / N, j = idx - i * N;
        // four independent partial sums: the LDS round trip of a load (~100 cycles) is the cost of a step, and one dependent chain
        // of N = 20 of them was most of this single-workgroup kernel's 70 us
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        const double* ar = A + i * N;
        const double* bc = B + j;
        int k = 0;
        for (; k + 3 < N; k += 4) {
            a0 += ar[k] * bc[k * N];
            a1 += ar[k + 1] * bc[(k + 1) * N];
            a2 += ar[k + 2] * bc[(k + 2) * N];
            a3 += ar[k + 3] * bc[(k + 3) * N];
        }
        for (; k < N; ++k) a0 += ar[k] * bc[k * N];
        Cm[idx] = ((a0 + a1) + (a2 + a3)) * scale;
    }
    __syncthreads();
}

// S <- expm(S) in place; B, P, T: three more N x N work matrices.
__device__ void expm_inplace(double* S, double* B, double* P, double* T, int N, double* red) {
    // 1-norm (max column sum) -> scaling so that |B|_1 <= 0.5
    double colsum = 0.0;
    if (threadIdx.x < N)
        for (int i = 0; i < N; ++i) colsum += fabs(S[i * N + threadIdx.x]);
    const double norm1 = block_max(colsum, red);
    int s = 0;
    if (norm1 > 0.5) s = (int)ceil(log2(norm1 / 0.5));
    const double scale = ldexp(1.0, -s);
    for (int idx = threadIdx.x; idx < N * N; idx += PHI_THREADS) {
        const double b = S[idx] * scale;
        B[idx] = b;
        P[idx] = b;
        const int i = idx / N, j = idx - i * N;
        S[idx] = b + (i == j ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int k = 2; k <= 12; ++k) {  // remainder 0.5^13 / 13! ~ 2e-14 of a result that is delivered in fp32 (was 18 terms: 1.6e-23)
        matmul(T, P, B, N, 1.0 / (double)k);
        for (int idx = threadIdx.x; idx < N * N; idx += PHI_THREADS) {
            P[idx] = T[idx];
            S[idx] += T[idx];
        }
        __syncthreads();
    }
    for (int q = 0; q < s; ++q) {
        matmul(T, S, S, N, 1.0);
        for (int idx = threadIdx.x; idx < N * N; idx += PHI_THREADS) S[idx] = T[idx];
        __syncthreads();
    }
}

struct PhiShared {
    double m[4][PHI_MAXN * PHI_MAXN];
    double red[PHI_THREADS];
};
static_assert(sizeof(PhiShared) <= 160 * 1024, "LDS budget");

__global__ __launch_bounds__(PHI_THREADS) void pmt_phi_forward_kernel(PmtPhiProgram prog, const float* __restrict__ theta,
                                                                       float* __restrict__ phi) {
    __shared__ __attribute__((aligned(16))) PhiShared sh;
    const PmtPhiSeg sg = prog.seg[blockIdx.x];
    const float* x = theta + sg.theta_off;
    float* y = phi + sg.phi_off;
    const int tid = threadIdx.x, n = sg.rows * sg.cols;
    if (sg.kind == PMT_PHI_EXP) {
        for (int i = tid; i < n; i += PHI_THREADS) y[i] = expf(x[i]);
    } else if (sg.kind == PMT_PHI_BOUNDED) {
        for (int i = tid; i < n; i += PHI_THREADS) y[i] = sg.p1 * (1.0f / (1.0f + expf(-x[i]))) + sg.p0;
    } else if (sg.kind == PMT_PHI_UNIT_ROWS) {
        for (int r = 0; r < sg.rows; ++r) {
            double q = 0.0;
            for (int i = tid; i < sg.cols; i += PHI_THREADS) q += (double)x[r * sg.cols + i] * (double)x[r * sg.cols + i];
            const float norm = (float)sqrt(block_sum(q, sh.red));
            for (int i = tid; i < sg.cols; i += PHI_THREADS) y[r * sg.cols + i] = x[r * sg.cols + i] / norm;
        }
    } else if (sg.kind == PMT_PHI_LOG_SOFTMAX) {
        for (int r = 0; r < sg.rows; ++r) {
            double mx = -INFINITY;
            for (int i = tid; i < sg.cols; i += PHI_THREADS) mx = fmax(mx, (double)x[r * sg.cols + i]);
            mx = block_max(mx, sh.red);
            double se = 0.0;
            for (int i = tid; i < sg.cols; i += PHI_THREADS) se += exp((double)x[r * sg.cols + i] - mx);
            const double lse = mx + log(block_sum(se, sh.red));
            for (int i = tid; i < sg.cols; i += PHI_THREADS) y[r * sg.cols + i] = (float)((double)x[r * sg.cols + i] - lse);
        }
    } else if (sg.kind == PMT_PHI_ORTHOGONAL) {
        const int N = sg.rows;
        double* S = sh.m[0];
        for (int idx = tid; idx < N * N; idx += PHI_THREADS) {  // A = tril(X) - tril(X)^T
            const int i = idx / N, j = idx - i * N;
            S[idx] = i > j ? (double)x[i * N + j] : (i < j ? -(double)x[j * N + i] : 0.0);
        }
        __syncthreads();
        expm_inplace(S, sh.m[1], sh.m[2], sh.m[3], N, sh.red);
        for (int idx = tid; idx < N * N; idx += PHI_THREADS) {  // W = base @ Q
            const int i = idx / N, j = idx - i * N;
            double acc = 0.0;
            if (sg.base != nullptr) {
                for (int k = 0; k < N; ++k) acc += (double)sg.base[i * sg.base_rs + k * sg.base_cs] * S[k * N + j];
            } else {
                acc = S[idx];
            }
            y[idx] = (float)acc;
        }
    }
}

__global__ __launch_bounds__(PHI_THREADS) void pmt_phi_backward_kernel(PmtPhiProgram prog, const float* __restrict__ theta,
                                                                        const float* __restrict__ phi,
                                                                        const float* __restrict__ gphi,
                                                                        float* __restrict__ gtheta) {
    __shared__ __attribute__((aligned(16))) PhiShared sh;
    const PmtPhiSeg sg = prog.seg[blockIdx.x];
    const float* x = theta + sg.theta_off;
    const float* y = phi + sg.phi_off;
    const float* gy = gphi + sg.phi_off;
    float* gx = gtheta + sg.theta_off;
    const int tid = threadIdx.x, n = sg.rows * sg.cols;
    if (sg.kind == PMT_PHI_EXP) {
        for (int i = tid; i < n; i += PHI_THREADS) gx[i] += gy[i] * y[i];
    } else if (sg.kind == PMT_PHI_BOUNDED) {
        for (int i = tid; i < n; i += PHI_THREADS) {
            const float s = 1.0f / (1.0f + expf(-x[i]));
            gx[i] += gy[i] * sg.p1 * s * (1.0f - s);
        }
    } else if (sg.kind == PMT_PHI_UNIT_ROWS) {
        for (int r = 0; r < sg.rows; ++r) {
            double q = 0.0, d = 0.0;
            for (int i = tid; i < sg.cols; i += PHI_THREADS) {
                q += (double)x[r * sg.cols + i] * (double)x[r * sg.cols + i];
                d += (double)y[r * sg.cols + i] * (double)gy[r * sg.cols + i];
            }
            const double norm = sqrt(block_sum(q, sh.red));
            const double dot = block_sum(d, sh.red);
            for (int i = tid; i < sg.cols; i += PHI_THREADS)
                gx[r * sg.cols + i] += (float)(((double)gy[r * sg.cols + i] - (double)y[r * sg.cols + i] * dot) / norm);
        }
    } else if (sg.kind == PMT_PHI_LOG_SOFTMAX) {
        for (int r = 0; r < sg.rows; ++r) {
            double g = 0.0;
            for (int i = tid; i < sg.cols; i += PHI_THREADS) g += (double)gy[r * sg.cols + i];
            const double gsum = block_sum(g, sh.red);
            for (int i = tid; i < sg.cols; i += PHI_THREADS)
                gx[r * sg.cols + i] += (float)((double)gy[r * sg.cols + i] - exp((double)y[r * sg.cols + i]) * gsum);
        }
    } else if (sg.kind == PMT_PHI_ORTHOGONAL) {
        const int nn = sg.rows, N = 2 * nn;
        double* S = sh.m[0];
        double* G = sh.m[1];  // G_Q = base^T @ G_W (n x n), staged before the block matrix is built
        for (int idx = tid; idx < nn * nn; idx += PHI_THREADS) {
            const int i = idx / nn, j = idx - i * nn;
            double acc = 0.0;
            if (sg.base != nullptr) {
                for (int k = 0; k < nn; ++k) acc += (double)sg.base[k * sg.base_rs + i * sg.base_cs] * (double)gy[k * nn + j];
            } else {
                acc = (double)gy[idx];
            }
            G[idx] = acc;
        }
        __syncthreads();
        for (int idx = tid; idx < N * N; idx += PHI_THREADS) {  // [[A^T, G], [0, A^T]],  A^T = -A
            const int i = idx / N, j = idx - i * N;
            const int bi = i % nn, bj = j % nn;
            double v = 0.0;
            if ((i < nn) == (j < nn)) {
                v = bi > bj ? -(double)x[bi * nn + bj] : (bi < bj ? (double)x[bj * nn + bi] : 0.0);
            } else if (i < nn) {
                v = G[bi * nn + bj];
            }
            S[idx] = v;
        }
        __syncthreads();
        double* B = sh.m[1];
        expm_inplace(S, B, sh.m[2], sh.m[3], N, sh.red);
        for (int idx = tid; idx < nn * nn; idx += PHI_THREADS) {  // G_X = tril(G_A - G_A^T), G_A = S[:n, n:]
            const int i = idx / nn, j = idx - i * nn;
            if (i > j) gx[idx] += (float)(S[i * N + nn + j] - S[j * N + nn + i]);
        }
    }
}

static int phi_check(const PmtPhiProgram* p) {
    if (!p || p->n_segs < 0 || p->n_segs > PMT_MAX_PHI_SEGS) return PMT_E_INVALID;
    for (int i = 0; i < p->n_segs; ++i) {
        const PmtPhiSeg* s = &p->seg[i];
        if (s->kind < PMT_PHI_EXP || s->kind > PMT_PHI_ORTHOGONAL || s->rows < 1 || s->cols < 1 || s->theta_off < 0 || s->phi_off < 0)
            return PMT_E_INVALID;
        if (s->kind == PMT_PHI_ORTHOGONAL && (s->rows != s->cols || s->rows > PMT_MAX_ORTHO_DIM)) return PMT_E_UNSUPPORTED;
    }
    return PMT_OK;
}

extern "C" int pmt_phi_forward(const PmtPhiProgram* prog, const float* theta, float* phi, void* stream) {
    const int rc = phi_check(prog);
    if (rc) return rc;
    if (!theta || !phi) return PMT_E_INVALID;
    if (prog->n_segs == 0) return PMT_OK;
    hipLaunchKernelGGL(pmt_phi_forward_kernel, dim3(prog->n_segs), dim3(PHI_THREADS), 0, reinterpret_cast<hipStream_t>(stream),
                       *prog, theta, phi);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

extern "C" int pmt_phi_backward(const PmtPhiProgram* prog, const float* theta, const float* phi, const float* grad_phi,
                                float* grad_theta, void* stream) {
    const int rc = phi_check(prog);
    if (rc) return rc;
    if (!theta || !phi || !grad_phi || !grad_theta) return PMT_E_INVALID;
    if (prog->n_segs == 0) return PMT_OK;
    hipLaunchKernelGGL(pmt_phi_backward_kernel, dim3(prog->n_segs), dim3(PHI_THREADS), 0, reinterpret_cast<hipStream_t>(stream),
                       *prog, theta, phi, grad_phi, grad_theta);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

// Small boundary entry points that are NOT on the IW-ELBO hot path but belong to the reference's API surface:
//   iwvi_gaussian_var_exp  gpflow Gaussian.variational_expectations as a callable (models.py:66,134)
//   iwvi_unwhiten          the second back-substitution of the unwhitened case (temp_workaround.py:63-65),
//                          applied once to the operands instead of per sample
// Both are HBM/latency-bound elementwise or M^3-sized float64 work; no MFMA shaping.
#include "iwvi_common.h"

namespace iwvi {

__global__ __launch_bounds__(256) void k_gauss_var_exp(const float* __restrict__ Fmu, const float* __restrict__ Fvar,
                                                       const float* __restrict__ Y, float inv_var, float cst,
                                                       long long n, int Dy, long long row_div, long long row_mod,
                                                       float* __restrict__ out) {
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long t = idx / Dy;
        const int d = (int)(idx - t * Dy);
        const float y = Y[((t / row_div) % row_mod) * Dy + d];
        const float e = y - Fmu[idx];
        out[idx] = cst - 0.5f * (e * e + Fvar[idx]) * inv_var;
    }
}

// out[b][i, j] = sum_{k >= j} Linv[i, k] * Bop_b[k, j]  (float64 accumulate, float32 out), for i >= j; 0 above the diagonal.
//   batch b < R: Bop = tril(q_sqrt[b]) [M, M] -> q_sqrt_w[b] [M, M];  batch R: Bop = f [M, R] (all columns) -> f_w [M, R]
// 16x16 output tile per workgroup, operands through LDS.
__global__ __launch_bounds__(256) void k_unwhiten(const double* __restrict__ Linv, int Mp, int M, int R, int nq,
                                                  const float* __restrict__ f, const float* __restrict__ q_sqrt,
                                                  float* __restrict__ f_w, float* __restrict__ q_w) {
    __shared__ double As[16][17], Bs[16][17];
    const int b = blockIdx.z;
    const bool is_f = (b == nq);                  // nq = R, or 0 when there is no q_sqrt
    const int ncol = is_f ? R : M;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    if (j0 >= ncol) return;
    const int i = i0 + ty, j = j0 + tx;
    if (!is_f && j0 > i0 + 15) {                 // tile strictly above the diagonal
        if (i < M && j < M) q_w[((size_t)b * M + i) * M + j] = 0.f;
        return;
    }
    double s = 0.0;
    const int kend = min(M, i0 + 16);            // Linv is lower triangular: k <= i
    const int kbeg = is_f ? 0 : (j0 / 16) * 16;  // tril(q_sqrt): k >= j
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        As[ty][tx] = (i < M && k0 + tx < M && k0 + tx <= i) ? Linv[(size_t)i * Mp + k0 + tx] : 0.0;
        double v = 0.0;
        const int k = k0 + ty;
        if (k < M && j < ncol) {
            if (is_f) v = (double)f[(size_t)k * R + j];
            else if (k >= j) v = (double)q_sqrt[((size_t)b * M + k) * M + j];
        }
        Bs[ty][tx] = v;
        __syncthreads();
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) s = fma(As[ty][k2], Bs[k2][tx], s);
        __syncthreads();
    }
    if (i >= M || j >= ncol) return;
    if (is_f) f_w[(size_t)i * R + j] = (float)s;
    else q_w[((size_t)b * M + i) * M + j] = (j <= i) ? (float)s : 0.f;
}

}  // namespace iwvi

using namespace iwvi;

extern "C" int iwvi_gaussian_var_exp(const float* Fmu, const float* Fvar, const float* Y, float lik_variance,
                                     int64_t T, int Dy, int64_t row_div, int64_t row_mod, float* out, void* stream_) {
    if (T < 0 || Dy <= 0 || row_div <= 0 || row_mod <= 0 || !(lik_variance > 0.f)) {
        set_error("iwvi_gaussian_var_exp: bad size / variance"); return IWVI_ERR_ARG;
    }
    if (T == 0) return IWVI_OK;
    if (!Fmu || !Fvar || !Y || !out) { set_error("iwvi_gaussian_var_exp: null pointer"); return IWVI_ERR_ARG; }
    const long long n = (long long)T * Dy;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    const float cst = (float)(-0.5 * 1.8378770664093453 - 0.5 * log((double)lik_variance));
    hipLaunchKernelGGL(k_gauss_var_exp, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, Fmu, Fvar, Y,
                       1.f / lik_variance, cst, n, Dy, (long long)row_div, (long long)row_mod, out);
    return check_launch("iwvi_gaussian_var_exp");
}

extern "C" int iwvi_unwhiten(const void* state, int M, int R, const float* f, const float* q_sqrt,
                             float* f_w, float* q_sqrt_w, void* stream_) {
    if (M <= 0 || M > IWVI_MAX_M || R <= 0 || R > IWVI_MAX_R) { set_error("iwvi_unwhiten: bad size"); return IWVI_ERR_ARG; }
    if (!state || !f || !f_w || (q_sqrt && !q_sqrt_w)) { set_error("iwvi_unwhiten: null pointer"); return IWVI_ERR_ARG; }
    const StateLayout s = state_layout(M, R);
    const double* Linv = reinterpret_cast<const double*>(static_cast<const char*>(state) + s.off_Linv);
    const int nt = (M + 15) / 16;
    // batches 0..R-1 = q_sqrt rows (skipped when q_sqrt == NULL by launching only batch R)
    if (q_sqrt) {
        hipLaunchKernelGGL(k_unwhiten, dim3(nt, nt, R + 1), dim3(256), 0, (hipStream_t)stream_, Linv, s.Mp, M, R, R,
                           f, q_sqrt, f_w, q_sqrt_w);
    } else {
        hipLaunchKernelGGL(k_unwhiten, dim3(nt, nt, 1), dim3(256), 0, (hipStream_t)stream_, Linv, s.Mp, M, R, 0,
                           f, (const float*)nullptr, f_w, (float*)nullptr);
    }
    return check_launch("iwvi_unwhiten");
}

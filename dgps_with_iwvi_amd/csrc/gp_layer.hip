// Single-layer entry points of the C-ABI on top of the fused forward kernel (dgp_forward.hip), and the
// full-covariance-over-samples variant (temp_workaround.py:45,56,83 with full_cov=True).
#include "iwvi_common.h"

namespace iwvi {

using f32x4 = __attribute__((ext_vector_type(4))) float;

int dgp_forward_impl(const iwvi_layer_desc* layers, int n_layers, const float* X, int Dx, const float* XY, int XYdim,
                     const float* Y, int Dy, int64_t T, int64_t row_div, int64_t row_mod, float lik_variance,
                     uint64_t seed, uint64_t* rng_state, float* out_logw, const iwvi_elbo_desc* elbo, hipStream_t stream);

__device__ __forceinline__ float kern_value_f32(float r2, int type, float var) {
    if (type == IWVI_KERN_MATERN52) {
        const float s5 = 2.2360679774997896f;
        float r = sqrtf(r2 + 1e-12f);
        return var * (1.0f + s5 * r + (5.0f / 3.0f) * r * r) * __expf(-s5 * r);
    }
    return var * __expf(-0.5f * r2);
}

// cov[s][r][i][j] = k(x_si, x_sj) - a_si . a_sj + u_rsi . u_rsj      (temp_workaround.py:45,56,83)
__global__ __launch_bounds__(256) void k_fullcov(const float* __restrict__ F, const float* __restrict__ invls,
                                                 const float* __restrict__ a, const float* __restrict__ u,
                                                 float* __restrict__ cov, long long S, int N, int D, int Mp,
                                                 int R, int kern_type, float variance) {
    const long long s = blockIdx.x;
    const int r = blockIdx.y;
    const long long T = S * N;
    const float* as = a + (size_t)s * N * Mp;
    const float* us = u + ((size_t)r * T + (size_t)s * N) * Mp;
    float* out = cov + ((size_t)s * R + r) * N * N;
    for (int idx = threadIdx.x; idx < N * N; idx += blockDim.x) {
        const int i = idx / N, jx = idx - i * N;
        if (jx > i) continue;
        float r2 = 0.f;
        for (int d = 0; d < D; ++d) {
            float df = (F[((size_t)s * N + i) * D + d] - F[((size_t)s * N + jx) * D + d]) * invls[d];
            r2 = fmaf(df, df, r2);
        }
        float acc = kern_value_f32(r2, kern_type, variance);
        const f32x4* ai = reinterpret_cast<const f32x4*>(as + (size_t)i * Mp);
        const f32x4* aj = reinterpret_cast<const f32x4*>(as + (size_t)jx * Mp);
        const f32x4* ui = reinterpret_cast<const f32x4*>(us + (size_t)i * Mp);
        const f32x4* uj = reinterpret_cast<const f32x4*>(us + (size_t)jx * Mp);
        float da = 0.f, du = 0.f;
        for (int m4 = 0; m4 < Mp / 4; ++m4) {
            f32x4 x = ai[m4], y = aj[m4], p = ui[m4], q = uj[m4];
            da = fmaf(x.x, y.x, da); da = fmaf(x.y, y.y, da); da = fmaf(x.z, y.z, da); da = fmaf(x.w, y.w, da);
            du = fmaf(p.x, q.x, du); du = fmaf(p.y, q.y, du); du = fmaf(p.z, q.z, du); du = fmaf(p.w, q.w, du);
        }
        acc = acc - da + du;
        out[(size_t)i * N + jx] = acc;
        out[(size_t)jx * N + i] = acc;
    }
}

static int layer_forward_impl(const void* state, int M, int D, int R, int P, int kern_type, float variance,
                              const float* F, const float* noise, const float* W,
                              int mf_type, const float* mf_A, const float* mf_b,
                              float* sample, float* mean, float* var, float* a_out, float* u_out,
                              int64_t T, int bcast_K, hipStream_t stream) {
    if (T <= 0) return IWVI_OK;                         // empty batch: nothing to do
    if (!state || !F) { set_error("iwvi_gp_layer_forward: null state or input"); return IWVI_ERR_ARG; }
    if (bcast_K < 1 || T % bcast_K != 0) { set_error("iwvi_gp_layer_forward: T=%lld is not a multiple of bcast_K=%d", (long long)T, bcast_K); return IWVI_ERR_ARG; }
    if (D <= 0 || D > IWVI_MAX_D) { set_error("iwvi_gp_layer_forward: size out of range (M=%d D=%d R=%d P=%d)", M, D, R, P); return IWVI_ERR_ARG; }
    iwvi_layer_desc d{};
    d.type = IWVI_LAYER_GP; d.state = state; d.M = M; d.D = D; d.R = R; d.P = P; d.kern_type = kern_type;
    d.mf_type = mf_type; d.variance = variance; d.W = W; d.mf_A = mf_A; d.mf_b = mf_b;
    d.noise = noise; d.sample = sample; d.mean = mean; d.var = var; d.a_out = a_out; d.u_out = u_out;
    d.zero_noise = 1;        // noise == NULL means z = 0 for this entry point (include/iwvi_hip.h)
    return dgp_forward_impl(&d, 1, F, D, nullptr, 0, nullptr, 0, T, bcast_K, T / bcast_K, 1.f, 0, nullptr, nullptr, nullptr, stream);
}

}  // namespace iwvi

using namespace iwvi;

extern "C" int iwvi_gp_layer_forward(const void* state, int M, int D, int R, int P, int kern_type, float variance,
                                     const float* F, const float* noise, const float* W,
                                     int mf_type, const float* mf_A, const float* mf_b,
                                     float* sample, float* mean, float* var, int64_t T, int bcast_K, void* stream) {
    return layer_forward_impl(state, M, D, R, P, kern_type, variance, F, noise, W, mf_type, mf_A, mf_b,
                              sample, mean, var, nullptr, nullptr, T, bcast_K, (hipStream_t)stream);
}

extern "C" size_t iwvi_gp_fullcov_ws_bytes(int64_t T, int M, int R) {
    if (T <= 0 || M <= 0 || R <= 0) return 0;
    size_t Mp = (size_t)round_up(M, 16);
    return sizeof(float) * (size_t)T * Mp * (size_t)(R + 1);
}

extern "C" int iwvi_gp_layer_fullcov(const void* state, int M, int D, int R, int kern_type, float variance,
                                     const float* F, int64_t S, int64_t N, float* mean, float* cov,
                                     void* ws, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (S <= 0 || N <= 0) return IWVI_OK;
    if (!state || !F || !cov || !ws) { set_error("iwvi_gp_layer_fullcov: null pointer"); return IWVI_ERR_ARG; }
    if (N > 4096 || S > 0x7fffffffLL) { set_error("iwvi_gp_layer_fullcov: N=%lld > 4096 or S too large", (long long)N); return IWVI_ERR_ARG; }
    const int64_t T = S * N;
    const size_t Mp = (size_t)round_up(M, 16);
    float* a = (float*)ws;
    float* u = a + (size_t)T * Mp;
    int rc = layer_forward_impl(state, M, D, R, R, kern_type, variance, F, nullptr, nullptr, IWVI_MF_ZERO,
                                nullptr, nullptr, nullptr, mean, nullptr, a, u, T, 1, stream);
    if (rc != IWVI_OK) return rc;
    StateLayout sl = state_layout(M, R);
    const float* invls = (const float*)((const char*)state + sl.off_cst);
    hipLaunchKernelGGL(k_fullcov, dim3((unsigned)S, (unsigned)R), dim3(256), 0, stream, F, invls,
                       (const float*)a, (const float*)u, cov, (long long)S, (int)N, D, (int)Mp, R, kern_type, variance);
    return check_launch("k_fullcov");
}

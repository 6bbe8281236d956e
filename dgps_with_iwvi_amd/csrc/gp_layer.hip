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

constexpr int FULLCOV_ROWS = 32;
// cov[s][r][i][j] = k(x_si, x_sj) - a_si . a_sj + u_rsi . u_rsj      (temp_workaround.py:45,56,83)
template <bool F64>
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
    // blockIdx.z = a strip of rows (FULLCOV_ROWS of them), so that a single large block (predict_f_full_cov on a
    // test set: S = 1, N in the thousands) still fills the chip
    const int i_lo = blockIdx.z * FULLCOV_ROWS, i_hi = min(N, i_lo + FULLCOV_ROWS);
    for (long long idx = (long long)i_lo * N + threadIdx.x; idx < (long long)i_hi * N; idx += blockDim.x) {
        const int i = (int)(idx / N), jx = (int)(idx - (long long)i * N);
        if (jx > i) continue;
        if constexpr (F64) {
            // float64 route (IWVI_LAYER_F64_STAGE1): the entry is a difference of O(sigma^2) terms -- formed and summed in float64 from the
            // float32 a (rounded behind a float64 solve) and u rows
            double r2 = 0.0;
            for (int d = 0; d < D; ++d) {
                const double df = ((double)F[((size_t)s * N + i) * D + d] - (double)F[((size_t)s * N + jx) * D + d]) * (double)invls[d];
                r2 = fma(df, df, r2);
            }
            double acc;
            if (kern_type == IWVI_KERN_MATERN52) {
                const double s5 = 2.23606797749978969641, r = sqrt(r2 + 1e-12);
                acc = (double)variance * (1.0 + s5 * r + (5.0 / 3.0) * r * r) * exp(-s5 * r);
            } else acc = (double)variance * exp(-0.5 * r2);
            const float* ai = as + (size_t)i * Mp; const float* aj = as + (size_t)jx * Mp;
            const float* ui = us + (size_t)i * Mp; const float* uj = us + (size_t)jx * Mp;
            double da = 0.0, du = 0.0;
            for (int m = 0; m < Mp; ++m) { da = fma((double)ai[m], (double)aj[m], da); du = fma((double)ui[m], (double)uj[m], du); }
            const float o = (float)(acc - da + du);
            out[(size_t)i * N + jx] = o;
            out[(size_t)jx * N + i] = o;
            continue;
        }
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

// sample[s, n, r] = mean[s, n, r] + (chol(cov[s, r] + jitter I) z[s, r])[n]      (temp_workaround.py:93-96)
// One workgroup per (s, r): the N x N block is factorised in LDS (float32, right-looking, column at a time -- on the
// IW path N is the number of importance samples); N > MVN_LDS_N works in a caller-provided global scratch instead
// (correct, latency-bound: only the 2-D predict path with a consumed inner-layer sample gets there).
// The block is a float32 difference k - a.a + u.u: for (near-)singular blocks -- X tiled over K gives rank-1 blocks,
// a "zero" inner layer gives pure rounding noise -- it is indefinite by rounding where the float64 reference's is barely
// positive.  So the factorisation is the rounding-tolerant PSD form: a pivot <= 1e-6 C_jj zeroes its column (no component
// along that direction) and every entry is clamped to |L_ij| <= sqrt(C_ii), the bound any PSD factor obeys, which keeps
// a tiny pivot from amplifying rounding noise.  On a well-conditioned block this is the plain Cholesky factor.
constexpr int MVN_LDS_N = 192;
__global__ __launch_bounds__(256) void k_mvn_sample(const float* __restrict__ mean, const float* __restrict__ cov,
                                                    const float* __restrict__ z, float* __restrict__ sample,
                                                    int N, int R, float jitter, float* __restrict__ scratch) {
    extern __shared__ float mvn_sm[];
    const long long s = blockIdx.x;
    const int r = blockIdx.y, tid = threadIdx.x;
    float* Lm = scratch ? scratch + ((size_t)s * R + r) * ((size_t)N * N + 2 * N) : mvn_sm;   // [N, N] lower triangle
    float* zz = Lm + (size_t)N * N;                                                            // [N]
    float* bound = zz + N;                                                                     // [N] sqrt(max(C_ii, 0))
    const float* C = cov + ((size_t)s * R + r) * N * N;
    for (long long idx = tid; idx < (long long)N * N; idx += 256) {
        const int i = (int)(idx / N), k = (int)(idx - (long long)i * N);
        Lm[idx] = (k <= i) ? C[idx] + (k == i ? jitter : 0.f) : 0.f;
    }
    for (int i = tid; i < N; i += 256) {
        zz[i] = z[((size_t)s * R + r) * N + i];
        bound[i] = sqrtf(fmaxf(C[(size_t)i * N + i] + jitter, 0.f));
    }
    for (int j = 0; j < N; ++j) {
        __syncthreads();
        const float d = Lm[(size_t)j * N + j];
        // a pivot below 1e-6 of its original diagonal entry is float32 cancellation noise (~16 ulp), not signal
        const bool live = d > 1e-6f * bound[j] * bound[j];
        const float inv = live ? rsqrtf(d) : 0.f;
        __syncthreads();
        for (int i = j + 1 + tid; i < N; i += 256) {
            const float b = bound[i];
            Lm[(size_t)i * N + j] = fminf(fmaxf(Lm[(size_t)i * N + j] * inv, -b), b);
        }
        if (tid == 0) Lm[(size_t)j * N + j] = live ? fminf(d * inv, bound[j]) : 0.f;
        __syncthreads();
        const long long n = N - j - 1;
        for (long long idx = tid; idx < n * n; idx += 256) {
            const int i = j + 1 + (int)(idx / n), k = j + 1 + (int)(idx % n);
            if (k <= i) Lm[(size_t)i * N + k] = fmaf(-Lm[(size_t)i * N + j], Lm[(size_t)k * N + j], Lm[(size_t)i * N + k]);
        }
    }
    __syncthreads();
    for (int i = tid; i < N; i += 256) {
        float acc = 0.f;
        for (int k = 0; k <= i; ++k) acc = fmaf(Lm[(size_t)i * N + k], zz[k], acc);
        sample[((size_t)s * N + i) * R + r] = mean[((size_t)s * N + i) * R + r] + acc;
    }
}

static int layer_forward_impl(const void* state, int M, int D, int R, int P, int kern_type, float variance,
                              const float* F, const float* noise, const float* W,
                              int mf_type, const float* mf_A, const float* mf_b,
                              float* sample, float* mean, float* var, float* a_out, float* u_out,
                              int64_t T, int bcast_K, int layer_flags, hipStream_t stream) {
    if (T <= 0) return IWVI_OK;                         // empty batch: nothing to do
    if (!state || !F) { set_error("iwvi_gp_layer_forward: null state or input"); return IWVI_ERR_ARG; }
    if (bcast_K < 1 || T % bcast_K != 0) { set_error("iwvi_gp_layer_forward: T=%lld is not a multiple of bcast_K=%d", (long long)T, bcast_K); return IWVI_ERR_ARG; }
    if (D <= 0 || D > IWVI_MAX_D) { set_error("iwvi_gp_layer_forward: size out of range (M=%d D=%d R=%d P=%d)", M, D, R, P); return IWVI_ERR_ARG; }
    iwvi_layer_desc d{};
    d.type = IWVI_LAYER_GP; d.state = state; d.M = M; d.D = D; d.R = R; d.P = P; d.kern_type = kern_type;
    d.mf_type = mf_type; d.variance = variance; d.W = W; d.mf_A = mf_A; d.mf_b = mf_b;
    d.noise = noise; d.sample = sample; d.mean = mean; d.var = var; d.a_out = a_out; d.u_out = u_out;
    d.zero_noise = 1;        // noise == NULL means z = 0 for this entry point (include/iwvi_hip.h)
    d.flags = layer_flags;
    return dgp_forward_impl(&d, 1, F, D, nullptr, 0, nullptr, 0, T, bcast_K, T / bcast_K, 1.f, 0, nullptr, nullptr, nullptr, stream);
}

}  // namespace iwvi

using namespace iwvi;

extern "C" int iwvi_gp_layer_forward_ex(const void* state, int M, int D, int R, int P, int kern_type, float variance,
                                        const float* F, const float* noise, const float* W,
                                        int mf_type, const float* mf_A, const float* mf_b,
                                        float* sample, float* mean, float* var, int64_t T, int bcast_K, int layer_flags, void* stream) {
    return layer_forward_impl(state, M, D, R, P, kern_type, variance, F, noise, W, mf_type, mf_A, mf_b,
                              sample, mean, var, nullptr, nullptr, T, bcast_K, layer_flags, (hipStream_t)stream);
}
extern "C" int iwvi_gp_layer_forward(const void* state, int M, int D, int R, int P, int kern_type, float variance,
                                     const float* F, const float* noise, const float* W,
                                     int mf_type, const float* mf_A, const float* mf_b,
                                     float* sample, float* mean, float* var, int64_t T, int bcast_K, void* stream) {
    return iwvi_gp_layer_forward_ex(state, M, D, R, P, kern_type, variance, F, noise, W, mf_type, mf_A, mf_b, sample, mean, var, T, bcast_K, 0, stream);
}

extern "C" size_t iwvi_gp_fullcov_ws_bytes(int64_t T, int M, int R) {
    if (T <= 0 || M <= 0 || R <= 0) return 0;
    size_t Mp = (size_t)round_up(M, 16);
    return sizeof(float) * (size_t)T * Mp * (size_t)(R + 1);
}

extern "C" int iwvi_gp_layer_fullcov(const void* state, int M, int D, int R, int kern_type, float variance,
                                     const float* F, int64_t S, int64_t N,
                                     int mf_type, const float* mf_A, const float* mf_b,
                                     float* mean, float* cov, void* ws, void* stream_) {
    return iwvi_gp_layer_fullcov_ex(state, M, D, R, kern_type, variance, F, S, N, mf_type, mf_A, mf_b, mean, cov, ws, 0, stream_);
}
extern "C" int iwvi_gp_layer_fullcov_ex(const void* state, int M, int D, int R, int kern_type, float variance,
                                        const float* F, int64_t S, int64_t N,
                                        int mf_type, const float* mf_A, const float* mf_b,
                                        float* mean, float* cov, void* ws, int layer_flags, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (S <= 0 || N <= 0) return IWVI_OK;
    if (!state || !F || !cov || !ws) { set_error("iwvi_gp_layer_fullcov: null pointer"); return IWVI_ERR_ARG; }
    if (N > 46340 || S > 0x7fffffffLL) { set_error("iwvi_gp_layer_fullcov: N=%lld > 46340 or S too large", (long long)N); return IWVI_ERR_ARG; }
    const int64_t T = S * N;
    const size_t Mp = (size_t)round_up(M, 16);
    float* a = (float*)ws;
    float* u = a + (size_t)T * Mp;
    int rc = layer_forward_impl(state, M, D, R, R, kern_type, variance, F, nullptr, nullptr, mf_type,
                                mf_A, mf_b, nullptr, mean, nullptr, a, u, T, 1, layer_flags, stream);
    if (rc != IWVI_OK) return rc;
    StateLayout sl = state_layout(M, R);
    const float* invls = (const float*)((const char*)state + sl.off_cst);
    const dim3 grid((unsigned)S, (unsigned)R, (unsigned)((N + FULLCOV_ROWS - 1) / FULLCOV_ROWS));
    if (layer_flags & IWVI_LAYER_F64_STAGE1)
        hipLaunchKernelGGL(k_fullcov<true>, grid, dim3(256), 0, stream, F, invls,
                           (const float*)a, (const float*)u, cov, (long long)S, (int)N, D, (int)Mp, R, kern_type, variance);
    else
        hipLaunchKernelGGL(k_fullcov<false>, grid, dim3(256), 0, stream, F, invls,
                           (const float*)a, (const float*)u, cov, (long long)S, (int)N, D, (int)Mp, R, kern_type, variance);
    return check_launch("k_fullcov");
}

extern "C" size_t iwvi_mvn_sample_ws_bytes(int64_t S, int N, int R) {
    if (S <= 0 || N <= MVN_LDS_N || R <= 0) return 0;
    return sizeof(float) * (size_t)S * R * ((size_t)N * N + 2 * (size_t)N);
}

extern "C" int iwvi_mvn_sample(const float* mean, const float* cov, const float* z, float* sample,
                               int64_t S, int N, int R, float jitter, void* ws, void* stream_) {
    if (S <= 0 || N <= 0) return IWVI_OK;
    if (!mean || !cov || !z || !sample) { set_error("iwvi_mvn_sample: null pointer"); return IWVI_ERR_ARG; }
    if (N > 46340 || R <= 0 || R > 65535 || S > 0x7fffffffLL) { set_error("iwvi_mvn_sample: bad size (S=%lld N=%d R=%d)", (long long)S, N, R); return IWVI_ERR_ARG; }
    if (N > MVN_LDS_N && !ws) { set_error("iwvi_mvn_sample: N=%d > %d needs the scratch of iwvi_mvn_sample_ws_bytes", N, MVN_LDS_N); return IWVI_ERR_ARG; }
    const size_t lds = N > MVN_LDS_N ? 0 : sizeof(float) * ((size_t)N * N + 2 * (size_t)N);
    static bool attr_set = false;          // not a stream operation: once, outside any capture's steady state
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_mvn_sample, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_error("iwvi_mvn_sample: hipFuncSetAttribute: %s", hipGetErrorString(e)); return IWVI_ERR_LAUNCH; }
        attr_set = true;
    }
    hipLaunchKernelGGL(k_mvn_sample, dim3((unsigned)S, (unsigned)R), dim3(256), lds, (hipStream_t)stream_,
                       mean, cov, z, sample, N, R, jitter, N > MVN_LDS_N ? (float*)ws : (float*)nullptr);
    return check_launch("k_mvn_sample");
}

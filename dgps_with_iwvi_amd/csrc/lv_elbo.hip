// The IW-ELBO reduction, the K-shard merge, the counter-based normal generator, and the single-layer
// LatentVariableLayer entry point (on the fused kernel of dgp_forward.hip).
// Reference: layers.py:72-105,137-152; models.py:133-150.
#include "iwvi_common.h"

namespace iwvi {

int dgp_forward_impl(const iwvi_layer_desc* layers, int n_layers, const float* X, int Dx, const float* XY, int XYdim,
                     const float* Y, int Dy, int64_t T, int64_t row_div, int64_t row_mod, float lik_variance,
                     uint64_t seed, uint64_t* rng_state, float* out_logw, const iwvi_elbo_desc* elbo, hipStream_t stream);

// ------------------------------------------------------------------------------------------
// IW-ELBO reduction: one wave per data point.
// ------------------------------------------------------------------------------------------
constexpr int MAX_GLOB = 16;
struct ReduceArgs {
    const float* fmean; const float* fvar; const float* Y; const float* logw;
    const float* kl[IWVI_MAX_KL]; int kl_dims[IWVI_MAX_KL]; int n_kl;
    long long B, stride_b, stride_k; int K, Dy, K_total, mode_vi;
    float lik_variance;
    const float* lik_var_dev;        // optional device scalar read instead of lik_variance (a trained likelihood variance; iwvi_iw_elbo_reduce_dev)
    float* ms; float* logp;
    // fused final sum (last-arriving workgroup): out_elbo = sum(logp) * scale - sum(global KLs)
    double* elbo; unsigned long long* ticket; double scale;
    const double* klg[MAX_GLOB]; int klg_n[MAX_GLOB]; int n_glob;
};

template <int SEG>
__device__ __forceinline__ float seg_max(float v) {
#pragma unroll
    for (int o = SEG / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
template <int SEG>
__device__ __forceinline__ float seg_sum(float v) {
#pragma unroll
    for (int o = SEG / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

constexpr int ELBO_THREADS = 256;
constexpr int ELBO_PTS = 64;            // points per workgroup

// SEG lanes (a power of two <= 64, >= min(K, 64)) cooperate on one data point: Gaussian variational
// expectations (models.py:134), minus local regularisers (:140-142), log-sum-exp over K (:148).  The last
// workgroup to finish (agent-scope release / ticket / acquire, cdna guide G16) adds the points up in a
// fixed order, so the ELBO (:150) is bit-reproducible and needs no second launch.
template <int SEG>
__global__ __launch_bounds__(ELBO_THREADS) void k_elbo(ReduceArgs g) {
    __shared__ double red[ELBO_THREADS];
    __shared__ int is_last;
    const int tid = threadIdx.x, sl = tid % SEG, sg = tid / SEG;
    constexpr int PPP = ELBO_THREADS / SEG;          // points per pass
    const int K = g.K, Dy = g.Dy;
    const float likv = g.lik_var_dev ? *g.lik_var_dev : g.lik_variance;
    const float c0 = -0.5f * 1.8378770664093453f - 0.5f * logf(likv);
    const float inv2s = 0.5f / likv;
    for (int pp = 0; pp < ELBO_PTS; pp += PPP) {
        const long long b = (long long)blockIdx.x * ELBO_PTS + pp + sg;
        const bool live = b < g.B;                    // uniform within a segment
        float m = -INFINITY, ssum = 0.f, lsum = 0.f;
        for (int k0 = 0; k0 < K; k0 += SEG) {
            const int k = k0 + sl;
            float L = -INFINITY;
            if (live && k < K) {
                const long long t = b * g.stride_b + k * g.stride_k;
                float acc = 0.f;
                if (g.logw) acc = g.logw[t];                   // models.py:134-142 done by the fused forward
                else {
                    for (int d = 0; d < Dy; ++d) {
                        const float df = g.Y[b * Dy + d] - g.fmean[t * Dy + d];
                        acc += c0 - (df * df + g.fvar[t * Dy + d]) * inv2s;
                    }
                    for (int i = 0; i < g.n_kl; ++i)
                        for (int d = 0; d < g.kl_dims[i]; ++d) acc -= g.kl[i][t * g.kl_dims[i] + d];
                }
                L = acc;
            }
            if (g.mode_vi) { lsum += seg_sum<SEG>((live && k < K) ? L : 0.f); continue; }
            const float cm = seg_max<SEG>(L);
            const float nm = fmaxf(m, cm);
            const float e = (live && k < K) ? __expf(L - nm) : 0.f;
            const float cs = seg_sum<SEG>(e);
            ssum = (m == -INFINITY ? 0.f : ssum * __expf(m - nm)) + cs;
            m = nm;
        }
        if (live && sl == 0) {
            if (g.mode_vi) {
                if (g.logp) g.logp[b] = lsum / (float)K;                               // models.py:84
            } else {
                if (g.ms) { g.ms[2 * b] = m; g.ms[2 * b + 1] = ssum; }
                if (g.logp) g.logp[b] = m + logf(ssum) - logf((float)g.K_total);      // models.py:148
            }
        }
    }
    if (!g.elbo) return;
    // ---- publish this workgroup's logp, draw a ticket, the last arriver sums everything ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t = __hip_atomic_fetch_add(g.ticket, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (unsigned long long)gridDim.x - 1);
        if (last) {
            __hip_atomic_store(g.ticket, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        is_last = last;
    }
    __syncthreads();
    if (!is_last) return;
    double acc = 0.0;
    for (long long b = tid; b < g.B; b += ELBO_THREADS) acc += (double)g.logp[b];
    red[tid] = acc;
    __syncthreads();
    for (int s2 = ELBO_THREADS / 2; s2 > 0; s2 >>= 1) {
        if (tid < s2) red[tid] += red[tid + s2];
        __syncthreads();
    }
    if (tid == 0) {
        double kl = 0.0;
        for (int i = 0; i < g.n_glob; ++i)
            for (int c = 0; c < g.klg_n[i]; ++c) kl += g.klg[i][c];
        *g.elbo = red[0] * g.scale - kl;                                              // models.py:150
    }
}

struct FinalArgs {
    const float* logp; const float* ms_all; int G;
    long long B; long long rstride;      // points between two ranks' partials of one evaluation (B, or n_steps * B when batched)
    int K_total; double scale;
    const double* klg[MAX_GLOB]; int klg_n[MAX_GLOB]; int n_glob;
    float* logp_out; double* elbo;
};

// optional merge of G gathered (max, sumexp) partials per point, then the deterministic final sum
__global__ __launch_bounds__(1024) void k_elbo_final(FinalArgs g) {
    __shared__ double red[1024];
    double acc = 0.0;
    // block e: evaluation e of a batch laid out [rank][evaluation][B][2] (a single evaluation: one block, rstride = B)
    const float* ms = g.ms_all ? g.ms_all + (size_t)blockIdx.x * g.B * 2 : nullptr;
    float* lpo = g.logp_out ? g.logp_out + (size_t)blockIdx.x * g.B : nullptr;
    for (long long b = threadIdx.x; b < g.B; b += blockDim.x) {
        float lp;
        if (ms) {
            float m = -INFINITY;
            for (int r = 0; r < g.G; ++r) m = fmaxf(m, ms[((size_t)r * g.rstride + b) * 2]);
            float s = 0.f;
            for (int r = 0; r < g.G; ++r) {
                const float* p = ms + ((size_t)r * g.rstride + b) * 2;
                s += p[1] * __expf(p[0] - m);
            }
            lp = m + logf(s) - logf((float)g.K_total);
            if (lpo) lpo[b] = lp;
        } else {
            lp = g.logp[b];
        }
        acc += (double)lp;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && g.elbo) {
        double kl = 0.0;
        for (int i = 0; i < g.n_glob; ++i)
            for (int c = 0; c < g.klg_n[i]; ++c) kl += g.klg[i][c];
        g.elbo[blockIdx.x] = red[0] * g.scale - kl;                               // models.py:150
    }
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller.  Element i of the output uses counter (offset + i/4, 0, 0, 0), key
// (seed_lo, seed_hi), word i%4: words (0,1) -> (r cos, r sin), words (2,3) likewise.
// ------------------------------------------------------------------------------------------
template <bool DEVCTR>
__global__ void k_fill_normal(float* out, long long n, uint64_t seed, uint64_t offset, unsigned long long* state) {
    const long long nq = (n + 3) / 4;
    if (DEVCTR) offset = state[0];            // every block reads the counter before any block can bump it
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < nq;
         q += (long long)gridDim.x * blockDim.x) {
        uint64_t ctr = offset + (uint64_t)q;
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        float v[4];
        box_muller4(c, v);
        for (int e = 0; e < 4; ++e) if (4 * q + e < n) out[4 * q + e] = v[e];
    }
    if (DEVCTR) {
        // the last block of THIS launch to get here advances the counter for the next launch/replay
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t = atomicAdd(&state[1], 1ULL);
            if ((t + 1) % gridDim.x == 0) atomicAdd(&state[0], (unsigned long long)nq);
        }
    }
}

}  // namespace iwvi

using namespace iwvi;

extern "C" int iwvi_lv_layer_forward(const float* F, const float* XY, const float* noise,
                                     const float* const* enc_W, const float* const* enc_b,
                                     const int32_t* dims, int n_enc, int D, int Lw, int sampled_kl,
                                     float* sample, float* mean, float* cov, float* kl,
                                     int64_t T, void* stream_) {
    return iwvi_lv_layer_forward_act(F, XY, noise, enc_W, enc_b, dims, n_enc, IWVI_ACT_TANH, D, Lw, sampled_kl, sample, mean, cov, kl, T, stream_);
}

extern "C" int iwvi_lv_layer_forward_act(const float* F, const float* XY, const float* noise,
                                         const float* const* enc_W, const float* const* enc_b,
                                         const int32_t* dims, int n_enc, int act, int D, int Lw, int sampled_kl,
                                         float* sample, float* mean, float* cov, float* kl,
                                         int64_t T, void* stream_) {
    if (T <= 0) return IWVI_OK;
    if (!F) { set_error("iwvi_lv_layer_forward: null input"); return IWVI_ERR_ARG; }
    if (D <= 0 || D > IWVI_MAX_D || Lw <= 0) { set_error("iwvi_lv_layer_forward: bad D=%d (1..32) or latent_dim=%d", D, Lw); return IWVI_ERR_ARG; }
    iwvi_layer_desc d{};
    d.type = IWVI_LAYER_LV; d.D = D; d.latent_dim = Lw; d.sampled_kl = sampled_kl;
    if (XY) { d.enc_W = enc_W; d.enc_b = enc_b; d.enc_dims = dims; d.n_enc = n_enc; d.enc_act = act;
              if (!enc_W || !dims) { set_error("iwvi_lv_layer_forward: encoder inputs without an encoder"); return IWVI_ERR_ARG; } }
    d.noise = noise; d.zero_noise = 1; d.sample = sample; d.mean = mean; d.var = cov; d.kl_local = kl;
    return dgp_forward_impl(&d, 1, F, D, XY, XY ? dims[0] : 0, nullptr, 0, T, 1, T, 1.f, 0, nullptr, nullptr, nullptr,
                            (hipStream_t)stream_);
}

template <typename ArgsT>
static int fill_globals(ArgsT& f, const double* const* klg, const int32_t* counts, int n_glob) {
    if (n_glob < 0 || n_glob > MAX_GLOB) { set_error("too many global KL terms (%d > %d)", n_glob, MAX_GLOB); return IWVI_ERR_ARG; }
    for (int i = 0; i < n_glob; ++i) {
        if (!klg || !klg[i]) { set_error("null global KL pointer %d", i); return IWVI_ERR_ARG; }
        f.klg[i] = klg[i];
        f.klg_n[i] = counts ? counts[i] : 1;
        if (f.klg_n[i] <= 0 || f.klg_n[i] > IWVI_MAX_R) { set_error("bad global KL count %d", f.klg_n[i]); return IWVI_ERR_ARG; }
    }
    f.n_glob = n_glob;
    return IWVI_OK;
}

template <int SEG>
static int launch_elbo(const ReduceArgs& g, hipStream_t stream) {
    long long blocks = (g.B + ELBO_PTS - 1) / ELBO_PTS;
    hipLaunchKernelGGL(k_elbo<SEG>, dim3((unsigned)blocks), dim3(ELBO_THREADS), 0, stream, g);
    return check_launch("k_elbo");
}

extern "C" int iwvi_iw_elbo_reduce(const float* fmean, const float* fvar, const float* Y,
                                   float lik_variance, int64_t B, int K, int Dy,
                                   int64_t stride_b, int64_t stride_k,
                                   const float* const* kl_local, const int32_t* kl_dims, int n_kl,
                                   const double* const* kl_global, const int32_t* kl_global_counts, int n_glob,
                                   double scale, int K_total, int mode_vi,
                                   float* out_ms, float* out_logp, double* out_elbo, uint64_t* ticket,
                                   void* stream_) {
    return iwvi_iw_elbo_reduce_dev(fmean, fvar, Y, lik_variance, nullptr, B, K, Dy, stride_b, stride_k, kl_local, kl_dims, n_kl,
                                   kl_global, kl_global_counts, n_glob, scale, K_total, mode_vi, out_ms, out_logp, out_elbo, ticket, stream_);
}

extern "C" int iwvi_iw_elbo_reduce_dev(const float* fmean, const float* fvar, const float* Y,
                                       float lik_variance, const float* lik_variance_dev, int64_t B, int K, int Dy,
                                       int64_t stride_b, int64_t stride_k,
                                       const float* const* kl_local, const int32_t* kl_dims, int n_kl,
                                       const double* const* kl_global, const int32_t* kl_global_counts, int n_glob,
                                       double scale, int K_total, int mode_vi,
                                       float* out_ms, float* out_logp, double* out_elbo, uint64_t* ticket,
                                       void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!fmean || !fvar || !Y) { set_error("iwvi_iw_elbo_reduce: null input"); return IWVI_ERR_ARG; }
    if (B <= 0) { set_error("iwvi_iw_elbo_reduce: empty minibatch"); return IWVI_ERR_ARG; }
    if (K <= 0 || Dy <= 0 || !(lik_variance > 0.f)) { set_error("iwvi_iw_elbo_reduce: bad K=%d, Dy=%d or likelihood variance", K, Dy); return IWVI_ERR_ARG; }
    if (n_kl < 0 || n_kl > IWVI_MAX_KL) { set_error("iwvi_iw_elbo_reduce: %d local regularisers (max %d)", n_kl, IWVI_MAX_KL); return IWVI_ERR_ARG; }
    if (out_elbo && (!out_logp || !ticket)) { set_error("iwvi_iw_elbo_reduce: out_elbo needs out_logp (scratch) and a zero-initialised ticket word"); return IWVI_ERR_ARG; }
    ReduceArgs g{};
    g.fmean = fmean; g.fvar = fvar; g.Y = Y; g.n_kl = n_kl;
    for (int i = 0; i < n_kl; ++i) {
        if (!kl_local || !kl_local[i] || !kl_dims || kl_dims[i] <= 0) { set_error("iwvi_iw_elbo_reduce: bad local regulariser %d", i); return IWVI_ERR_ARG; }
        g.kl[i] = kl_local[i]; g.kl_dims[i] = kl_dims[i];
    }
    g.stride_b = stride_b; g.stride_k = stride_k;
    g.B = B; g.K = K; g.Dy = Dy; g.K_total = K_total > 0 ? K_total : K; g.mode_vi = mode_vi;
    g.lik_variance = lik_variance; g.lik_var_dev = lik_variance_dev; g.ms = out_ms; g.logp = out_logp;
    g.elbo = out_elbo; g.ticket = (unsigned long long*)ticket; g.scale = scale;
    int rc;
    if (out_elbo && (rc = fill_globals(g, kl_global, kl_global_counts, n_glob)) != IWVI_OK) return rc;
    if (K <= 4) return launch_elbo<4>(g, stream);
    if (K <= 8) return launch_elbo<8>(g, stream);
    if (K <= 16) return launch_elbo<16>(g, stream);
    if (K <= 32) return launch_elbo<32>(g, stream);
    return launch_elbo<64>(g, stream);
}

extern "C" int iwvi_logw_reduce(const float* logw, int64_t B, int K, int64_t stride_b, int64_t stride_k,
                                const double* const* kl_global, const int32_t* kl_global_counts, int n_glob,
                                double scale, int K_total, int mode_vi,
                                float* out_ms, float* out_logp, double* out_elbo, uint64_t* ticket, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!logw) { set_error("iwvi_logw_reduce: null input"); return IWVI_ERR_ARG; }
    if (B <= 0 || K <= 0) { set_error("iwvi_logw_reduce: empty minibatch or K=%d", K); return IWVI_ERR_ARG; }
    if (out_elbo && (!out_logp || !ticket)) { set_error("iwvi_logw_reduce: out_elbo needs out_logp (scratch) and a zero-initialised ticket word"); return IWVI_ERR_ARG; }
    ReduceArgs g{};
    g.logw = logw; g.stride_b = stride_b; g.stride_k = stride_k;
    g.B = B; g.K = K; g.Dy = 1; g.K_total = K_total > 0 ? K_total : K; g.mode_vi = mode_vi;
    g.lik_variance = 1.f; g.ms = out_ms; g.logp = out_logp;
    g.elbo = out_elbo; g.ticket = (unsigned long long*)ticket; g.scale = scale;
    int rc;
    if (out_elbo && (rc = fill_globals(g, kl_global, kl_global_counts, n_glob)) != IWVI_OK) return rc;
    if (K <= 4) return launch_elbo<4>(g, stream);
    if (K <= 8) return launch_elbo<8>(g, stream);
    if (K <= 16) return launch_elbo<16>(g, stream);
    if (K <= 32) return launch_elbo<32>(g, stream);
    return launch_elbo<64>(g, stream);
}

extern "C" int iwvi_lse_merge(const float* ms_all, int G, int64_t B, int K_total,
                              const double* const* kl_global, const int32_t* kl_global_counts, int n_glob,
                              double scale, float* out_logp, double* out_elbo, void* stream_) {
    if (!ms_all || G <= 0 || B <= 0 || K_total <= 0) { set_error("iwvi_lse_merge: bad argument"); return IWVI_ERR_ARG; }
    FinalArgs f{};
    f.ms_all = ms_all; f.G = G; f.B = B; f.rstride = B; f.K_total = K_total; f.scale = scale;
    f.logp_out = out_logp; f.elbo = out_elbo;
    int rc;
    if ((rc = fill_globals(f, kl_global, kl_global_counts, n_glob)) != IWVI_OK) return rc;
    hipLaunchKernelGGL(k_elbo_final, dim3(1), dim3(1024), 0, (hipStream_t)stream_, f);
    return check_launch("k_elbo_final(merge)");
}

extern "C" int iwvi_lse_merge_steps(const float* ms_all, int G, int n_steps, int64_t B, int K_total,
                                    const double* const* kl_global, const int32_t* kl_global_counts, int n_glob,
                                    double scale, float* out_logp, double* out_elbo, void* stream_) {
    if (!ms_all || !out_elbo || G <= 0 || n_steps <= 0 || B <= 0 || K_total <= 0) { set_error("iwvi_lse_merge_steps: bad argument"); return IWVI_ERR_ARG; }
    FinalArgs f{};
    f.ms_all = ms_all; f.G = G; f.B = B; f.rstride = (long long)n_steps * B; f.K_total = K_total; f.scale = scale;
    f.logp_out = out_logp; f.elbo = out_elbo;
    int rc;
    if ((rc = fill_globals(f, kl_global, kl_global_counts, n_glob)) != IWVI_OK) return rc;
    hipLaunchKernelGGL(k_elbo_final, dim3(n_steps), dim3(1024), 0, (hipStream_t)stream_, f);
    return check_launch("k_elbo_final(merge, batched)");
}

static int fill_normal_impl(float* out, int64_t n, uint64_t seed, uint64_t offset, unsigned long long* state,
                            hipStream_t stream) {
    if (n <= 0) return IWVI_OK;
    if (!out) { set_error("iwvi_fill_normal: null output"); return IWVI_ERR_ARG; }
    long long nq = (n + 3) / 4;
    long long blocks = (nq + 255) / 256; if (blocks > 4096) blocks = 4096;
    if (state) hipLaunchKernelGGL(k_fill_normal<true>, dim3((unsigned)blocks), dim3(256), 0, stream, out, (long long)n, seed, offset, state);
    else hipLaunchKernelGGL(k_fill_normal<false>, dim3((unsigned)blocks), dim3(256), 0, stream, out, (long long)n, seed, offset, state);
    return check_launch("k_fill_normal");
}

extern "C" int iwvi_fill_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream_) {
    return fill_normal_impl(out, n, seed, offset, nullptr, (hipStream_t)stream_);
}

extern "C" int iwvi_fill_normal_dev(float* out, int64_t n, uint64_t seed, uint64_t* state, void* stream_) {
    if (!state) { set_error("iwvi_fill_normal_dev: null state"); return IWVI_ERR_ARG; }
    return fill_normal_impl(out, n, seed, 0, (unsigned long long*)state, (hipStream_t)stream_);
}

// ------------------------------------------------------------------------------------------------------------
// Test log-likelihood of experiments/run_conditional_density_estimation.py:148-169, batched: per test point a
// Gaussian kernel-density estimate over its S predictive samples with Silverman's bandwidth (:158-162), the log
// density at the observed y, and the squared error of the sample mean (:165).  One wave per point, lanes over S.
// ------------------------------------------------------------------------------------------------------------
namespace iwvi {
__device__ __forceinline__ double wsum_d(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__global__ __launch_bounds__(256) void k_kde_loglik(const float* samples, long long sstride, long long nstride, const float* y,
                                                    long long N, int S, float* logp, float* sqerr, float* stats) {
    const int lane = threadIdx.x & 63;
    const long long n = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* s = samples + n * nstride;
    double sum = 0.0;
    for (int i = lane; i < S; i += 64) sum += (double)s[i * sstride];
    const double mean = wsum_d(sum) / S;
    double ss = 0.0;
    for (int i = lane; i < S; i += 64) { const double e = (double)s[i * sstride] - mean; ss += e * e; }
    const double sd = sqrt(wsum_d(ss) / S);                              // np.std: population standard deviation
    const double bw = 1.06 * sd * pow((double)S, -0.2);                  // Silverman (1986), :158
    const double yy = (double)y[n];
    float mx = -INFINITY;
    for (int i = lane; i < S; i += 64) { const double e = (yy - (double)s[i * sstride]) / bw; mx = fmaxf(mx, (float)(-0.5 * e * e)); }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    double se = 0.0;
    for (int i = lane; i < S; i += 64) { const double e = (yy - (double)s[i * sstride]) / bw; se += exp(-0.5 * e * e - (double)mx); }
    se = wsum_d(se);
    if (lane == 0) {
        if (logp) logp[n] = (float)((double)mx + log(se) - log((double)S * bw) - 0.9189385332046727);   // - log sqrt(2 pi)
        if (sqerr) sqerr[n] = (float)((mean - yy) * (mean - yy));
        if (stats) { stats[2 * n] = (float)mean; stats[2 * n + 1] = (float)sd; }
    }
}
}  // namespace iwvi

extern "C" int iwvi_kde_loglik(const float* samples, int64_t sample_stride, int64_t point_stride, const float* y,
                               int64_t N, int S, float* out_logp, float* out_sqerr, float* out_mean_std, void* stream_) {
    if (!samples || !y || N <= 0 || S <= 1 || sample_stride <= 0 || point_stride <= 0) { iwvi::set_error("iwvi_kde_loglik: bad argument"); return IWVI_ERR_ARG; }
    hipLaunchKernelGGL(iwvi::k_kde_loglik, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, samples,
                       (long long)sample_stride, (long long)point_stride, y, (long long)N, S, out_logp, out_sqerr, out_mean_std);
    return iwvi::check_launch("k_kde_loglik");
}

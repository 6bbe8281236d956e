// LatentVariableLayer forward (+ Encoder MLP), the IW-ELBO reduction, the K-shard merge and the
// counter-based normal generator.  Reference: layers.py:72-105,137-152; models.py:133-150.
#include "iwvi_common.h"

namespace iwvi {

// ------------------------------------------------------------------------------------------
// LatentVariableLayer: one thread per sample; activations live in LDS as [feature][thread] (conflict
// free), weights are wave-uniform so they come through the scalar cache.
// ------------------------------------------------------------------------------------------
constexpr int LV_THREADS = 128;
constexpr int LV_MAXDIM = 64;

struct LvArgs {
    const float* F; const float* XY; const float* noise;
    const float* W[IWVI_MAX_ENC]; const float* b[IWVI_MAX_ENC];
    int dims[IWVI_MAX_ENC + 1];
    int n_enc, D, Lw, sampled_kl, maxdim;
    float* sample; float* mean; float* cov; float* kl;
    long long T;
};

extern __shared__ __attribute__((aligned(16))) unsigned char lv_smem[];

__device__ __forceinline__ float softplus_f(float x) {
    return x > 20.f ? x : log1pf(expf(x));
}

__global__ __launch_bounds__(LV_THREADS) void k_lv_layer(LvArgs g) {
    const int tid = threadIdx.x;
    const long long t = (long long)blockIdx.x * LV_THREADS + tid;
    const bool live = t < g.T;
    float* act0 = reinterpret_cast<float*>(lv_smem);
    float* act1 = act0 + (size_t)g.maxdim * LV_THREADS;
    const int D = g.D, Lw = g.Lw;
    if (g.XY) {
        const int d0 = g.dims[0];
        for (int i = 0; i < d0; ++i) act0[i * LV_THREADS + tid] = live ? g.XY[t * d0 + i] : 0.f;
        float* in = act0; float* out = act1;
        for (int l = 0; l < g.n_enc; ++l) {
            const int din = g.dims[l], dout = g.dims[l + 1];
            const float* W = g.W[l]; const float* b = g.b[l];
            for (int o = 0; o < dout; ++o) {
                float acc = b ? b[o] : 0.f;
                for (int i = 0; i < din; ++i) acc = fmaf(in[i * LV_THREADS + tid], W[i * dout + o], acc);
                if (l < g.n_enc - 1) acc = tanhf(acc);                         // layers.py:143-144
                if (din == dout) acc += in[o * LV_THREADS + tid];              // layers.py:146-147
                out[o * LV_THREADS + tid] = acc;
            }
            float* tmp = in; in = out; out = tmp;
        }
        if (in != act0) for (int i = 0; i < 2 * Lw; ++i) act0[i * LV_THREADS + tid] = in[i * LV_THREADS + tid];
    }
    if (!live) return;
    const int Do = D + Lw;
    for (int d = 0; d < D; ++d) {
        float f = g.F[t * D + d];
        if (g.sample) g.sample[t * Do + d] = f;
        if (g.mean) g.mean[t * Do + d] = f;
        if (g.cov) g.cov[t * Do + d] = 0.f;
    }
    for (int l = 0; l < Lw; ++l) {
        float mu = 0.f, sg = 1.f;                                               // prior (layers.py:73-81)
        if (g.XY) { mu = act0[l * LV_THREADS + tid]; sg = softplus_f(act0[(Lw + l) * LV_THREADS + tid] - 3.f); }
        float z = g.noise ? g.noise[t * Lw + l] : 0.f;
        float w = fmaf(z, sg, mu);                                              // layers.py:86-87
        if (g.sample) g.sample[t * Do + D + l] = w;
        if (g.mean) g.mean[t * Do + D + l] = mu;
        if (g.cov) g.cov[t * Do + D + l] = sg * sg;
        if (g.kl) {
            float kl;
            if (g.sampled_kl) kl = -0.5f * z * z - logf(sg) + 0.5f * w * w;     // log q(W) - log p(W)
            else kl = 0.5f * (sg * sg + mu * mu - 1.f) - logf(sg);              // KL(N(mu,sg)||N(0,1))
            g.kl[t * Lw + l] = kl;
        }
    }
}

// ------------------------------------------------------------------------------------------
// IW-ELBO reduction: one wave per data point.
// ------------------------------------------------------------------------------------------
constexpr int MAX_GLOB = 16;
struct ReduceArgs {
    const float* fmean; const float* fvar; const float* Y;
    const float* kl[IWVI_MAX_KL]; int kl_dims[IWVI_MAX_KL]; int n_kl;
    long long B, stride_b, stride_k; int K, Dy, K_total, mode_vi;
    float lik_variance;
    float* ms; float* logp;
};

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void k_elbo_points(ReduceArgs g) {
    const int lane = threadIdx.x & 63;
    const long long b = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= g.B) return;
    const int K = g.K, Dy = g.Dy;
    const float c0 = -0.5f * 1.8378770664093453f - 0.5f * logf(g.lik_variance);   // -1/2 log 2pi - 1/2 log s2
    const float inv2s = 0.5f / g.lik_variance;
    // pass 1: log-weights of this point (kept in registers for K <= 64*4, recomputed otherwise)
    float m = -INFINITY, ssum = 0.f, lsum = 0.f;
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = k0 + lane;
        float L = -INFINITY;
        if (k < K) {
            const long long t = b * g.stride_b + k * g.stride_k;
            float acc = 0.f;
            for (int d = 0; d < Dy; ++d) {
                float df = g.Y[b * Dy + d] - g.fmean[t * Dy + d];
                acc += c0 - (df * df + g.fvar[t * Dy + d]) * inv2s;                  // models.py:134
            }
            for (int i = 0; i < g.n_kl; ++i)
                for (int d = 0; d < g.kl_dims[i]; ++d) acc -= g.kl[i][t * g.kl_dims[i] + d];   // :140-142
            L = acc;
        }
        if (g.mode_vi) { lsum += wave_sum(k < K ? L : 0.f); continue; }
        // online log-sum-exp across 64-wide chunks
        float cm = wave_max(L);
        float nm = fmaxf(m, cm);
        float e = (k < K) ? __expf(L - nm) : 0.f;
        float cs = wave_sum(e);
        ssum = ssum * __expf(m - nm) + cs;
        m = nm;
    }
    if (lane == 0) {
        if (g.mode_vi) {
            if (g.logp) g.logp[b] = lsum / (float)K;                               // models.py:84
        } else {
            if (g.ms) { g.ms[2 * b] = m; g.ms[2 * b + 1] = ssum; }
            if (g.logp) g.logp[b] = m + logf(ssum) - logf((float)g.K_total);      // models.py:148
        }
    }
}

struct FinalArgs {
    const float* logp; const float* ms_all; int G;
    long long B; int K_total; double scale;
    const double* klg[MAX_GLOB]; int klg_n[MAX_GLOB]; int n_glob;
    float* logp_out; double* elbo;
};

// optional merge of G gathered (max, sumexp) partials per point, then the deterministic final sum
__global__ __launch_bounds__(1024) void k_elbo_final(FinalArgs g) {
    __shared__ double red[1024];
    double acc = 0.0;
    for (long long b = threadIdx.x; b < g.B; b += blockDim.x) {
        float lp;
        if (g.ms_all) {
            float m = -INFINITY;
            for (int r = 0; r < g.G; ++r) m = fmaxf(m, g.ms_all[((size_t)r * g.B + b) * 2]);
            float s = 0.f;
            for (int r = 0; r < g.G; ++r) {
                const float* p = g.ms_all + ((size_t)r * g.B + b) * 2;
                s += p[1] * __expf(p[0] - m);
            }
            lp = m + logf(s) - logf((float)g.K_total);
            if (g.logp_out) g.logp_out[b] = lp;
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
        *g.elbo = red[0] * g.scale - kl;                                          // models.py:150
    }
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller.  Element i of the output uses counter (offset + i/4, 0, 0, 0), key
// (seed_lo, seed_hi), word i%4: words (0,1) -> (r cos, r sin), words (2,3) likewise.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

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
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float u1 = ((float)c[2 * p] + 0.5f) * 2.3283064365386963e-10f;       // (0,1)
            float u2 = ((float)c[2 * p + 1] + 0.5f) * 2.3283064365386963e-10f;
            u1 = fminf(fmaxf(u1, 1.1754944e-38f), 0.99999994f);
            float rad = sqrtf(-2.f * logf(u1));
            float sn, cs;
            sincosf(6.283185307179586f * u2, &sn, &cs);
            v[2 * p] = rad * cs; v[2 * p + 1] = rad * sn;
        }
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
    if (T <= 0) return IWVI_OK;
    if (!F) { set_error("iwvi_lv_layer_forward: null input"); return IWVI_ERR_ARG; }
    if (D <= 0 || Lw <= 0) { set_error("iwvi_lv_layer_forward: bad D=%d or latent_dim=%d", D, Lw); return IWVI_ERR_ARG; }
    LvArgs g{};
    g.F = F; g.XY = XY; g.noise = noise; g.D = D; g.Lw = Lw; g.sampled_kl = sampled_kl;
    g.sample = sample; g.mean = mean; g.cov = cov; g.kl = kl; g.T = T;
    int maxdim = 2 * Lw;
    if (XY) {
        if (!enc_W || !dims || n_enc <= 0 || n_enc > IWVI_MAX_ENC) {
            set_error("iwvi_lv_layer_forward: encoder with %d layers (1..%d supported)", n_enc, IWVI_MAX_ENC); return IWVI_ERR_ARG;
        }
        if (dims[n_enc] != 2 * Lw) { set_error("iwvi_lv_layer_forward: encoder output %d != 2*latent_dim %d", dims[n_enc], 2 * Lw); return IWVI_ERR_ARG; }
        for (int i = 0; i <= n_enc; ++i) {
            if (dims[i] <= 0 || dims[i] > LV_MAXDIM) { set_error("iwvi_lv_layer_forward: encoder width %d out of range (1..%d)", dims[i], LV_MAXDIM); return IWVI_ERR_ARG; }
            g.dims[i] = dims[i];
            if (dims[i] > maxdim) maxdim = dims[i];
        }
        for (int i = 0; i < n_enc; ++i) {
            if (!enc_W[i]) { set_error("iwvi_lv_layer_forward: null encoder weight %d", i); return IWVI_ERR_ARG; }
            g.W[i] = enc_W[i]; g.b[i] = enc_b ? enc_b[i] : nullptr;
        }
        g.n_enc = n_enc;
    }
    if (maxdim > LV_MAXDIM) { set_error("iwvi_lv_layer_forward: latent_dim too large"); return IWVI_ERR_ARG; }
    g.maxdim = maxdim;
    size_t lds = sizeof(float) * 2 * (size_t)maxdim * LV_THREADS;
    long long blocks = (T + LV_THREADS - 1) / LV_THREADS;
    hipLaunchKernelGGL(k_lv_layer, dim3((unsigned)blocks), dim3(LV_THREADS), lds, (hipStream_t)stream_, g);
    return check_launch("k_lv_layer");
}

static int fill_globals(FinalArgs& f, const double* const* klg, const int32_t* counts, int n_glob) {
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

extern "C" int iwvi_iw_elbo_reduce(const float* fmean, const float* fvar, const float* Y,
                                   float lik_variance, int64_t B, int K, int Dy,
                                   int64_t stride_b, int64_t stride_k,
                                   const float* const* kl_local, const int32_t* kl_dims, int n_kl,
                                   const double* const* kl_global, const int32_t* kl_global_counts, int n_glob,
                                   double scale, int K_total, int mode_vi,
                                   float* out_ms, float* out_logp, double* out_elbo, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!fmean || !fvar || !Y) { set_error("iwvi_iw_elbo_reduce: null input"); return IWVI_ERR_ARG; }
    if (B <= 0) { set_error("iwvi_iw_elbo_reduce: empty minibatch"); return IWVI_ERR_ARG; }
    if (K <= 0 || Dy <= 0 || !(lik_variance > 0.f)) { set_error("iwvi_iw_elbo_reduce: bad K=%d, Dy=%d or likelihood variance", K, Dy); return IWVI_ERR_ARG; }
    if (n_kl < 0 || n_kl > IWVI_MAX_KL) { set_error("iwvi_iw_elbo_reduce: %d local regularisers (max %d)", n_kl, IWVI_MAX_KL); return IWVI_ERR_ARG; }
    if (out_elbo && !out_logp) { set_error("iwvi_iw_elbo_reduce: out_elbo needs out_logp as scratch"); return IWVI_ERR_ARG; }
    ReduceArgs g{};
    g.fmean = fmean; g.fvar = fvar; g.Y = Y; g.n_kl = n_kl;
    for (int i = 0; i < n_kl; ++i) {
        if (!kl_local || !kl_local[i] || !kl_dims || kl_dims[i] <= 0) { set_error("iwvi_iw_elbo_reduce: bad local regulariser %d", i); return IWVI_ERR_ARG; }
        g.kl[i] = kl_local[i]; g.kl_dims[i] = kl_dims[i];
    }
    g.stride_b = stride_b; g.stride_k = stride_k;
    g.B = B; g.K = K; g.Dy = Dy; g.K_total = K_total > 0 ? K_total : K; g.mode_vi = mode_vi;
    g.lik_variance = lik_variance; g.ms = out_ms; g.logp = out_logp;
    long long blocks = (B + 3) / 4;
    hipLaunchKernelGGL(k_elbo_points, dim3((unsigned)blocks), dim3(256), 0, stream, g);
    int rc = check_launch("k_elbo_points");
    if (rc != IWVI_OK || !out_elbo) return rc;
    FinalArgs f{};
    f.logp = out_logp; f.B = B; f.K_total = g.K_total; f.scale = scale; f.elbo = out_elbo;
    if ((rc = fill_globals(f, kl_global, kl_global_counts, n_glob)) != IWVI_OK) return rc;
    hipLaunchKernelGGL(k_elbo_final, dim3(1), dim3(1024), 0, stream, f);
    return check_launch("k_elbo_final");
}

extern "C" int iwvi_lse_merge(const float* ms_all, int G, int64_t B, int K_total,
                              const double* const* kl_global, const int32_t* kl_global_counts, int n_glob,
                              double scale, float* out_logp, double* out_elbo, void* stream_) {
    if (!ms_all || G <= 0 || B <= 0 || K_total <= 0) { set_error("iwvi_lse_merge: bad argument"); return IWVI_ERR_ARG; }
    FinalArgs f{};
    f.ms_all = ms_all; f.G = G; f.B = B; f.K_total = K_total; f.scale = scale;
    f.logp_out = out_logp; f.elbo = out_elbo;
    int rc;
    if ((rc = fill_globals(f, kl_global, kl_global_counts, n_glob)) != IWVI_OK) return rc;
    hipLaunchKernelGGL(k_elbo_final, dim3(1), dim3(1024), 0, (hipStream_t)stream_, f);
    return check_launch("k_elbo_final(merge)");
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

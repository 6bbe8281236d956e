// LatentVariableLayer forward (+ Encoder MLP), the IW-ELBO reduction, the K-shard merge and the
// counter-based normal generator.  Reference: layers.py:72-105,137-152; models.py:133-150.
#include "iwvi_common.h"

namespace iwvi {

// ------------------------------------------------------------------------------------------
// LatentVariableLayer: 32 lanes cooperate on one encoder row (one lane per output unit of each MLP
// layer, weights staged in LDS once per workgroup), then write that row's bcast_K samples.  In the
// IW path the encoder input [x_b, y_b] is the same for all K samples of a point (models.py:113-116), so
// the MLP runs once per point and F / XY are read untiled ([B, .] rows, bcast_K = K).
// ------------------------------------------------------------------------------------------
constexpr int LV_THREADS = 256;
constexpr int LV_GROUPS = LV_THREADS / 32;
constexpr int LV_MAXDIM = 64;

struct LvArgs {
    const float* F; const float* XY; const float* noise;
    const float* W[IWVI_MAX_ENC]; const float* b[IWVI_MAX_ENC];
    int dims[IWVI_MAX_ENC + 1];
    int n_enc, D, Lw, sampled_kl, maxdim, bcast_K, bcast_F, wtotal;
    float* sample; float* mean; float* cov; float* kl;
    long long E;                      // encoder rows = T / bcast_K
};

extern __shared__ __attribute__((aligned(16))) unsigned char lv_smem[];

__device__ __forceinline__ float softplus_f(float x) {
    return x > 20.f ? x : log1pf(expf(x));
}

__global__ __launch_bounds__(LV_THREADS) void k_lv_layer(LvArgs g) {
    const int tid = threadIdx.x, grp = tid >> 5, ln = tid & 31;
    const long long e = (long long)blockIdx.x * LV_GROUPS + grp;
    const bool live = e < g.E;
    const int D = g.D, Lw = g.Lw, K = g.bcast_K, Do = D + Lw;
    // LDS: weights+biases of all layers | per group: act[2][maxdim], frow[32]
    float* wts = reinterpret_cast<float*>(lv_smem);
    float* gbase = wts + g.wtotal + grp * (2 * g.maxdim + 32);
    float* act0 = gbase;
    float* act1 = gbase + g.maxdim;
    float* frow = gbase + 2 * g.maxdim;
    if (g.XY) {
        int off = 0;
        for (int l = 0; l < g.n_enc; ++l) {
            const int nW = g.dims[l] * g.dims[l + 1], nb = g.dims[l + 1];
            for (int i = tid; i < nW; i += LV_THREADS) wts[off + i] = g.W[l][i];
            for (int i = tid; i < nb; i += LV_THREADS) wts[off + nW + i] = g.b[l] ? g.b[l][i] : 0.f;
            off += nW + nb;
        }
        const int d0 = g.dims[0];
        for (int i = ln; i < d0; i += 32) act0[i] = live ? g.XY[e * d0 + i] : 0.f;
    }
    if (ln < D) frow[ln] = (live && g.bcast_F) ? g.F[e * D + ln] : 0.f;
    __syncthreads();
    float* in = act0; float* out = act1;
    if (g.XY) {
        int off = 0;
        for (int l = 0; l < g.n_enc; ++l) {
            const int din = g.dims[l], dout = g.dims[l + 1];
            const float* W = wts + off; const float* b = W + din * dout;
            for (int o = ln; o < dout; o += 32) {
                float acc = b[o];
                for (int i = 0; i < din; ++i) acc = fmaf(in[i], W[i * dout + o], acc);
                if (l < g.n_enc - 1) acc = tanhf(acc);                         // layers.py:143-144
                if (din == dout) acc += in[o];                                 // layers.py:146-147
                out[o] = acc;
            }
            off += din * dout + dout;
            __syncthreads();
            float* tmp = in; in = out; out = tmp;
        }
    }
    if (!live) return;
    // `in` holds [means (Lw) | raw (Lw)]; this group's K samples are rows e*K .. e*K+K-1 (contiguous)
    const long long t0 = e * K;
    for (int idx = ln; idx < K * Do; idx += 32) {
        const int k = idx / Do, c = idx - k * Do;
        float vs, vm, vc;
        if (c < D) { vs = vm = g.bcast_F ? frow[c] : g.F[(t0 + k) * D + c]; vc = 0.f; }
        else {
            const int l = c - D;
            float mu = 0.f, sg = 1.f;                                           // prior (layers.py:73-81)
            if (g.XY) { mu = in[l]; sg = softplus_f(in[Lw + l] - 3.f); }
            const float z = g.noise ? g.noise[(t0 + k) * Lw + l] : 0.f;
            vs = fmaf(z, sg, mu); vm = mu; vc = sg * sg;                        // layers.py:86-91
        }
        if (g.sample) g.sample[t0 * Do + idx] = vs;
        if (g.mean) g.mean[t0 * Do + idx] = vm;
        if (g.cov) g.cov[t0 * Do + idx] = vc;
    }
    if (g.kl) {
        for (int idx = ln; idx < K * Lw; idx += 32) {
            const int l = idx % Lw;
            float mu = 0.f, sg = 1.f;
            if (g.XY) { mu = in[l]; sg = softplus_f(in[Lw + l] - 3.f); }
            const float z = g.noise ? g.noise[t0 * Lw + idx] : 0.f;
            const float w = fmaf(z, sg, mu);
            float kl;
            if (g.sampled_kl) kl = -0.5f * z * z - logf(sg) + 0.5f * w * w;     // log q(W) - log p(W), :98-100
            else kl = 0.5f * (sg * sg + mu * mu - 1.f) - logf(sg);              // KL(N(mu,sg)||N(0,1)), :101-103
            g.kl[t0 * Lw + idx] = kl;
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
    const float c0 = -0.5f * 1.8378770664093453f - 0.5f * logf(g.lik_variance);
    const float inv2s = 0.5f / g.lik_variance;
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
                for (int d = 0; d < Dy; ++d) {
                    const float df = g.Y[b * Dy + d] - g.fmean[t * Dy + d];
                    acc += c0 - (df * df + g.fvar[t * Dy + d]) * inv2s;
                }
                for (int i = 0; i < g.n_kl; ++i)
                    for (int d = 0; d < g.kl_dims[i]; ++d) acc -= g.kl[i][t * g.kl_dims[i] + d];
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
        const int last = ((t + 1) % gridDim.x) == 0;
        if (last) {
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
                                     int64_t T, int bcast_K, int bcast_F, void* stream_) {
    if (T <= 0) return IWVI_OK;
    if (!F) { set_error("iwvi_lv_layer_forward: null input"); return IWVI_ERR_ARG; }
    if (D <= 0 || D > 32 || Lw <= 0) { set_error("iwvi_lv_layer_forward: bad D=%d (1..32) or latent_dim=%d", D, Lw); return IWVI_ERR_ARG; }
    if (bcast_K < 1 || T % bcast_K != 0) { set_error("iwvi_lv_layer_forward: T=%lld is not a multiple of bcast_K=%d", (long long)T, bcast_K); return IWVI_ERR_ARG; }
    LvArgs g{};
    g.F = F; g.XY = XY; g.noise = noise; g.D = D; g.Lw = Lw; g.sampled_kl = sampled_kl;
    g.sample = sample; g.mean = mean; g.cov = cov; g.kl = kl; g.E = T / bcast_K; g.bcast_K = bcast_K; g.bcast_F = (bcast_F || bcast_K == 1) ? 1 : 0;
    int maxdim = 2 * Lw, wtotal = 0;
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
            wtotal += dims[i] * dims[i + 1] + dims[i + 1];
        }
        g.n_enc = n_enc;
    }
    if (maxdim > LV_MAXDIM) { set_error("iwvi_lv_layer_forward: latent_dim too large"); return IWVI_ERR_ARG; }
    g.maxdim = maxdim; g.wtotal = wtotal;
    size_t lds = sizeof(float) * ((size_t)wtotal + (size_t)LV_GROUPS * (2 * maxdim + 32));
    long long blocks = (g.E + LV_GROUPS - 1) / LV_GROUPS;
    hipLaunchKernelGGL(k_lv_layer, dim3((unsigned)blocks), dim3(LV_THREADS), lds, (hipStream_t)stream_, g);
    return check_launch("k_lv_layer");
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
    g.lik_variance = lik_variance; g.ms = out_ms; g.logp = out_logp;
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

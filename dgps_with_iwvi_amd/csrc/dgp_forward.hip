// The whole layer stack of one IW-ELBO evaluation for a chunk of samples, fused into one launch (gfx950).
//
// Replaces, per GP layer (reference file:line): Kuf (temp_workaround.py:44), matrix_triangular_solve (:51),
// Kdiag - sum A^2 (:59), A^T q_mu (:68), einsum('rMm,sMn->srmn') (:78), + sum LTA^2 (:85), the marginal sample
// (:89-91), the SharedMixedMok mixing (:142-145), the mean-function add (layers.py:46-48); per latent-variable
// layer layers.py:72-105 with its encoder (:137-152); the tiling of models.py:113-116 and the Gaussian
// variational expectation minus the local regularisers (models.py:134-142).  Every sample's path through the
// layers is independent of the other samples, so nothing the reference materialises (Kmn, A, LTA, the tiled
// inputs, the per-layer samples) reaches HBM unless the caller asks for a per-layer output.
//
// One 512-thread workgroup (8 waves, two per SIMD) owns a chunk of 16*NS samples, all layers.  Per GP layer:
//   x~      augmented, scaled, centred inputs -> LDS (so that the Gram is one small MFMA product)
//   Gram    k = exp2(Z~ x~)               16x16x4 MFMA, result already in B-operand order      -> LDS kuf
//   stage 1 a = Lm^-1 k  (lower blocks),  mean = (Lm^-T q_mu)^T k                               -> LDS at, |a|^2
//   stage 2 u_r = tril(q_sqrt_r)^T a (upper blocks) -> |u_r|^2 only (never stored)
//   epilogue var, sample, mixing, mean function -> next layer's input in LDS (+ optional HBM outputs)
// A wave owns one 16-row block of the output for ALL NS sample sub-tiles of the chunk: each 1-KiB packed A
// block is loaded once (coalesced, straight from L2 to registers, prefetched one block ahead) and feeds
// 4*NS MFMAs; B operands are ds_read_b128 of the LDS tile.  Row-block jobs are handed out from an LDS
// counter in order of decreasing cost, so the triangular imbalance is absorbed inside the workgroup, and
// every job writes its partial sums to its own LDS slot: results are bit-reproducible whichever wave ran it.
#include "iwvi_common.h"

namespace iwvi {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int FW_THREADS = 512;
constexpr int FW_WAVES = FW_THREADS / 64;
constexpr int XSTR = 37;              // row stride (floats) of the activation tiles: D, P <= 32, D + 2 <= 36
constexpr int FW_MAXNS = 5;

struct FwGp {
    const f32x4* LinvP; const f32x4* LrTP; const f32x4* WqP; const float* ZtP; const float* zc; const float* invls;
    const float* W; const float* mfA; const float* mfb;
    float* a_out; float* u_out;
    int M, Mp, nbk, nrb, nsteps, R, P, kern_type, mf_type;
    float variance;
};
struct FwLv {
    const float* W[IWVI_MAX_ENC]; const float* b[IWVI_MAX_ENC];
    float* kl_local;
    int dims[IWVI_MAX_ENC + 1];
    int n_enc, Lw, sampled_kl, wtotal, maxdim;
};
struct FwLayer {
    int type, D, zero_noise;
    const float* noise; float* noise_out; float* sample; float* mean; float* var;
    union { FwGp gp; FwLv lv; };
};
// LDS carve, float offsets (all multiples of 4)
struct FwLds { int xa, xb, xt, lw, rowi, pidx, asq, meanp, gbuf, obuf, cnt, scratch, total; };

struct FwArgs {
    FwLayer L[IWVI_MAX_STACK];
    int n_layers;
    const float* X; const float* XY; const float* Y;
    int Dx, XYdim, Dy;
    long long T, row_div, row_mod;
    float lik_variance;
    unsigned long long seed; unsigned long long* rng_state;
    float* out_logw;
    FwLds lds;
};

__host__ __device__ static inline int up4(int x) { return (x + 3) & ~3; }

// scratch needs (floats) of a layer for a chunk of nsamp samples
static inline int gp_scratch_floats(int Mp, int nbk, int R, int nsamp) {
    // Gram tile + solved tile; |u|^2 slots alias the (dead) Gram tile when they fit, else get their own region
    return 2 * Mp * nsamp + (R * nbk > Mp ? R * nbk * nsamp : 0);
}
static inline int lv_scratch_floats(int wtotal, int maxdim, int Lw, int nsamp) {
    return up4(wtotal) + 2 * nsamp * up4(maxdim);
}
static inline FwLds fw_lds_layout(int nsamp, int maxR, int maxP, int max_nbk, int scratch) {
    FwLds l; int o = 0;
    l.xa = o; o += up4(nsamp * XSTR);
    l.xb = o; o += up4(nsamp * XSTR);
    l.xt = o; o += up4(nsamp * XSTR);
    l.lw = o; o += nsamp;
    l.rowi = o; o += nsamp;
    l.pidx = o; o += nsamp;
    l.asq = o; o += max_nbk * nsamp;
    l.meanp = o; o += maxR * nsamp;
    l.gbuf = o; o += 3 * maxR * nsamp;
    l.obuf = o; o += 2 * maxP * nsamp;
    l.cnt = o; o += 4;
    l.scratch = o; o += up4(scratch);
    l.total = o;
    return l;
}

__device__ __forceinline__ float kern_from_acc(float acc, int type, float var) {
    if (type == IWVI_KERN_MATERN52) {
        const float s5 = 2.2360679774997896f;
        const float r2 = fmaxf(acc, 0.f);
        const float r = sqrtf(r2 + 1e-12f);
        return var * (1.0f + s5 * r + (5.0f / 3.0f) * r2) * __expf(-s5 * r);
    }
    return __builtin_amdgcn_exp2f(acc);            // log2(var) is folded into Z~
}

// acc[t] += A(row-block, chunks c0 .. c0+nch-1) * B(chunks, sub-tile t); Ablk = first packed block of the run,
// Bt = the LDS tile at chunk c0.  One 1-KiB A load (prefetched one block ahead) per 4*NS MFMAs.
template <int NS>
__device__ __forceinline__ void mma_rowblock(const f32x4* __restrict__ Ablk, int nch, const f32x4* Bt, int lane,
                                             f32x4 (&acc)[NS]) {
    constexpr int NSAMP = 16 * NS;
    const int g = lane >> 4, j = lane & 15;
    const f32x4* Ap = Ablk + lane;
    const f32x4* Bp = Bt + g * NSAMP + j;
    f32x4 a = Ap[0];
    for (int c = 0; c < nch; ++c) {
        const f32x4 an = Ap[(size_t)(c + 1 < nch ? c + 1 : c) * 64];
        f32x4 b[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) b[t] = Bp[16 * t];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[t][s], acc[t], 0, 0, 0);
        }
        a = an;
        Bp += 4 * NSAMP;
    }
}

// sum over the 16 rows of a result block, per sample column: 4 registers, then across the 4 lane groups
__device__ __forceinline__ float colsumsq(const f32x4& v) {
    float s = v[0] * v[0];
    s = fmaf(v[1], v[1], s); s = fmaf(v[2], v[2], s); s = fmaf(v[3], v[3], s);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    return s;
}

// N(0,1) draw for (layer li, sample t, component r): Philox4x32-10 with
//   counter = (t_lo, t_hi, li * 256 + r / 4, step_lo), key = (seed_lo, seed_hi ^ step_hi), word r % 4
__device__ __forceinline__ float draw_normal_at(unsigned long long seed, unsigned long long step, int li,
                                                long long t, int r) {
    uint32_t c[4] = {(uint32_t)t, (uint32_t)((unsigned long long)t >> 32), (uint32_t)(li * 256 + (r >> 2)), (uint32_t)step};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(step >> 32));
    float v[4];
    box_muller4(c, v);
    return v[r & 3];
}

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

extern __shared__ __attribute__((aligned(16))) unsigned char fw_smem[];

template <int NS>
__global__ __launch_bounds__(FW_THREADS) void k_dgp_forward(const FwArgs g) {
    constexpr int NSAMP = 16 * NS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gq = lane >> 4, jq = lane & 15;
    const long long t0 = (long long)blockIdx.x * NSAMP;
    const int nvalid = (int)((g.T - t0) < (long long)NSAMP ? (g.T - t0) : (long long)NSAMP);

    float* sm = reinterpret_cast<float*>(fw_smem);
    float* xin = sm + g.lds.xa;
    float* xout = sm + g.lds.xb;
    float* xt = sm + g.lds.xt;
    float* lw = sm + g.lds.lw;
    int* rowi = reinterpret_cast<int*>(sm + g.lds.rowi);
    int* pidx = reinterpret_cast<int*>(sm + g.lds.pidx);
    float* asq = sm + g.lds.asq;
    float* meanp = sm + g.lds.meanp;
    float* gbuf = sm + g.lds.gbuf;
    float* obuf = sm + g.lds.obuf;
    int* counters = reinterpret_cast<int*>(sm + g.lds.cnt);
    float* scratch = sm + g.lds.scratch;

    const unsigned long long step = g.rng_state ? g.rng_state[0] : 0ULL;

    // ---- the chunk's rows of X (models.py:113 / :50 tiling done here) -----------------------------
    const long long p_first = t0 / g.row_div;
    if (tid < NSAMP) {
        const long long t = t0 + tid;
        const long long p = (t < g.T ? t : g.T - 1) / g.row_div;
        rowi[tid] = (int)(p % g.row_mod);
        pidx[tid] = (int)(p - p_first);
        lw[tid] = 0.f;
    }
    __syncthreads();
    for (int idx = tid; idx < NSAMP * g.Dx; idx += FW_THREADS) {
        const int j = idx / g.Dx, d = idx - j * g.Dx;
        xin[j * XSTR + d] = (j < nvalid) ? g.X[(size_t)rowi[j] * g.Dx + d] : 0.f;
    }
    __syncthreads();

    for (int li = 0; li < g.n_layers; ++li) {
        const FwLayer& L = g.L[li];
        const int D = L.D;
        if (L.type == IWVI_LAYER_LV) {
            // ================= LatentVariableLayer (layers.py:72-105) =================================
            const FwLv& V = L.lv;
            const int Lw = V.Lw, Do = D + Lw;
            const int npts = pidx[nvalid - 1] + 1;               // distinct data points in this chunk
            const int mdim = up4(V.maxdim);
            float* wts = scratch;
            float* act0 = scratch + up4(V.wtotal);
            float* act1 = act0 + NSAMP * mdim;
            float* in = act0; float* out = act1;
            if (V.n_enc > 0) {
                int off = 0;
                for (int l = 0; l < V.n_enc; ++l) {
                    const int nW = V.dims[l] * V.dims[l + 1], nbias = V.dims[l + 1];
                    for (int i = tid; i < nW; i += FW_THREADS) wts[off + i] = V.W[l][i];
                    for (int i = tid; i < nbias; i += FW_THREADS) wts[off + nW + i] = V.b[l] ? V.b[l][i] : 0.f;
                    off += nW + nbias;
                }
                const int d0 = V.dims[0];
                for (int idx = tid; idx < npts * d0; idx += FW_THREADS) {
                    const int p = idx / d0, i = idx - p * d0;
                    const long long row = (p_first + p) % g.row_mod;
                    act0[p * mdim + i] = g.XY[(size_t)row * g.XYdim + i];
                }
                __syncthreads();
                off = 0;
                for (int l = 0; l < V.n_enc; ++l) {
                    const int din = V.dims[l], dout = V.dims[l + 1];
                    const float* W = wts + off; const float* b = W + din * dout;
                    for (int idx = tid; idx < npts * dout; idx += FW_THREADS) {
                        const int p = idx / dout, o = idx - p * dout;
                        float acc = b[o];
                        for (int i = 0; i < din; ++i) acc = fmaf(in[p * mdim + i], W[i * dout + o], acc);
                        if (l < V.n_enc - 1) acc = tanhf(acc);                       // layers.py:143-144
                        if (din == dout) acc += in[p * mdim + o];                    // layers.py:146-147
                        out[p * mdim + o] = acc;
                    }
                    off += din * dout + dout;
                    __syncthreads();
                    float* tmp = in; in = out; out = tmp;
                }
            }
            // `in` rows hold [means (Lw) | raw (Lw)] per distinct point
            for (int idx = tid; idx < NSAMP * D; idx += FW_THREADS) {
                const int j = idx / D, c = idx - j * D;
                const float v = xin[j * XSTR + c];
                xout[j * XSTR + c] = v;
                if (j < nvalid) {
                    const long long t = t0 + j;
                    if (L.sample) L.sample[t * Do + c] = v;
                    if (L.mean) L.mean[t * Do + c] = v;
                    if (L.var) L.var[t * Do + c] = 0.f;
                }
            }
            if (tid < NSAMP) {
                const int j = tid;
                const long long t = t0 + j;
                float klsum = 0.f;
                for (int l = 0; l < Lw; ++l) {
                    float mu = 0.f, sg = 1.f;                                        // prior (layers.py:73-81)
                    if (V.n_enc > 0) { mu = in[pidx[j] * mdim + l]; sg = softplus_f(in[pidx[j] * mdim + Lw + l] - 3.f); }
                    float z = 0.f;
                    if (j < nvalid) z = L.noise ? L.noise[t * Lw + l] : (L.zero_noise ? 0.f : draw_normal_at(g.seed, step, li, t, l));
                    const float w = fmaf(z, sg, mu);                                 // layers.py:86-87
                    float kl;
                    if (V.sampled_kl) kl = -0.5f * z * z - logf(sg) + 0.5f * w * w;  // log q(W) - log p(W), :98-100
                    else kl = 0.5f * (sg * sg + mu * mu - 1.f) - logf(sg);           // KL(N(mu,sg)||N(0,1)), :101-103
                    klsum += kl;
                    xout[j * XSTR + D + l] = w;
                    if (j < nvalid) {
                        if (V.kl_local) V.kl_local[t * Lw + l] = kl;
                        if (L.noise_out) L.noise_out[t * Lw + l] = z;
                        if (L.sample) L.sample[t * Do + D + l] = w;                  // layers.py:89-91
                        if (L.mean) L.mean[t * Do + D + l] = mu;
                        if (L.var) L.var[t * Do + D + l] = sg * sg;
                    }
                }
                lw[j] += klsum;
            }
            __syncthreads();
        } else {
            // ================= GPLayer (layers.py:35-50) ==============================================
            const FwGp& G = L.gp;
            const int nbk = G.nbk, R = G.R, P = G.P, nsteps = G.nsteps;
            f32x4* kuf = reinterpret_cast<f32x4*>(scratch);
            f32x4* at = reinterpret_cast<f32x4*>(scratch + (size_t)G.Mp * NSAMP);
            float* usq = (R * nbk <= G.Mp) ? scratch : scratch + (size_t)2 * G.Mp * NSAMP;
            const bool rbf = G.kern_type == IWVI_KERN_RBF;

            // ---- x~ = [x/l - zc, -|.|^2/2 (RBF) or |.|^2 (Matern52), 1, 0..] ----------------------------
            if (tid < NSAMP) {
                float n2 = 0.f;
                for (int d = 0; d < D; ++d) {
                    const float v = fmaf(xin[tid * XSTR + d], G.invls[d], -G.zc[d]);
                    xt[tid * XSTR + d] = v;
                    n2 = fmaf(v, v, n2);
                }
                xt[tid * XSTR + D] = rbf ? -0.5f * n2 : n2;
                xt[tid * XSTR + D + 1] = 1.f;
                for (int d = D + 2; d < 4 * nsteps; ++d) xt[tid * XSTR + d] = 0.f;
            }
            if (tid < 4) counters[tid] = 0;
            __syncthreads();

            // ---- Gram: kuf block bi = kernel(Z~_bi x~^T), written in B-operand order -----------------------
            for (int bi = wave; bi < nbk; bi += FW_WAVES) {
                f32x4 acc[NS];
#pragma unroll
                for (int t = 0; t < NS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                const float* zp = G.ZtP + (size_t)bi * nsteps * 64 + lane;
                for (int s = 0; s < nsteps; ++s) {
                    const float a = zp[s * 64];
#pragma unroll
                    for (int t = 0; t < NS; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xt[(16 * t + jq) * XSTR + 4 * s + gq], acc[t], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                    f32x4 k;
#pragma unroll
                    for (int e = 0; e < 4; ++e) k[e] = kern_from_acc(acc[t][e], G.kern_type, G.variance);
                    kuf[(bi * 4 + gq) * NSAMP + 16 * t + jq] = k;
                }
            }
            __syncthreads();

            // ---- stage 1: mean row-blocks (cost nbk) then a-blocks nbk-1 .. 0 (cost bi + 1) -------------
            {
                const int nmb = G.nrb, njobs = nmb + nbk;
                for (;;) {
                    int job = 0;
                    if (lane == 0) job = atomicAdd(&counters[0], 1);
                    job = __builtin_amdgcn_readfirstlane(job);
                    if (job >= njobs) break;
                    f32x4 acc[NS];
#pragma unroll
                    for (int t = 0; t < NS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (job < nmb) {
                        mma_rowblock<NS>(G.WqP + (size_t)job * nbk * 64, nbk, kuf, lane, acc);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = 16 * job + 4 * gq + e;
                            if (r < R) {
#pragma unroll
                                for (int t = 0; t < NS; ++t) meanp[r * NSAMP + 16 * t + jq] = acc[t][e];
                            }
                        }
                    } else {
                        const int bi = nbk - 1 - (job - nmb);
                        mma_rowblock<NS>(G.LinvP + (size_t)tri_lower_off(bi) * 64, bi + 1, kuf, lane, acc);
#pragma unroll
                        for (int t = 0; t < NS; ++t) {
                            at[(bi * 4 + gq) * NSAMP + 16 * t + jq] = acc[t];
                            const float s = colsumsq(acc[t]);
                            if (gq == 0) asq[bi * NSAMP + 16 * t + jq] = s;
                            if (G.a_out) {
                                const int j = 16 * t + jq;
                                if (j < nvalid)
                                    *reinterpret_cast<f32x4*>(G.a_out + (size_t)(t0 + j) * G.Mp + 16 * bi + 4 * gq) = acc[t];
                            }
                        }
                    }
                }
            }
            __syncthreads();

            // ---- stage 2: u block (r, bi) = sum_{bk >= bi} LrT(bi, bk) a(bk); only |u|^2 is kept ------------
            {
                const int njobs = R * nbk, ntri = tri_blocks(nbk);
                for (;;) {
                    int job = 0;
                    if (lane == 0) job = atomicAdd(&counters[1], 1);
                    job = __builtin_amdgcn_readfirstlane(job);
                    if (job >= njobs) break;
                    const int bi = job / R, r = job - bi * R;
                    f32x4 acc[NS];
#pragma unroll
                    for (int t = 0; t < NS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    mma_rowblock<NS>(G.LrTP + ((size_t)r * ntri + tri_upper_off(nbk, bi)) * 64, nbk - bi,
                                     at + (size_t)bi * 4 * NSAMP, lane, acc);
#pragma unroll
                    for (int t = 0; t < NS; ++t) {
                        const float s = colsumsq(acc[t]);
                        if (gq == 0) usq[(r * nbk + bi) * NSAMP + 16 * t + jq] = s;
                        if (G.u_out) {
                            const int j = 16 * t + jq;
                            if (j < nvalid)
                                *reinterpret_cast<f32x4*>(G.u_out + ((size_t)r * g.T + (t0 + j)) * G.Mp + 16 * bi + 4 * gq) = acc[t];
                        }
                    }
                }
            }
            __syncthreads();

            // ---- epilogue (i): per (sample, latent GP): variance, sample (temp_workaround.py:59,85,89-91) ----
            for (int idx = tid; idx < NSAMP * R; idx += FW_THREADS) {
                const int r = idx / NSAMP, j = idx - r * NSAMP;
                const long long t = t0 + j;
                float a2 = 0.f, u2 = 0.f;
                for (int i = 0; i < nbk; ++i) { a2 += asq[i * NSAMP + j]; u2 += usq[(r * nbk + i) * NSAMP + j]; }
                const float mu = meanp[r * NSAMP + j];
                const float v = fmaxf(G.variance - a2 + u2, 0.f);
                float z = 0.f;
                if (j < nvalid) {
                    z = L.noise ? L.noise[t * R + r] : (L.zero_noise ? 0.f : draw_normal_at(g.seed, step, li, t, r));
                    if (L.noise_out) L.noise_out[t * R + r] = z;
                }
                gbuf[(0 * R + r) * NSAMP + j] = mu;
                gbuf[(1 * R + r) * NSAMP + j] = v;
                gbuf[(2 * R + r) * NSAMP + j] = fmaf(z, sqrtf(v), mu);
            }
            __syncthreads();
            // ---- epilogue (ii): mixing (:142-145) + mean function (layers.py:46-48) -> next layer's input ----
            const bool last = (li == g.n_layers - 1);
            for (int idx = tid; idx < NSAMP * P; idx += FW_THREADS) {
                const int p = idx / NSAMP, j = idx - p * NSAMP;
                const long long t = t0 + j;
                float o_s, o_m, o_v;
                if (G.W) {
                    o_s = o_m = o_v = 0.f;
                    for (int r = 0; r < R; ++r) {
                        const float w = G.W[p * R + r];
                        o_m = fmaf(w, gbuf[(0 * R + r) * NSAMP + j], o_m);
                        o_v = fmaf(w * w, gbuf[(1 * R + r) * NSAMP + j], o_v);
                        o_s = fmaf(w, gbuf[(2 * R + r) * NSAMP + j], o_s);
                    }
                } else {
                    o_m = gbuf[(0 * R + p) * NSAMP + j]; o_v = gbuf[(1 * R + p) * NSAMP + j]; o_s = gbuf[(2 * R + p) * NSAMP + j];
                }
                float mf = 0.f;
                if (G.mf_type == IWVI_MF_IDENTITY) mf = xin[j * XSTR + p];
                else if (G.mf_type == IWVI_MF_LINEAR) {
                    for (int d = 0; d < D; ++d) mf = fmaf(xin[j * XSTR + d], G.mfA[d * P + p], mf);
                    if (G.mfb) mf += G.mfb[p];
                }
                xout[j * XSTR + p] = o_s + mf;
                if (last) { obuf[p * NSAMP + j] = o_m + mf; obuf[(P + p) * NSAMP + j] = o_v; }
                if (j < nvalid) {
                    if (L.sample) L.sample[t * P + p] = o_s + mf;
                    if (L.mean) L.mean[t * P + p] = o_m + mf;
                    if (L.var) L.var[t * P + p] = o_v;
                }
            }
            __syncthreads();
        }
        float* tmp = xin; xin = xout; xout = tmp;
    }

    // ---- per-sample log-weight: Gaussian variational expectation (models.py:134,138) minus the local
    //      regularisers (:140-142) ----------------------------------------------------------------------
    if (g.out_logw && tid < nvalid) {
        const int Dy = g.Dy;
        const float c0 = -0.5f * 1.8378770664093453f - 0.5f * logf(g.lik_variance);
        const float inv2s = 0.5f / g.lik_variance;
        float acc = 0.f;
        for (int d = 0; d < Dy; ++d) {
            const float df = g.Y[(size_t)rowi[tid] * Dy + d] - obuf[d * NSAMP + tid];
            acc += c0 - (df * df + obuf[(Dy + d) * NSAMP + tid]) * inv2s;
        }
        g.out_logw[t0 + tid] = acc - lw[tid];
    }
    // ---- advance the noise stream once every workgroup has read the step counter --------------------
    if (g.rng_state) {
        __syncthreads();
        if (tid == 0) {
            const unsigned long long tk = atomicAdd(&g.rng_state[1], 1ULL);
            if (tk == (unsigned long long)gridDim.x - 1) {
                atomicExch(&g.rng_state[1], 0ULL);
                atomicAdd(&g.rng_state[0], 1ULL);
            }
        }
    }
}

template <int NS>
static int launch_forward(const FwArgs& a, unsigned grid, size_t lds_bytes, hipStream_t stream) {
    static size_t attr_set = 0;
    if (lds_bytes > attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_dgp_forward<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(k_dgp_forward, %zu B): %s", lds_bytes, hipGetErrorString(e)); return IWVI_ERR_LAUNCH; }
        attr_set = lds_bytes;
    }
    hipLaunchKernelGGL(k_dgp_forward<NS>, dim3(grid), dim3(FW_THREADS), lds_bytes, stream, a);
    return check_launch("k_dgp_forward");
}

int dgp_forward_impl(const iwvi_layer_desc* layers, int n_layers, const float* X, int Dx, const float* XY, int XYdim,
                     const float* Y, int Dy, int64_t T, int64_t row_div, int64_t row_mod, float lik_variance,
                     uint64_t seed, uint64_t* rng_state, float* out_logw, hipStream_t stream) {
    if (T <= 0) return IWVI_OK;                         // empty batch: nothing to do
    if (!layers || n_layers <= 0 || n_layers > IWVI_MAX_STACK) { set_error("iwvi_dgp_forward: %d layers (1..%d supported)", n_layers, IWVI_MAX_STACK); return IWVI_ERR_ARG; }
    if (!X || Dx <= 0 || Dx > IWVI_MAX_D) { set_error("iwvi_dgp_forward: null X or Dx=%d out of range (1..%d)", Dx, IWVI_MAX_D); return IWVI_ERR_ARG; }
    if (row_div < 1 || row_mod < 1) { set_error("iwvi_dgp_forward: row_div=%lld / row_mod=%lld must be >= 1", (long long)row_div, (long long)row_mod); return IWVI_ERR_ARG; }
    if (row_mod > 0x7fffffffLL) { set_error("iwvi_dgp_forward: more than 2^31 data rows"); return IWVI_ERR_ARG; }
    if (out_logw && (!Y || Dy <= 0 || Dy > IWVI_MAX_P || !(lik_variance > 0.f))) { set_error("iwvi_dgp_forward: out_logw needs Y, 1 <= Dy <= %d and a positive likelihood variance", IWVI_MAX_P); return IWVI_ERR_ARG; }
    FwArgs a{};
    a.n_layers = n_layers; a.X = X; a.XY = XY; a.Y = Y; a.Dx = Dx; a.XYdim = XYdim; a.Dy = Dy;
    a.T = T; a.row_div = row_div; a.row_mod = row_mod; a.lik_variance = lik_variance;
    a.seed = seed; a.rng_state = (unsigned long long*)rng_state; a.out_logw = out_logw;
    int D = Dx, maxR = 1, maxP = 1, max_nbk = 1;
    bool need_rng = false;
    for (int i = 0; i < n_layers; ++i) {
        const iwvi_layer_desc& d = layers[i];
        FwLayer& L = a.L[i];
        L.type = d.type; L.D = D; L.zero_noise = d.zero_noise;
        L.noise = d.noise; L.noise_out = d.noise_out; L.sample = d.sample; L.mean = d.mean; L.var = d.var;
        if (!d.noise && !d.zero_noise) need_rng = true;
        if (d.type == IWVI_LAYER_GP) {
            if (d.D != D) { set_error("iwvi_dgp_forward: layer %d expects D=%d but its input has %d columns", i, d.D, D); return IWVI_ERR_ARG; }
            if (!d.state) { set_error("iwvi_dgp_forward: layer %d has no precomputed state", i); return IWVI_ERR_ARG; }
            if (d.M <= 0 || d.M > IWVI_MAX_M || D > IWVI_MAX_D || d.R <= 0 || d.R > IWVI_MAX_R || d.P <= 0 || d.P > IWVI_MAX_P) {
                set_error("iwvi_gp_layer_forward: size out of range (M=%d D=%d R=%d P=%d)", d.M, D, d.R, d.P); return IWVI_ERR_ARG;
            }
            if (!d.W && d.P != d.R) { set_error("iwvi_gp_layer_forward: P=%d must equal R=%d without a mixing matrix", d.P, d.R); return IWVI_ERR_ARG; }
            if (d.mf_type == IWVI_MF_IDENTITY && d.P != D) { set_error("iwvi_gp_layer_forward: Identity mean function needs P == D (%d vs %d)", d.P, D); return IWVI_ERR_ARG; }
            if (d.mf_type == IWVI_MF_LINEAR && !d.mf_A) { set_error("iwvi_gp_layer_forward: Linear mean function without A"); return IWVI_ERR_ARG; }
            if (d.mf_type < IWVI_MF_ZERO || d.mf_type > IWVI_MF_LINEAR) { set_error("iwvi_gp_layer_forward: unknown mean function %d", d.mf_type); return IWVI_ERR_UNSUPPORTED; }
            if (d.kern_type != IWVI_KERN_RBF && d.kern_type != IWVI_KERN_MATERN52) { set_error("iwvi_gp_layer_forward: unknown kernel type %d", d.kern_type); return IWVI_ERR_UNSUPPORTED; }
            if ((d.a_out || d.u_out) && n_layers != 1) { set_error("iwvi_dgp_forward: a_out / u_out are single-layer outputs"); return IWVI_ERR_ARG; }
            const StateLayout s = state_layout(d.M, d.R);
            const char* st = (const char*)d.state;
            FwGp& G = L.gp;
            G.LinvP = (const f32x4*)(st + s.off_LinvP); G.LrTP = (const f32x4*)(st + s.off_LrTP);
            G.WqP = (const f32x4*)(st + s.off_WqP); G.ZtP = (const float*)(st + s.off_ZtP);
            G.zc = (const float*)(st + s.off_zc); G.invls = (const float*)(st + s.off_invls);
            G.W = d.W; G.mfA = d.mf_A; G.mfb = d.mf_b; G.a_out = d.a_out; G.u_out = d.u_out;
            G.M = d.M; G.Mp = s.Mp; G.nbk = s.nbk; G.nrb = s.nrb; G.nsteps = round_up(D + 2, 4) / 4;
            G.R = d.R; G.P = d.P; G.kern_type = d.kern_type; G.mf_type = d.mf_type; G.variance = d.variance;
            if (d.R > maxR) maxR = d.R;
            if (d.P > maxP) maxP = d.P;
            if (s.nbk > max_nbk) max_nbk = s.nbk;
            D = d.P;
        } else if (d.type == IWVI_LAYER_LV) {
            FwLv& V = L.lv;
            if (d.latent_dim <= 0 || D + d.latent_dim > IWVI_MAX_D) { set_error("iwvi_lv_layer_forward: bad D=%d (1..32) or latent_dim=%d", D, d.latent_dim); return IWVI_ERR_ARG; }
            V.Lw = d.latent_dim; V.sampled_kl = d.sampled_kl; V.kl_local = d.kl_local;
            V.n_enc = 0; V.wtotal = 0; V.maxdim = 2 * d.latent_dim;
            if (d.enc_W) {
                if (!XY || XYdim <= 0) { set_error("iwvi_dgp_forward: layer %d has an encoder but there are no encoder inputs", i); return IWVI_ERR_ARG; }
                if (!d.enc_dims || d.n_enc <= 0 || d.n_enc > IWVI_MAX_ENC) { set_error("iwvi_lv_layer_forward: encoder with %d layers (1..%d supported)", d.n_enc, IWVI_MAX_ENC); return IWVI_ERR_ARG; }
                if (d.enc_dims[0] != XYdim) { set_error("iwvi_lv_layer_forward: encoder expects %d inputs, XY has %d", d.enc_dims[0], XYdim); return IWVI_ERR_ARG; }
                if (d.enc_dims[d.n_enc] != 2 * d.latent_dim) { set_error("iwvi_lv_layer_forward: encoder output %d != 2*latent_dim %d", d.enc_dims[d.n_enc], 2 * d.latent_dim); return IWVI_ERR_ARG; }
                for (int k = 0; k <= d.n_enc; ++k) {
                    if (d.enc_dims[k] <= 0 || d.enc_dims[k] > 64) { set_error("iwvi_lv_layer_forward: encoder width %d out of range (1..64)", d.enc_dims[k]); return IWVI_ERR_ARG; }
                    V.dims[k] = d.enc_dims[k];
                    if (d.enc_dims[k] > V.maxdim) V.maxdim = d.enc_dims[k];
                }
                for (int k = 0; k < d.n_enc; ++k) {
                    if (!d.enc_W[k]) { set_error("iwvi_lv_layer_forward: null encoder weight %d", k); return IWVI_ERR_ARG; }
                    V.W[k] = d.enc_W[k]; V.b[k] = d.enc_b ? d.enc_b[k] : nullptr;
                    V.wtotal += d.enc_dims[k] * d.enc_dims[k + 1] + d.enc_dims[k + 1];
                }
                V.n_enc = d.n_enc;
            }
            if (D + d.latent_dim > maxP) maxP = D + d.latent_dim;
            D += d.latent_dim;
        } else { set_error("iwvi_dgp_forward: unknown layer type %d", d.type); return IWVI_ERR_ARG; }
    }
    if (out_logw) {
        if (layers[n_layers - 1].type != IWVI_LAYER_GP || D != Dy) { set_error("iwvi_dgp_forward: the last layer must be a GP layer with P == Dy (%d vs %d)", D, Dy); return IWVI_ERR_ARG; }
    }
    if (need_rng && !rng_state) { set_error("iwvi_dgp_forward: a layer draws its own noise but rng_state is NULL"); return IWVI_ERR_ARG; }
    // chunk size: the largest NS (16*NS samples per workgroup) whose LDS image fits, then no larger than
    // what gives every CU a workgroup
    int ns_max = 0; FwLds lay{};
    for (int ns = FW_MAXNS; ns >= 1; --ns) {
        const int nsamp = 16 * ns;
        int scratch = 0;
        for (int i = 0; i < n_layers; ++i) {
            const FwLayer& L = a.L[i];
            const int need = L.type == IWVI_LAYER_GP ? gp_scratch_floats(L.gp.Mp, L.gp.nbk, L.gp.R, nsamp)
                                                     : lv_scratch_floats(L.lv.wtotal, L.lv.maxdim, L.lv.Lw, nsamp);
            if (need > scratch) scratch = need;
        }
        lay = fw_lds_layout(nsamp, maxR, maxP, max_nbk, scratch);
        if ((size_t)lay.total * sizeof(float) <= 160 * 1024) { ns_max = ns; break; }
    }
    if (ns_max == 0) { set_error("iwvi_dgp_forward: the layer stack needs %zu B of LDS per 16 samples (> 160 KiB)", (size_t)lay.total * sizeof(float)); return IWVI_ERR_UNSUPPORTED; }
    int ns = (int)((T + 16 * 256 - 1) / (16 * 256));
    if (ns > ns_max) ns = ns_max;
    if (ns < 1) ns = 1;
    if (ns != ns_max) {
        const int nsamp = 16 * ns;
        int scratch = 0;
        for (int i = 0; i < n_layers; ++i) {
            const FwLayer& L = a.L[i];
            const int need = L.type == IWVI_LAYER_GP ? gp_scratch_floats(L.gp.Mp, L.gp.nbk, L.gp.R, nsamp)
                                                     : lv_scratch_floats(L.lv.wtotal, L.lv.maxdim, L.lv.Lw, nsamp);
            if (need > scratch) scratch = need;
        }
        lay = fw_lds_layout(nsamp, maxR, maxP, max_nbk, scratch);
    }
    a.lds = lay;
    const long long chunks = (T + 16 * ns - 1) / (16 * ns);
    if (chunks > 0x7fffffffLL) { set_error("iwvi_dgp_forward: T too large"); return IWVI_ERR_ARG; }
    const size_t lds_bytes = (size_t)lay.total * sizeof(float);
    switch (ns) {
        case 1: return launch_forward<1>(a, (unsigned)chunks, lds_bytes, stream);
        case 2: return launch_forward<2>(a, (unsigned)chunks, lds_bytes, stream);
        case 3: return launch_forward<3>(a, (unsigned)chunks, lds_bytes, stream);
        case 4: return launch_forward<4>(a, (unsigned)chunks, lds_bytes, stream);
        default: return launch_forward<5>(a, (unsigned)chunks, lds_bytes, stream);
    }
}

}  // namespace iwvi

using namespace iwvi;

extern "C" int iwvi_dgp_forward(const iwvi_layer_desc* layers, int n_layers, const float* X, int Dx,
                                const float* XY, int XYdim, const float* Y, int Dy, int64_t T, int64_t row_div,
                                int64_t row_mod, float lik_variance, uint64_t seed, uint64_t* rng_state,
                                float* out_logw, void* stream) {
    return dgp_forward_impl(layers, n_layers, X, Dx, XY, XYdim, Y, Dy, T, row_div, row_mod, lik_variance, seed,
                            rng_state, out_logw, (hipStream_t)stream);
}

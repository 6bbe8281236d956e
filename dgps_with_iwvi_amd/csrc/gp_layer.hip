// Fused sparse-GP layer forward for a flattened (minibatch x importance-sample) batch on gfx950.
//
// Replaces, per layer (reference file:line): Kuf (temp_workaround.py:44), matrix_triangular_solve (:51),
// Kdiag - sum A^2 (:59), A^T q_mu (:68), einsum('rMm,sMn->srmn') (:78), + sum LTA^2 (:85), the marginal
// sample (:89-91), the SharedMixedMok mixing (:142-145) and the mean-function add (layers.py:46-48).
// Nothing the reference materialises (Kmn, A, LTA: 10 MB ... 8 GB) ever reaches HBM.
//
// One 256-thread workgroup (4 waves, one per SIMD) owns a tile of 32 samples:
//   phase 1  Gram    : k = K(Zs, x/l) for the tile -> LDS, already in MFMA B-operand order
//   phase 2  stage 1 : a = Lm^-1 k  (lower-triangular blocks only) and mean = q_mu^T a as
//                      v_mfma_f32_32x32x2_f32 jobs; a -> LDS in B-operand order, |a|^2 per sample
//   phase 3  stage 2 : u_r = tril(q_sqrt_r)^T a (upper-triangular blocks only) -> |u_r|^2 per sample
//   phase 4  epilogue: var, sample, mixing, mean function -> [T, P] outputs
// Jobs (one 32-row output block each) are handed to the four waves from an LDS counter in order of
// decreasing cost, so the triangular imbalance is absorbed inside the workgroup; every job writes its
// partial sums to its own LDS slot, so results are bit-reproducible whatever wave ran it.
// A operands come straight from L2 in the pre-packed fragment order (one 1-KiB coalesced load per
// wave per four MFMAs); B operands are ds_read_b128 from the tile.
#include "iwvi_common.h"

namespace iwvi {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct LayerArgs {
    const float* LinvP; const float* LrTP; const float* QmuP; const float* Zs;
    const float* F; const float* noise; const float* W; const float* mfA; const float* mfb;
    float* sample; float* mean; float* var;
    float* a_out; float* u_out;          // optional [T, Mp] / [R, T, Mp] (full-covariance path only)
    long long T;
    int M, Mp, nb, D, R, P, kern_type, mf_type, bcast_K;
    float variance;
    const float* invls;                  // [32] 1/lengthscale (state buffer)
};

__device__ __forceinline__ float kern_value_f32(float r2, int type, float var) {
    if (type == IWVI_KERN_MATERN52) {
        const float s5 = 2.2360679774997896f;
        float r = sqrtf(r2 + 1e-12f);
        return var * (1.0f + s5 * r + (5.0f / 3.0f) * r * r) * __expf(-s5 * r);
    }
    return var * __expf(-0.5f * r2);
}

// acc += A(block row) * B for k-blocks [bk0, bk1).  Ablk points at packed block (bi, bk0).
__device__ __forceinline__ f32x16 mfma_job(const float* __restrict__ Ablk, int bk0, int bk1,
                                            const f32x4* __restrict__ Bt, int lane) {
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int j = lane & 31, h = lane >> 5;
    const f32x4* Ap = reinterpret_cast<const f32x4*>(Ablk) + lane;
    const f32x4* Bp = Bt + (size_t)(8 * bk0 + h) * 32 + j;
    for (int bk = bk0; bk < bk1; ++bk) {
        f32x4 a0 = Ap[0], a1 = Ap[64], a2 = Ap[128], a3 = Ap[192];
        f32x4 b0 = Bp[0], b1 = Bp[64], b2 = Bp[128], b3 = Bp[192];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, b2.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, b2.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.z, b2.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.w, b2.w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.x, b3.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.y, b3.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.z, b3.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.w, b3.w, acc, 0, 0, 0);
        Ap += 256;
        Bp += 256;
    }
    return acc;
}

__device__ __forceinline__ float sumsq16(const f32x16& v) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s = fmaf(v[i], v[i], s);
    return s;
}

extern __shared__ __attribute__((aligned(16))) unsigned char lsmem[];

// dynamic LDS carve (floats): kuf[Mp*32] | at[Mp*32] | xraw[32*32] | asq[nb*32] | usq[R*nb*32]
//                             | mean[nb*R*32] | counters[4]
__host__ __device__ static inline size_t layer_lds_floats(int Mp, int nb, int R) {
    return (size_t)Mp * 32 * 2 + 32 * 32 + (size_t)nb * 32 + (size_t)R * nb * 32 + (size_t)nb * R * 32 + 4;
}

template <int DP>
__global__ __launch_bounds__(256) void k_gp_layer(LayerArgs g) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int Mp = g.Mp, nb = g.nb, R = g.R, D = g.D;
    const long long t0 = (long long)blockIdx.x * 32;

    float* sm = reinterpret_cast<float*>(lsmem);
    f32x4* kuf = reinterpret_cast<f32x4*>(sm);
    f32x4* at = reinterpret_cast<f32x4*>(sm + (size_t)Mp * 32);
    float* xraw = sm + (size_t)Mp * 64;
    float* asq = xraw + 32 * 32;
    float* usq = asq + nb * 32;
    float* meanp = usq + (size_t)R * nb * 32;
    int* counters = reinterpret_cast<int*>(meanp + (size_t)nb * R * 32);

    // ---- phase 0: stage the raw input tile (rows beyond T are zero) --------------------------
    for (int idx = tid; idx < 32 * 32; idx += blockDim.x) {
        int jj = idx >> 5, d = idx & 31;
        long long t = t0 + jj;
        xraw[idx] = (t < g.T && d < D) ? g.F[(t / g.bcast_K) * D + d] : 0.f;
    }
    if (tid < 4) counters[tid] = 0;
    __syncthreads();

    // ---- phase 1: Gram tile, written in B-operand order: float4 (chunk c, sample j) = k[4c..4c+3][j]
    {
        float xs[DP];
#pragma unroll
        for (int d = 0; d < DP; ++d) xs[d] = (d < D) ? xraw[j * 32 + d] * g.invls[d] : 0.f;
        const int nchunk = Mp >> 2;
        for (int c = 2 * wave + h; c < nchunk; c += 2 * nw) {
            f32x4 kv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4* zr = reinterpret_cast<const f32x4*>(g.Zs + (size_t)(4 * c + e) * 32);
                float r2 = 0.f;
#pragma unroll
                for (int d4 = 0; d4 < DP / 4; ++d4) {
                    f32x4 z = zr[d4];
                    float t0_ = z.x - xs[4 * d4 + 0], t1_ = z.y - xs[4 * d4 + 1];
                    float t2_ = z.z - xs[4 * d4 + 2], t3_ = z.w - xs[4 * d4 + 3];
                    r2 = fmaf(t0_, t0_, r2); r2 = fmaf(t1_, t1_, r2);
                    r2 = fmaf(t2_, t2_, r2); r2 = fmaf(t3_, t3_, r2);
                }
                kv[e] = kern_value_f32(r2, g.kern_type, g.variance);
            }
            kuf[c * 32 + j] = kv;
        }
    }
    __syncthreads();

    // ---- phase 2: stage 1 jobs, decreasing cost: a-blocks nb-1 .. 0 (block bi sums k-blocks 0..bi) ----
    for (;;) {
        int job = 0;
        if (lane == 0) job = atomicAdd(&counters[0], 1);
        job = __builtin_amdgcn_readfirstlane(job);
        if (job >= nb) break;
        const int bi = nb - 1 - job;
        f32x16 acc = mfma_job(g.LinvP + (size_t)(bi * nb) * 1024, 0, bi + 1, kuf, lane);
        // accumulator regs 4q..4q+3 of lane (j,h) = rows 32bi + 8q + 4h + (0..3) of column j
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            at[(8 * bi + 2 * q + h) * 32 + j] = v;
            if (g.a_out) {
                long long t = t0 + j;
                if (t < g.T)
                    *reinterpret_cast<f32x4*>(g.a_out + t * Mp + 32 * bi + 8 * q + 4 * h) = v;
            }
        }
        float s = sumsq16(acc);
        s += __shfl_xor(s, 32);
        if (h == 0) asq[bi * 32 + j] = s;
    }
    __syncthreads();

    // ---- phase 3: stage 2 (u_r blocks) and mean_r = sum_m q_mu[m][r] a[m] (unit jobs over k-blocks).
    // Both read the finished a tile; one job list, decreasing cost:
    //   (i = 0..nb-1) x (r = 0..R-1) -> u block (r, i), cost nb - i ; then nb unit mean jobs.
    {
        const int nu = R * nb, njobs = nu + nb;
        for (;;) {
            int job = 0;
            if (lane == 0) job = atomicAdd(&counters[1], 1);
            job = __builtin_amdgcn_readfirstlane(job);
            if (job >= njobs) break;
            if (job < nu) {
                const int bi = job / R, r = job - bi * R;
                f32x16 acc = mfma_job(g.LrTP + ((size_t)r * nb * nb + (size_t)bi * nb + bi) * 1024,
                                      bi, nb, at, lane);
                if (g.u_out) {
                    long long t = t0 + j;
                    if (t < g.T) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                            *reinterpret_cast<f32x4*>(g.u_out + ((size_t)r * g.T + t) * Mp + 32 * bi + 8 * q + 4 * h) = v;
                        }
                    }
                }
                float s = sumsq16(acc);
                s += __shfl_xor(s, 32);
                if (h == 0) usq[(r * nb + bi) * 32 + j] = s;
            } else {
                const int bk = job - nu;
                f32x16 acc = mfma_job(g.QmuP + (size_t)bk * 1024, bk, bk + 1, at, lane);
                // row rho = (reg&3) + 8*(reg>>2) + 4h of the 32-row block holds latent GP rho
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    int rho = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    if (rho < R) meanp[(bk * R + rho) * 32 + j] = acc[reg];
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 4: epilogue, one thread per (sample, output) ---------------------------------
    const int P = g.P;
    for (int idx = tid; idx < 32 * P; idx += blockDim.x) {
        const int jj = idx / P, p = idx - jj * P;
        const long long t = t0 + jj;
        if (t >= g.T) continue;
        float a2 = 0.f;
        for (int i = 0; i < nb; ++i) a2 += asq[i * 32 + jj];
        const float base = g.variance - a2;
        float o_s = 0.f, o_m = 0.f, o_v = 0.f;
        for (int r = 0; r < R; ++r) {
            float w = 1.f;
            if (g.W) w = g.W[p * R + r];
            else if (r != p) continue;
            float u2 = 0.f, mu = 0.f;
            for (int i = 0; i < nb; ++i) { u2 += usq[(r * nb + i) * 32 + jj]; mu += meanp[(i * R + r) * 32 + jj]; }
            float v = fmaxf(base + u2, 0.f);
            float z = g.noise ? g.noise[t * R + r] : 0.f;
            float gs = fmaf(z, sqrtf(v), mu);
            o_s = fmaf(w, gs, o_s);
            o_m = fmaf(w, mu, o_m);
            o_v = fmaf(w * w, v, o_v);
        }
        float mf = 0.f;
        if (g.mf_type == IWVI_MF_IDENTITY) mf = xraw[jj * 32 + p];
        else if (g.mf_type == IWVI_MF_LINEAR) {
            for (int d = 0; d < D; ++d) mf = fmaf(xraw[jj * 32 + d], g.mfA[d * P + p], mf);
            if (g.mfb) mf += g.mfb[p];
        }
        if (g.sample) g.sample[t * P + p] = o_s + mf;
        if (g.mean) g.mean[t * P + p] = o_m + mf;
        if (g.var) g.var[t * P + p] = o_v;
    }
}

}  // namespace iwvi

namespace iwvi {

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

template <int DP>
static int launch_layer(const LayerArgs& g, size_t lds, hipStream_t stream) {
    static size_t attr_set = 0;     // largest LDS size this instantiation was configured for
    if (lds > attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gp_layer<DP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(k_gp_layer, %zu B): %s", lds, hipGetErrorString(e)); return IWVI_ERR_LAUNCH; }
        attr_set = lds;
    }
    long long tiles = (g.T + 31) / 32;
    hipLaunchKernelGGL(k_gp_layer<DP>, dim3((unsigned)tiles), dim3(256), lds, stream, g);
    return check_launch("k_gp_layer");
}

static int layer_forward_impl(const void* state, int M, int D, int R, int P, int kern_type, float variance,
                              const float* F, const float* noise, const float* W,
                              int mf_type, const float* mf_A, const float* mf_b,
                              float* sample, float* mean, float* var, float* a_out, float* u_out,
                              int64_t T, int bcast_K, hipStream_t stream) {
    if (T <= 0) return IWVI_OK;                         // empty batch: nothing to do
    if (!state || !F) { set_error("iwvi_gp_layer_forward: null state or input"); return IWVI_ERR_ARG; }
    if (M <= 0 || M > IWVI_MAX_M || D <= 0 || D > IWVI_MAX_D || R <= 0 || R > IWVI_MAX_R || P <= 0 || P > IWVI_MAX_P) {
        set_error("iwvi_gp_layer_forward: size out of range (M=%d D=%d R=%d P=%d)", M, D, R, P); return IWVI_ERR_ARG;
    }
    if (!W && P != R) { set_error("iwvi_gp_layer_forward: P=%d must equal R=%d without a mixing matrix", P, R); return IWVI_ERR_ARG; }
    if (mf_type == IWVI_MF_IDENTITY && P != D) { set_error("iwvi_gp_layer_forward: Identity mean function needs P == D (%d vs %d)", P, D); return IWVI_ERR_ARG; }
    if (mf_type == IWVI_MF_LINEAR && !mf_A) { set_error("iwvi_gp_layer_forward: Linear mean function without A"); return IWVI_ERR_ARG; }
    if (mf_type < IWVI_MF_ZERO || mf_type > IWVI_MF_LINEAR) { set_error("iwvi_gp_layer_forward: unknown mean function %d", mf_type); return IWVI_ERR_UNSUPPORTED; }
    if (kern_type != IWVI_KERN_RBF && kern_type != IWVI_KERN_MATERN52) { set_error("iwvi_gp_layer_forward: unknown kernel type %d", kern_type); return IWVI_ERR_UNSUPPORTED; }
    if (bcast_K < 1 || T % bcast_K != 0) { set_error("iwvi_gp_layer_forward: T=%lld is not a multiple of bcast_K=%d", (long long)T, bcast_K); return IWVI_ERR_ARG; }
    if ((T + 31) / 32 > 0x7fffffffLL) { set_error("iwvi_gp_layer_forward: T too large"); return IWVI_ERR_ARG; }
    StateLayout s = state_layout(M, R);
    const char* st = (const char*)state;
    LayerArgs g{};
    g.LinvP = (const float*)(st + s.off_LinvP); g.LrTP = (const float*)(st + s.off_LrTP);
    g.QmuP = (const float*)(st + s.off_QmuP); g.Zs = (const float*)(st + s.off_Zs);
    g.invls = (const float*)(st + s.off_invls);
    g.F = F; g.noise = noise; g.W = W; g.mfA = mf_A; g.mfb = mf_b;
    g.sample = sample; g.mean = mean; g.var = var; g.a_out = a_out; g.u_out = u_out;
    g.T = T; g.M = M; g.Mp = s.Mp; g.nb = s.nb; g.D = D; g.R = R; g.P = P;
    g.kern_type = kern_type; g.mf_type = mf_type; g.variance = variance; g.bcast_K = bcast_K;
    size_t lds = layer_lds_floats(s.Mp, s.nb, R) * sizeof(float);
    if (lds > 160 * 1024) { set_error("iwvi_gp_layer_forward: M=%d, R=%d needs %zu B of LDS (> 160 KiB)", M, R, lds); return IWVI_ERR_UNSUPPORTED; }
    if (D <= 4) return launch_layer<4>(g, lds, stream);
    if (D <= 8) return launch_layer<8>(g, lds, stream);
    if (D <= 12) return launch_layer<12>(g, lds, stream);
    if (D <= 16) return launch_layer<16>(g, lds, stream);
    if (D <= 24) return launch_layer<24>(g, lds, stream);
    return launch_layer<32>(g, lds, stream);
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
    size_t Mp = (size_t)round_up(M, 32);
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
    const size_t Mp = (size_t)round_up(M, 32);
    float* a = (float*)ws;
    float* u = a + (size_t)T * Mp;
    int rc = layer_forward_impl(state, M, D, R, R, kern_type, variance, F, nullptr, nullptr, IWVI_MF_ZERO,
                                nullptr, nullptr, nullptr, mean, nullptr, a, u, T, 1, stream);
    if (rc != IWVI_OK) return rc;
    StateLayout sl = state_layout(M, R);
    const float* invls = (const float*)((const char*)state + sl.off_invls);
    hipLaunchKernelGGL(k_fullcov, dim3((unsigned)S, (unsigned)R), dim3(256), 0, stream, F, invls,
                       (const float*)a, (const float*)u, cov, (long long)S, (int)N, D, (int)Mp, R, kern_type, variance);
    return check_launch("k_fullcov");
}

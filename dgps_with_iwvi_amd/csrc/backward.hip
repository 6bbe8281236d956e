// Backward pass of the IW-ELBO path (the reference obtains these gradients from
// TensorFlow's autodiff of the graph built by models.py:112-150, experiments/build_models.py:284-304).
// gfx950 only.  First, correct, version: layer by layer, intermediates in HBM, float32 MFMA products for everything
// that sums over samples, float64 for the Cholesky adjoint.  Not yet fused like the forward.
//
// One GP layer (temp_workaround.py:39-91 + :142-145 + layers.py:46-48), per sample t with a = Lm^-1 k(Z, x),
// u_r = L_r^T a, mu_r = a . q_mu_r, v_r = s2 - |a|^2 + |u_r|^2, g_r = mu_r + eps_r sqrt(v_r), f = W g + x A:
//   heads     dmu_r = W^T (df_s + df_m),  dv_r = (W*W)^T df_v + (W^T df_s)_r eps_r / (2 sqrt v_r)
//   da        = sum_r [ q_mu_r dmu_r + 2 dv_r (L_r u_r - a) ]                  (row GEMMs on U_r, L_r^T)
//   dk        = Lm^-T da                                                       (row GEMM with Lm^-1)
//   c_m       = -1/2 k_m dk_m ;  dx~ = 2 x~ sum_m c_m - 2 C Z~ ;  dZ~ = 2 Z~ o colsum(C) - 2 C^T X~
//   dq_mu     = A^T dMU ;  dL_r = tril(2 A^T diag(dv_r) U_r) ;  dLm = -tril(dK^T A)   (split-K GEMMs over samples)
//   dKuu      = 1/2 (S + S^T),  S = Lm^-T Phi(Lm^T dLm) Lm^-1                  (float64)
// and the closed-form gradient of the whitened KL (temp_workaround.py:186-188) with weight -kl_weight.
#include "iwvi_common.h"
#include <cmath>

namespace iwvi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------
// C[m, n] = alpha * sum_k s * A(m, k) * B(k, n) (+ beta * C), any strides; s = scale[k] or scale[m] (optional).
// 64x64 output tile per workgroup, 16-deep LDS stages, v_mfma_f32_16x16x4_f32: wave w owns rows 16w..16w+15.
// nsplit > 1: workgroup z sums k in [z*kchunk, (z+1)*kchunk) into part[z][M][N] (summed by k_reduce_parts, in
// float64, in a fixed order: the gradients are bit-reproducible).
// ------------------------------------------------------------------------------------------------------------
struct GemmArgs {
    const float* A; long long a_sm, a_sk;
    const float* B; long long b_sk, b_sn;
    const float* scale; long long s_stride; int scale_on_k;
    float* C; long long ldc;
    float* part;
    int M, N, K, nsplit, kchunk;
    float alpha, beta;
    int b_keep_n_ge_k;                  // B(k, n) read as 0 where n < k (a lower-triangular matrix indexed [n][k])
};
constexpr int GT = 64, GK = 16, GLD = GT + 4;

__global__ __launch_bounds__(256) void k_gemm(GemmArgs g) {
    __shared__ float As[GK][GLD];
    __shared__ float Bs[GK][GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    const int kb = (int)blockIdx.z * g.kchunk;
    const int ke = (kb + g.kchunk < g.K) ? kb + g.kchunk : g.K;
    const bool a_k_contig = g.a_sk == 1, b_n_contig = g.b_sn == 1;
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = kb; k0 < ke; k0 += GK) {
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            int m, k;
            if (a_k_contig) { m = idx >> 4; k = idx & 15; } else { k = idx >> 6; m = idx & 63; }
            const int gm = m0 + m, gk = k0 + k;
            float v = 0.f;
            if (gm < g.M && gk < ke) {
                v = g.A[gm * g.a_sm + gk * g.a_sk];
                if (g.scale) v *= g.scale[(g.scale_on_k ? gk : gm) * g.s_stride];
            }
            As[k][m] = v;
            int n;
            if (b_n_contig) { k = idx >> 6; n = idx & 63; } else { n = idx >> 4; k = idx & 15; }
            const int gn = n0 + n, gk2 = k0 + k;
            float w = 0.f;
            if (gn < g.N && gk2 < ke && (!g.b_keep_n_ge_k || gn >= gk2)) w = g.B[gk2 * g.b_sk + gn * g.b_sn];
            Bs[k][n] = w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float a = As[4 * kk + (lane >> 4)][16 * wave + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float b = Bs[4 * kk + (lane >> 4)][16 * j + (lane & 15)];
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    for (int j = 0; j < 4; ++j)
        for (int v = 0; v < 4; ++v) {
            const int m = m0 + 16 * wave + 4 * (lane >> 4) + v, n = n0 + 16 * j + (lane & 15);
            if (m >= g.M || n >= g.N) continue;
            if (g.nsplit > 1) g.part[((size_t)blockIdx.z * g.M + m) * g.N + n] = acc[j][v];
            else {
                float* c = g.C + m * g.ldc + n;
                *c = g.alpha * acc[j][v] + (g.beta != 0.f ? g.beta * *c : 0.f);
            }
        }
}

// out[m, n] = alpha * sum_s part[s][m][n] (+ beta * out); tri: entries above the diagonal become 0.  Either output
// may be null (float / double).
struct ReduceArgs { const float* part; int S, M, N; float* out; double* out64; long long ldo; double alpha, beta; int tri; };
__global__ void k_reduce_parts(ReduceArgs r) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= r.M * r.N) return;
    const int m = idx / r.N, n = idx - m * r.N;
    double s = 0.0;
    for (int k = 0; k < r.S; ++k) s += (double)r.part[(size_t)k * r.M * r.N + idx];
    s *= r.alpha;
    if (r.tri && n > m) s = 0.0;
    if (r.out) { float* o = r.out + m * r.ldo + n; *o = (float)(s + (r.beta != 0.0 ? r.beta * (double)*o : 0.0)); }
    if (r.out64) { double* o = r.out64 + m * r.ldo + n; *o = s + (r.beta != 0.0 ? r.beta * *o : 0.0); }
}

static int gemm(hipStream_t st, GemmArgs g, float* part_ws, size_t part_floats, float* out, double* out64, long long ldo,
                double alpha, double beta, int tri) {
    // split-K form: C / alpha / beta of `g` unused; the reduction writes out / out64
    const int kchunk = 512;
    g.nsplit = (g.K + kchunk - 1) / kchunk; g.kchunk = kchunk;
    if (g.nsplit < 2) { g.nsplit = 2; g.kchunk = round_up((g.K + 1) / 2, GK); if (g.kchunk < GK) g.kchunk = GK; }
    if ((size_t)g.nsplit * g.M * g.N > part_floats) { set_error("backward: split-K workspace too small"); return IWVI_ERR_ARG; }
    g.part = part_ws;
    hipLaunchKernelGGL(k_gemm, dim3((g.N + GT - 1) / GT, (g.M + GT - 1) / GT, g.nsplit), dim3(256), 0, st, g);
    ReduceArgs r{part_ws, g.nsplit, g.M, g.N, out, out64, ldo, alpha, beta, tri};
    hipLaunchKernelGGL(k_reduce_parts, dim3((g.M * g.N + 255) / 256), dim3(256), 0, st, r);
    return check_launch("k_gemm (split-K)");
}
static int gemm_rows(hipStream_t st, GemmArgs g) {     // many rows, short K: direct store
    g.nsplit = 1; g.kchunk = round_up(g.K, GK); g.part = nullptr;
    hipLaunchKernelGGL(k_gemm, dim3((g.N + GT - 1) / GT, (g.M + GT - 1) / GT, 1), dim3(256), 0, st, g);
    return check_launch("k_gemm (rows)");
}

// ------------------------------------------------------------------------------------------------------------
// per-sample heads: one wave per sample
// ------------------------------------------------------------------------------------------------------------
struct HeadArgs {
    const float* A; const float* U; const float* eps; const float* W; const float* mfA;
    const float* dFs; const float* dFm; const float* dFv;
    float* DMU; float* DV2; float* SDV; float* dF;
    long long T; int M, Mp, D, R, P, mf_type; float variance;
};
__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(256) void k_bw_heads(HeadArgs h) {
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= h.T) return;
    const float* a = h.A + t * h.Mp;
    float aa = 0.f;
    for (int m = lane; m < h.M; m += 64) aa = fmaf(a[m], a[m], aa);
    aa = wave_sum(aa);
    float mine_dv = 0.f, mine_dmu = 0.f;                  // lane r keeps latent r's heads
    for (int r = 0; r < h.R; ++r) {
        const float* u = h.U + ((size_t)r * h.T + t) * h.Mp;
        float uu = 0.f;
        for (int m = lane; m < h.M; m += 64) uu = fmaf(u[m], u[m], uu);
        uu = wave_sum(uu);
        float dg = 0.f, dm = 0.f, dvv = 0.f;
        if (h.W) {
            for (int p = 0; p < h.P; ++p) {
                const float w = h.W[p * h.R + r];
                if (h.dFs) dg = fmaf(w, h.dFs[t * h.P + p], dg);
                if (h.dFm) dm = fmaf(w, h.dFm[t * h.P + p], dm);
                if (h.dFv) dvv = fmaf(w * w, h.dFv[t * h.P + p], dvv);
            }
        } else {
            if (h.dFs) dg = h.dFs[t * h.P + r];
            if (h.dFm) dm = h.dFm[t * h.P + r];
            if (h.dFv) dvv = h.dFv[t * h.P + r];
        }
        const float v = h.variance - aa + uu;
        float dv = dvv;
        if (v > 0.f) { if (h.eps) dv += dg * h.eps[t * h.R + r] * 0.5f / sqrtf(v); } else dv = 0.f;   // the forward clamps v at 0
        if (lane == r) { mine_dv = dv; mine_dmu = dg + dm; }
    }
    if (lane < h.R) { h.DMU[t * h.R + lane] = mine_dmu; h.DV2[t * h.R + lane] = 2.f * mine_dv; }
    const float sdv = wave_sum(lane < h.R ? mine_dv : 0.f);
    if (lane == 0) h.SDV[t] = sdv;
    // the mean function's share of dF (layers.py:46-48: added to the samples and to the mean)
    if (h.dF) for (int d = lane; d < h.D; d += 64) {
        float acc = 0.f;
        if (h.mf_type == IWVI_MF_LINEAR) {
            for (int p = 0; p < h.P; ++p) {
                float up = 0.f;
                if (h.dFs) up += h.dFs[t * h.P + p];
                if (h.dFm) up += h.dFm[t * h.P + p];
                acc = fmaf(h.mfA[d * h.P + p], up, acc);
            }
        } else if (h.mf_type == IWVI_MF_IDENTITY) {
            if (h.dFs) acc += h.dFs[t * h.P + d];
            if (h.dFm) acc += h.dFm[t * h.P + d];
        }
        h.dF[t * h.D + d] = acc;
    }
}

// DA[t, m] -= 2 * SDV[t] * A[t, m]
__global__ void k_bw_axpy(float* DA, const float* A, const float* SDV, long long T, int M, int Mp) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * M) return;
    const long long t = idx / M; const int m = (int)(idx - t * M);
    DA[idx] = fmaf(-2.f * SDV[t], A[t * Mp + m], DA[idx]);
}

// K_uf entries again (RBF, direct differences), c = -1/2 k dk written over DA; per-sample sum_m c and sum_m k dk
struct KernArgs { const float* F; const float* Zt; const float* invls; const float* DK; float* C; float* RS; long long T; int M, D; float variance; };
__global__ __launch_bounds__(256) void k_bw_kernel(KernArgs a) {
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= a.T) return;
    float xt[IWVI_MAX_D];
    for (int d = 0; d < a.D; ++d) xt[d] = a.F[t * a.D + d] * a.invls[d];
    float sc = 0.f, skd = 0.f;
    for (int m = lane; m < a.M; m += 64) {
        float d2 = 0.f;
        for (int d = 0; d < a.D; ++d) { const float e = xt[d] - a.Zt[m * a.D + d]; d2 = fmaf(e, e, d2); }
        const float k = a.variance * __expf(-0.5f * d2);
        const float kd = k * a.DK[t * a.M + m];
        a.C[t * a.M + m] = -0.5f * kd;
        sc += -0.5f * kd; skd += kd;
    }
    sc = wave_sum(sc); skd = wave_sum(skd);
    if (lane == 0) { a.RS[2 * t] = sc; a.RS[2 * t + 1] = skd; }
}

// dx~ = 2 x~ rowsum(C) - 2 C Z~ ;  dF += dx~ * invls ;  Q = dx~ o x (its column sums are d/d invls through x~)
__global__ void k_bw_dx(const float* F, const float* invls, const float* RS, const float* CZ, float* dF, float* Q, long long T, int D) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * D) return;
    const long long t = idx / D; const int d = (int)(idx - t * D);
    const float x = F[idx], il = invls[d];
    const float dxt = 2.f * (x * il) * RS[2 * t] - 2.f * CZ[idx];
    if (dF) dF[idx] = fmaf(dxt, il, dF[idx]);
    Q[idx] = dxt * x;
}

// ------------------------------------------------------------------------------------------------------------
// float64 side: small dense products for the Cholesky adjoint, K_uu's own gradient, final assembly
// ------------------------------------------------------------------------------------------------------------
// C[i, j] = sum_k A(i, k) B(k, j); post = 1: Phi (strict upper -> 0, diagonal halved)
__global__ void k_dmm(const double* A, long long a_si, long long a_sk, const double* B, long long b_sk, long long b_sj,
                      double* C, int n, int ldc, int post) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= n) return;
    double s = 0.0;
    for (int k = 0; k < n; ++k) s = fma(A[i * a_si + k * a_sk], B[k * b_sk + j * b_sj], s);
    if (post == 1) s = (j > i) ? 0.0 : (j == i ? 0.5 * s : s);
    C[(size_t)i * ldc + j] = s;
}
__global__ void k_prep(const float* Z, const float* ls, const double* Linv64, int Mp, float* Zt, float* invls, float* LinvF, int M, int D) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < D) invls[idx] = 1.f / ls[idx];
    if (idx < M * D) Zt[idx] = Z[idx] / ls[idx % D];
    if (idx < M * M) { const int i = idx / M, j = idx - i * M; LinvF[idx] = j <= i ? (float)Linv64[(size_t)i * Mp + j] : 0.f; }
}
// row m of K_uu: dZ~_uu[m, :] = -2 sum_n Sbar_mn K_mn (z~_m - z~_n),  dvar_m = sum_n Sbar_mn K_mn / s2,  Sbar = (S + S^T)/2
__global__ __launch_bounds__(256) void k_kuu_bwd(const float* Zt, const double* S, int M, int D, double variance, double* dZt_uu, double* dvar_m) {
    __shared__ double red[256];
    const int m = blockIdx.x, tid = threadIdx.x;
    double acc[IWVI_MAX_D + 1];
    for (int d = 0; d <= D; ++d) acc[d] = 0.0;
    for (int n = tid; n < M; n += 256) {
        double d2 = 0.0;
        for (int d = 0; d < D; ++d) { const double e = (double)Zt[m * D + d] - (double)Zt[n * D + d]; d2 += e * e; }
        const double k = variance * exp(-0.5 * d2);
        const double sk = 0.5 * (S[(size_t)m * M + n] + S[(size_t)n * M + m]) * k;
        for (int d = 0; d < D; ++d) acc[d] += -2.0 * sk * ((double)Zt[m * D + d] - (double)Zt[n * D + d]);
        acc[D] += sk / variance;
    }
    for (int d = 0; d <= D; ++d) {
        red[tid] = acc[d];
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
        if (tid == 0) { if (d < D) dZt_uu[m * D + d] = red[0]; else dvar_m[m] = red[0]; }
        __syncthreads();
    }
}
struct FinalArgsB {
    const float* Z; const float* ls; const float* q_mu; const float* q_sqrt; const float* Zt; const float* invls;
    const float* colsumC; const float* CtF; const float* sums; const float* dinvls_x; const double* dZt_uu; const double* dvar_m;
    float* dZ; float* dls; float* dvariance; float* dq_mu; float* dq_sqrt;
    int M, D, R; double kl_weight, variance;
};
// one workgroup: dZ, dls, dvariance and the KL terms
__global__ __launch_bounds__(256) void k_bw_final(FinalArgsB f) {
    __shared__ double dil[IWVI_MAX_D];
    const int tid = threadIdx.x;
    if (tid < f.D) dil[tid] = 0.0;
    __syncthreads();
    if (tid < f.D) {                                        // thread d walks its column: fixed order
        const int d = tid;
        double s = (double)f.dinvls_x[d];
        for (int m = 0; m < f.M; ++m) {
            const double zt = f.Zt[m * f.D + d];
            const double dzt = 2.0 * zt * (double)f.colsumC[m] - 2.0 * (double)f.invls[d] * (double)f.CtF[m * f.D + d] + f.dZt_uu[m * f.D + d];
            if (f.dZ) f.dZ[m * f.D + d] = (float)(dzt * (double)f.invls[d]);
            s += dzt * (double)f.Z[m * f.D + d];
        }
        if (f.dls) f.dls[d] = (float)(-s * (double)f.invls[d] * (double)f.invls[d]);
    }
    if (tid == 0 && f.dvariance) {
        double s = (double)f.sums[0] + (double)f.sums[1] / f.variance;     // sum_t sum_r dv_r  +  sum k dk / s2
        for (int m = 0; m < f.M; ++m) s += f.dvar_m[m];
        f.dvariance[0] = (float)s;
    }
    // - kl_weight * d KL: KL = 1/2 (sum q_mu^2 - R M - sum log L_ii^2 + sum L^2)   (temp_workaround.py:186-188)
    if (f.dq_mu) for (int i = tid; i < f.M * f.R; i += 256) f.dq_mu[i] = (float)((double)f.dq_mu[i] - f.kl_weight * (double)f.q_mu[i]);
    if (f.dq_sqrt) for (long long i = tid; i < (long long)f.R * f.M * f.M; i += 256) {
        const int c = (int)(i % f.M), r_ = (int)((i / f.M) % f.M);
        if (c > r_) { f.dq_sqrt[i] = 0.f; continue; }
        const double L = f.q_sqrt[i];
        f.dq_sqrt[i] = (float)((double)f.dq_sqrt[i] - f.kl_weight * (L - (c == r_ ? 1.0 / L : 0.0)));
    }
}

struct BwdWs {
    float *DMU, *DV2, *SDV, *DA, *DK, *CZ, *Q, *RS, *part, *LinvF, *Zt, *invls, *colsumC, *CtF, *sums, *dinvls_x, *one;
    double *Lbar, *T1, *T2, *S, *dZt_uu, *dvar_m;
    size_t part_floats, bytes;
};
static BwdWs bwd_layout(char* base, long long T, int M, int D, int R) {
    BwdWs w; size_t o = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o = align256(o + bytes); return p; };
    const int nsplit = (int)((T + 511) / 512) + 2;
    w.DMU = (float*)take(sizeof(float) * T * R); w.DV2 = (float*)take(sizeof(float) * T * R); w.SDV = (float*)take(sizeof(float) * T);
    w.DA = (float*)take(sizeof(float) * T * M); w.DK = (float*)take(sizeof(float) * T * M);
    w.CZ = (float*)take(sizeof(float) * T * D); w.Q = (float*)take(sizeof(float) * T * D); w.RS = (float*)take(sizeof(float) * T * 2);
    w.part_floats = (size_t)nsplit * M * M;
    w.part = (float*)take(sizeof(float) * w.part_floats);
    w.LinvF = (float*)take(sizeof(float) * M * M); w.Zt = (float*)take(sizeof(float) * M * D); w.invls = (float*)take(sizeof(float) * IWVI_MAX_D);
    w.colsumC = (float*)take(sizeof(float) * M); w.CtF = (float*)take(sizeof(float) * M * D); w.sums = (float*)take(sizeof(float) * 4);
    w.dinvls_x = (float*)take(sizeof(float) * IWVI_MAX_D); w.one = (float*)take(sizeof(float) * 4);
    w.Lbar = (double*)take(sizeof(double) * M * M); w.T1 = (double*)take(sizeof(double) * M * M); w.T2 = (double*)take(sizeof(double) * M * M);
    w.S = (double*)take(sizeof(double) * M * M); w.dZt_uu = (double*)take(sizeof(double) * M * D); w.dvar_m = (double*)take(sizeof(double) * M);
    w.bytes = o;
    return w;
}
__global__ void k_set_one(float* p) { if (threadIdx.x < 4) p[threadIdx.x] = 1.f; }


// ------------------------------------------------------------------------------------------------------------
// ELBO tail (models.py:134-150), one thread per data point: L_nk, softmax over the K samples, heads of the final layer
// ------------------------------------------------------------------------------------------------------------
struct ElboBwdArgs {
    const float* fmean; const float* fvar; const float* Y; int Dy;
    const float* kl[IWVI_MAX_KL]; int kl_dims[IWVI_MAX_KL]; int n_kl;
    long long B; int K; float lik_var; double scale;
    float* w; float* d_mean; float* d_var; double* part;   // part[0..B) = lse - log K, part[B..2B) = d lik_var share
};
__global__ void k_elbo_bwd(ElboBwdArgs a) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const float s = a.lik_var, c0 = -0.5f * logf(6.283185307179586f * s);
    auto logw = [&](long long t) {
        float l = 0.f;
        for (int j = 0; j < a.Dy; ++j) {
            const float e = a.Y[b * a.Dy + j] - a.fmean[t * a.Dy + j];
            l += c0 - 0.5f * (e * e + a.fvar[t * a.Dy + j]) / s;
        }
        for (int i = 0; i < a.n_kl; ++i)
            for (int q = 0; q < a.kl_dims[i]; ++q) l -= a.kl[i][t * a.kl_dims[i] + q];
        return l;
    };
    float mx = -INFINITY;
    for (int k = 0; k < a.K; ++k) mx = fmaxf(mx, logw(b * a.K + k));
    double se = 0.0;
    for (int k = 0; k < a.K; ++k) se += (double)__expf(logw(b * a.K + k) - mx);
    a.part[b] = (double)mx + log(se) - log((double)a.K);
    double ds = 0.0;
    for (int k = 0; k < a.K; ++k) {
        const long long t = b * a.K + k;
        const float wt = (float)(a.scale * (double)__expf(logw(t) - mx) / se);
        if (a.w) a.w[t] = wt;
        for (int j = 0; j < a.Dy; ++j) {
            const float e = a.Y[b * a.Dy + j] - a.fmean[t * a.Dy + j], v = a.fvar[t * a.Dy + j];
            if (a.d_mean) a.d_mean[t * a.Dy + j] = wt * e / s;
            if (a.d_var) a.d_var[t * a.Dy + j] = -0.5f * wt / s;
            ds += (double)wt * (-0.5 / (double)s + 0.5 * ((double)e * e + (double)v) / ((double)s * s));
        }
    }
    a.part[a.B + b] = ds;
}
// out[i] = sum of part[i*n .. (i+1)*n), one workgroup per i, fixed order
__global__ __launch_bounds__(256) void k_dsum(const double* part, long long n, double* out) {
    __shared__ double red[256];
    const double* p = part + (size_t)blockIdx.x * n;
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += 256) s += p[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// LatentVariableLayer (layers.py:83-103): W = mu + eps sigma; d(enc_out) [B, 2 Lw] = sum over the K samples of (dmu | draw)
struct LvBwdArgs {
    const float* mu; const float* sigma; const float* eps; const float* dFn; int ld, col0;
    const float* w; int Lw; long long B; int K, sampled; float* d_out;
};
__global__ void k_lv_bwd(LvBwdArgs a) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.B * a.Lw) return;
    const long long b = idx / a.Lw; const int l = (int)(idx - b * a.Lw);
    const float mu = a.mu[idx], sg = a.sigma[idx];
    float dmu = 0.f, dsg = 0.f;
    for (int k = 0; k < a.K; ++k) {
        const long long t = b * a.K + k;
        const float e = a.eps[t * a.Lw + l], W = fmaf(e, sg, mu);
        const float dfw = a.dFn ? a.dFn[t * a.ld + a.col0 + l] : 0.f;
        const float dkl = a.w ? -a.w[t] : 0.f;                 // L_nk contains -kl (models.py:141-142)
        if (a.sampled) { const float dW = dfw + dkl * W; dmu += dW; dsg += dW * e - dkl / sg; }
        else { dmu += dfw + dkl * mu; dsg += dfw * e + dkl * (sg - 1.f / sg); }
    }
    a.d_out[b * 2 * a.Lw + l] = dmu;
    a.d_out[b * 2 * a.Lw + a.Lw + l] = dsg * (1.f - __expf(-sg));      // sigma = softplus(raw - 3): d sigma / d raw = 1 - exp(-sigma)
}

// Encoder MLP (layers.py:137-152), one thread per row: activations of every layer -> acts, then deltas (d / d pre-activation)
struct EncBwdArgs {
    const float* XY; long long rows; const float* W[IWVI_MAX_ENC]; const float* b[IWVI_MAX_ENC]; int dims[IWVI_MAX_ENC + 1]; int n;
    const float* d_out; float* acts[IWVI_MAX_ENC + 1]; float* delta[IWVI_MAX_ENC];
};
__global__ void k_enc_bwd(EncBwdArgs a) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.rows) return;
    for (int i = 0; i < a.dims[0]; ++i) a.acts[0][r * a.dims[0] + i] = a.XY[r * a.dims[0] + i];
    for (int l = 0; l < a.n; ++l) {
        const int din = a.dims[l], dout = a.dims[l + 1];
        const float* in = a.acts[l] + r * din;
        for (int o = 0; o < dout; ++o) {
            float acc = a.b[l] ? a.b[l][o] : 0.f;
            for (int i = 0; i < din; ++i) acc = fmaf(in[i], a.W[l][i * dout + o], acc);
            if (l < a.n - 1) acc = tanhf(acc);
            if (din == dout) acc += in[o];
            a.acts[l + 1][r * dout + o] = acc;
        }
    }
    float cur[64], prev[64];
    const int dl = a.dims[a.n];
    for (int o = 0; o < dl; ++o) cur[o] = a.d_out[r * dl + o];
    for (int l = a.n - 1; l >= 0; --l) {
        const int din = a.dims[l], dout = a.dims[l + 1];
        const bool skip = din == dout;
        for (int i = 0; i < din; ++i) prev[i] = skip ? cur[i] : 0.f;
        for (int o = 0; o < dout; ++o) {
            float dlin = cur[o];
            if (l < a.n - 1) {
                const float act = a.acts[l + 1][r * dout + o] - (skip ? a.acts[l][r * din + o] : 0.f);
                dlin *= 1.f - act * act;
            }
            a.delta[l][r * dout + o] = dlin;
            for (int i = 0; i < din; ++i) prev[i] = fmaf(dlin, a.W[l][i * dout + o], prev[i]);
        }
        for (int i = 0; i < din; ++i) cur[i] = prev[i];
    }
}


// ------------------------------------------------------------------------------------------------------------
// Optimiser steps of experiments/build_models.py:284-304: natural gradient on the final layer's (q_mu, q_sqrt)
// (GPflow NatGradOptimizer, natural parameterisation) and Adam on everything else (TensorFlow AdamOptimizer on
// GPflow's unconstrained variables).  float64 for the natural-gradient algebra, like the reference.
// ------------------------------------------------------------------------------------------------------------
// C[i, j] = alpha * sum_k A(i, k) B(k, j) + beta * E[i, j];  post 1: Phi, post 2: nothing
struct DmmArgs { const double* A; long long a_si, a_sk; const double* B; long long b_sk, b_sj; double* C; long long ldc;
                 int I, J, K; double alpha; const double* E; long long lde; double beta; int post; };
__global__ void k_dmm2(DmmArgs a) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= a.J) return;
    double s = 0.0;
    for (int k = 0; k < a.K; ++k) s = fma(a.A[i * a.a_si + k * a.a_sk], a.B[k * a.b_sk + j * a.b_sj], s);
    s *= a.alpha;
    if (a.E) s += a.beta * a.E[i * a.lde + j];
    if (a.post == 1) s = (j > i) ? 0.0 : (j == i ? 0.5 * s : s);
    a.C[i * a.ldc + j] = s;
}
static void dmm(hipStream_t st, const double* A, long long a_si, long long a_sk, const double* B, long long b_sk, long long b_sj,
                double* C, long long ldc, int I, int J, int K, double alpha = 1.0, const double* E = nullptr, long long lde = 0, double beta = 0.0, int post = 0) {
    DmmArgs a{A, a_si, a_sk, B, b_sk, b_sj, C, ldc, I, J, K, alpha, E, lde, beta, post};
    hipLaunchKernelGGL(k_dmm2, dim3((J + 127) / 128, I), dim3(J < 128 ? 64 : 128), 0, st, a);
}
// X = L^-1 for lower-triangular L [n, n] (row-major): thread j solves L x = e_j by forward substitution; zeros above
__global__ void k_tri_inv(const double* L, double* X, int n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    for (int i = 0; i < j; ++i) X[(size_t)i * n + j] = 0.0;
    for (int i = j; i < n; ++i) {
        double s = (i == j) ? 1.0 : 0.0;
        for (int k = j; k < i; ++k) s = fma(-L[(size_t)i * n + k], X[(size_t)k * n + j], s);
        X[(size_t)i * n + j] = s / L[(size_t)i * n + i];
    }
}
__global__ void k_f2d(const float* src, long long ld, double* dst, int rows, int cols, double scale, int tril) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const int i = idx / cols, j = idx - i * cols;
    dst[idx] = (tril && j > i) ? 0.0 : scale * (double)src[i * ld + j];
}
__global__ void k_d2f(const double* src, float* dst, long long ld, int rows, int cols, int tril) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const int i = idx / cols, j = idx - i * cols;
    dst[i * ld + j] = (tril && j > i) ? 0.f : (float)src[idx];
}
__global__ void k_axpby(const double* x, double a, const double* y, double b, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a * x[i] + b * y[i];
}
__global__ void k_symmetrise(double* Q, int n) {       // Q <- (Q + Q^T) / 2, one thread per (i >= j)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    const int i = idx / n, j = idx - i * n;
    if (j > i) return;
    const double v = 0.5 * (Q[(size_t)i * n + j] + Q[(size_t)j * n + i]);
    Q[(size_t)i * n + j] = v; Q[(size_t)j * n + i] = v;
}

// Adam on GPflow's unconstrained variables.  transform 1 = positive: p = softplus(x) + 1e-6 (gpflow.transforms.Log1pe)
struct AdamTensor { float* p; const float* g; float* x; float* m; float* v; long long n; int transform; };
constexpr int ADAM_MAX = 48;
struct AdamArgs { AdamTensor t[ADAM_MAX]; int n; float lr_t, b1, b2, eps, sign; int init; };
__global__ void k_adam(AdamArgs a) {
    const AdamTensor& T = a.t[blockIdx.y];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < T.n; i += (long long)gridDim.x * blockDim.x) {
        if (a.init) {
            const float p = T.p[i];
            float x = p;
            if (T.transform == 1) { const float y = p - 1e-6f; x = y > 20.f ? y : logf(expm1f(y)); }
            T.x[i] = x; T.m[i] = 0.f; T.v[i] = 0.f;
            continue;
        }
        float x = T.x[i], g = a.sign * T.g[i];
        if (T.transform == 1) g *= 1.f - __expf(-(T.p[i] - 1e-6f));          // d softplus(x) / dx = sigmoid(x)
        const float m = a.b1 * T.m[i] + (1.f - a.b1) * g;
        const float v = a.b2 * T.v[i] + (1.f - a.b2) * g * g;
        x -= a.lr_t * m / (sqrtf(v) + a.eps);
        T.m[i] = m; T.v[i] = v; T.x[i] = x;
        T.p[i] = (T.transform == 1) ? (x > 20.f ? x : log1pf(__expf(x))) + 1e-6f : x;
    }
}

}  // namespace iwvi

using namespace iwvi;

extern "C" size_t iwvi_gp_layer_backward_ws_bytes(int64_t T, int M, int D, int R) {
    if (T <= 0 || M <= 0 || D <= 0 || R <= 0) return 0;
    return bwd_layout(nullptr, T, M, D, R).bytes;
}

extern "C" int iwvi_gp_layer_backward(const iwvi_gp_bwd_desc* dp, int64_t T, void* ws_, void* stream_) {
    if (!dp || !ws_ || T <= 0) { set_error("iwvi_gp_layer_backward: bad argument"); return IWVI_ERR_ARG; }
    const iwvi_gp_bwd_desc& d = *dp;
    if (!d.state || !d.Z || !d.lengthscales || !d.q_mu || !d.q_sqrt || !d.F || !d.A || !d.U) { set_error("iwvi_gp_layer_backward: null input"); return IWVI_ERR_ARG; }
    if (d.M <= 0 || d.M > IWVI_MAX_M || d.D <= 0 || d.D > IWVI_MAX_D || d.R <= 0 || d.R > IWVI_MAX_R || d.P <= 0 || d.P > IWVI_MAX_P || T >= (1LL << 31) / (d.M > d.D ? d.M : d.D)) {
        set_error("iwvi_gp_layer_backward: size out of range"); return IWVI_ERR_ARG;
    }
    if (d.kern_type != IWVI_KERN_RBF) { set_error("iwvi_gp_layer_backward: only the RBF kernel has a backward pass so far"); return IWVI_ERR_UNSUPPORTED; }
    if (!d.W && d.P != d.R) { set_error("iwvi_gp_layer_backward: P != R without a mixing matrix"); return IWVI_ERR_ARG; }
    if (d.mf_type == IWVI_MF_LINEAR && !d.mf_A) { set_error("iwvi_gp_layer_backward: linear mean function without A"); return IWVI_ERR_ARG; }
    if (d.mf_type == IWVI_MF_IDENTITY && d.P != d.D) { set_error("iwvi_gp_layer_backward: identity mean function needs P == D"); return IWVI_ERR_ARG; }
    if (d.d_sample && !d.noise) { set_error("iwvi_gp_layer_backward: d_sample needs the forward's draws"); return IWVI_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream_;
    const int M = d.M, D = d.D, R = d.R;
    const StateLayout sl = state_layout(M, R);
    const int Mp = sl.Mp;
    const double* Lm64 = (const double*)((const char*)d.state + sl.off_Lm);
    const double* Linv64 = (const double*)((const char*)d.state + sl.off_Linv);
    BwdWs w = bwd_layout((char*)ws_, T, M, D, R);
    int rc;
    hipLaunchKernelGGL(k_set_one, dim3(1), dim3(64), 0, st, w.one);
    {
        const int n = M * M > M * D ? M * M : M * D;
        hipLaunchKernelGGL(k_prep, dim3((n + 255) / 256), dim3(256), 0, st, d.Z, d.lengthscales, Linv64, Mp, w.Zt, w.invls, w.LinvF, M, D);
    }
    HeadArgs h{d.A, d.U, d.noise, d.W, d.mf_A, d.d_sample, d.d_mean, d.d_var, w.DMU, w.DV2, w.SDV, d.dF, T, M, Mp, D, R, d.P, d.mf_type, d.variance};
    hipLaunchKernelGGL(k_bw_heads, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, st, h);
    if ((rc = check_launch("k_bw_heads")) != IWVI_OK) return rc;
    // DA = DMU q_mu^T
    GemmArgs g{};
    g.A = w.DMU; g.a_sm = R; g.a_sk = 1; g.B = d.q_mu; g.b_sk = 1; g.b_sn = R; g.C = w.DA; g.ldc = M; g.M = (int)T; g.N = M; g.K = R; g.alpha = 1.f; g.beta = 0.f;
    if ((rc = gemm_rows(st, g)) != IWVI_OK) return rc;
    // DA += (2 dv_r) o (U_r L_r^T)
    for (int r = 0; r < R; ++r) {
        GemmArgs q{};
        q.A = d.U + (size_t)r * T * Mp; q.a_sm = Mp; q.a_sk = 1;
        q.B = d.q_sqrt + (size_t)r * M * M; q.b_sk = 1; q.b_sn = M; q.b_keep_n_ge_k = 1;      // B(k = j, n = i) = L_r[i][j], i >= j
        q.scale = w.DV2 + r; q.s_stride = R; q.scale_on_k = 0;
        q.C = w.DA; q.ldc = M; q.M = (int)T; q.N = M; q.K = M; q.alpha = 1.f; q.beta = 1.f;
        if ((rc = gemm_rows(st, q)) != IWVI_OK) return rc;
    }
    hipLaunchKernelGGL(k_bw_axpy, dim3((unsigned)((T * M + 255) / 256)), dim3(256), 0, st, w.DA, d.A, w.SDV, (long long)T, M, Mp);
    // DK = DA Lm^-1
    {
        GemmArgs q{};
        q.A = w.DA; q.a_sm = M; q.a_sk = 1; q.B = w.LinvF; q.b_sk = M; q.b_sn = 1;
        q.C = w.DK; q.ldc = M; q.M = (int)T; q.N = M; q.K = M; q.alpha = 1.f; q.beta = 0.f;
        if ((rc = gemm_rows(st, q)) != IWVI_OK) return rc;
    }
    // dLm = -tril(DK^T A)  (float64 copy for the adjoint of the factorisation)
    {
        GemmArgs q{};
        q.A = w.DK; q.a_sm = 1; q.a_sk = M; q.B = d.A; q.b_sk = Mp; q.b_sn = 1; q.M = M; q.N = M; q.K = (int)T;
        if ((rc = gemm(st, q, w.part, w.part_floats, nullptr, w.Lbar, M, -1.0, 0.0, 1)) != IWVI_OK) return rc;
    }
    // dq_mu = A^T DMU
    if (d.dq_mu) {
        GemmArgs q{};
        q.A = d.A; q.a_sm = 1; q.a_sk = Mp; q.B = w.DMU; q.b_sk = R; q.b_sn = 1; q.M = M; q.N = R; q.K = (int)T;
        if ((rc = gemm(st, q, w.part, w.part_floats, d.dq_mu, nullptr, R, 1.0, 0.0, 0)) != IWVI_OK) return rc;
    }
    // dL_r = tril(A^T diag(2 dv_r) U_r)
    if (d.dq_sqrt) for (int r = 0; r < R; ++r) {
        GemmArgs q{};
        q.A = d.A; q.a_sm = 1; q.a_sk = Mp; q.B = d.U + (size_t)r * T * Mp; q.b_sk = Mp; q.b_sn = 1;
        q.scale = w.DV2 + r; q.s_stride = R; q.scale_on_k = 1; q.M = M; q.N = M; q.K = (int)T;
        if ((rc = gemm(st, q, w.part, w.part_floats, d.dq_sqrt + (size_t)r * M * M, nullptr, M, 1.0, 0.0, 1)) != IWVI_OK) return rc;
    }
    // C = -1/2 K o DK (over DA), per-sample sums
    KernArgs ka{d.F, w.Zt, w.invls, w.DK, w.DA, w.RS, T, M, D, d.variance};
    hipLaunchKernelGGL(k_bw_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, st, ka);
    if ((rc = check_launch("k_bw_kernel")) != IWVI_OK) return rc;
    // CZ = C Z~
    {
        GemmArgs q{};
        q.A = w.DA; q.a_sm = M; q.a_sk = 1; q.B = w.Zt; q.b_sk = D; q.b_sn = 1;
        q.C = w.CZ; q.ldc = D; q.M = (int)T; q.N = D; q.K = M; q.alpha = 1.f; q.beta = 0.f;
        if ((rc = gemm_rows(st, q)) != IWVI_OK) return rc;
    }
    hipLaunchKernelGGL(k_bw_dx, dim3((unsigned)((T * D + 255) / 256)), dim3(256), 0, st, d.F, w.invls, w.RS, w.CZ, d.dF, w.Q, (long long)T, D);
    // sums over samples: C^T F, colsum(C), colsum(Q), (sum SDV, sum k dk)
    {
        GemmArgs q{};
        q.A = w.DA; q.a_sm = 1; q.a_sk = M; q.B = d.F; q.b_sk = D; q.b_sn = 1; q.M = M; q.N = D; q.K = (int)T;
        if ((rc = gemm(st, q, w.part, w.part_floats, w.CtF, nullptr, D, 1.0, 0.0, 0)) != IWVI_OK) return rc;
        q.B = w.one; q.b_sk = 0; q.b_sn = 0; q.N = 1;
        if ((rc = gemm(st, q, w.part, w.part_floats, w.colsumC, nullptr, 1, 1.0, 0.0, 0)) != IWVI_OK) return rc;
        q.A = w.Q; q.a_sm = 1; q.a_sk = D; q.M = D;
        if ((rc = gemm(st, q, w.part, w.part_floats, w.dinvls_x, nullptr, 1, 1.0, 0.0, 0)) != IWVI_OK) return rc;
        q.A = w.SDV; q.a_sm = 0; q.a_sk = 1; q.M = 1;
        if ((rc = gemm(st, q, w.part, w.part_floats, w.sums, nullptr, 1, 1.0, 0.0, 0)) != IWVI_OK) return rc;
        q.A = w.RS + 1; q.a_sm = 0; q.a_sk = 2; q.M = 1;
        if ((rc = gemm(st, q, w.part, w.part_floats, w.sums + 1, nullptr, 1, 1.0, 0.0, 0)) != IWVI_OK) return rc;
    }
    // adjoint of Lm = chol(Kuu): S = Lm^-T Phi(Lm^T Lbar) Lm^-1
    {
        const dim3 grid((M + 127) / 128, M), block(128);
        hipLaunchKernelGGL(k_dmm, grid, block, 0, st, Lm64, 1LL, (long long)Mp, (const double*)w.Lbar, (long long)M, 1LL, w.T1, M, M, 1);
        hipLaunchKernelGGL(k_dmm, grid, block, 0, st, Linv64, 1LL, (long long)Mp, (const double*)w.T1, (long long)M, 1LL, w.T2, M, M, 0);
        hipLaunchKernelGGL(k_dmm, grid, block, 0, st, (const double*)w.T2, (long long)M, 1LL, Linv64, (long long)Mp, 1LL, w.S, M, M, 0);
        hipLaunchKernelGGL(k_kuu_bwd, dim3(M), dim3(256), 0, st, w.Zt, (const double*)w.S, M, D, (double)d.variance, w.dZt_uu, w.dvar_m);
        if ((rc = check_launch("cholesky adjoint")) != IWVI_OK) return rc;
    }
    FinalArgsB f{d.Z, d.lengthscales, d.q_mu, d.q_sqrt, w.Zt, w.invls, w.colsumC, w.CtF, w.sums, w.dinvls_x, w.dZt_uu, w.dvar_m,
                 d.dZ, d.dls, d.dvariance, d.dq_mu, d.dq_sqrt, M, D, R, d.kl_weight, (double)d.variance};
    hipLaunchKernelGGL(k_bw_final, dim3(1), dim3(256), 0, st, f);
    return check_launch("k_bw_final");
}

extern "C" int iwvi_iw_elbo_backward(const float* fmean, const float* fvar, const float* Y, int Dy,
                                     const float* const* kl_local, const int32_t* kl_dims, int n_local,
                                     int64_t B, int K, float lik_variance, double scale,
                                     float* out_w, float* d_mean, float* d_var,
                                     double* out_sums /* [2]: sum_n (lse - log K), d/d lik_variance */, double* ws, void* stream_) {
    if (!fmean || !fvar || !Y || !out_sums || !ws || Dy <= 0 || B <= 0 || K <= 0 || n_local < 0 || n_local > IWVI_MAX_KL || !(lik_variance > 0.f)) {
        set_error("iwvi_iw_elbo_backward: bad argument"); return IWVI_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream_;
    ElboBwdArgs a{};
    a.fmean = fmean; a.fvar = fvar; a.Y = Y; a.Dy = Dy; a.n_kl = n_local;
    for (int i = 0; i < n_local; ++i) {
        if (!kl_local || !kl_local[i] || !kl_dims || kl_dims[i] <= 0) { set_error("iwvi_iw_elbo_backward: bad local regulariser %d", i); return IWVI_ERR_ARG; }
        a.kl[i] = kl_local[i]; a.kl_dims[i] = kl_dims[i];
    }
    a.B = B; a.K = K; a.lik_var = lik_variance; a.scale = scale; a.w = out_w; a.d_mean = d_mean; a.d_var = d_var; a.part = ws;
    hipLaunchKernelGGL(k_elbo_bwd, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, a);
    hipLaunchKernelGGL(k_dsum, dim3(2), dim3(256), 0, st, (const double*)ws, (long long)B, out_sums);
    return check_launch("k_elbo_bwd");
}

extern "C" int iwvi_lv_layer_backward(const float* mu, const float* sigma, const float* noise,
                                      const float* dF_next, int ld_next, int col0, const float* w,
                                      int latent_dim, int64_t B, int K, int sampled_kl, float* d_enc_out, void* stream_) {
    if (!mu || !sigma || !noise || !d_enc_out || latent_dim <= 0 || B <= 0 || K <= 0 || (dF_next && (ld_next < col0 + latent_dim || col0 < 0))) {
        set_error("iwvi_lv_layer_backward: bad argument"); return IWVI_ERR_ARG;
    }
    LvBwdArgs a{mu, sigma, noise, dF_next, ld_next, col0, w, latent_dim, B, K, sampled_kl, d_enc_out};
    hipLaunchKernelGGL(k_lv_bwd, dim3((unsigned)((B * latent_dim + 63) / 64)), dim3(64), 0, (hipStream_t)stream_, a);
    return check_launch("k_lv_bwd");
}

static size_t enc_bwd_layout(int64_t rows, const int32_t* dims, int n, size_t* acts_off, size_t* delta_off, size_t* part_off, size_t* part_floats) {
    size_t o = 0;
    int wmax = 1;
    for (int l = 0; l <= n; ++l) { if (acts_off) acts_off[l] = o; o = align256(o + sizeof(float) * rows * dims[l]); if (dims[l] > wmax) wmax = dims[l]; }
    for (int l = 0; l < n; ++l) { if (delta_off) delta_off[l] = o; o = align256(o + sizeof(float) * rows * dims[l + 1]); }
    const size_t pf = (size_t)((rows + 511) / 512 + 2) * wmax * wmax;
    if (part_off) *part_off = o;
    if (part_floats) *part_floats = pf;
    o = align256(o + sizeof(float) * pf);
    o = align256(o + 16);                               // the constant 1
    return o;
}
extern "C" size_t iwvi_encoder_backward_ws_bytes(int64_t rows, const int32_t* dims, int n_enc) {
    if (rows <= 0 || !dims || n_enc <= 0 || n_enc > IWVI_MAX_ENC) return 0;
    return enc_bwd_layout(rows, dims, n_enc, nullptr, nullptr, nullptr, nullptr);
}
extern "C" int iwvi_encoder_backward(const float* XY, int64_t rows, const float* const* enc_W, const float* const* enc_b,
                                     const int32_t* dims, int n_enc, const float* d_out,
                                     float* const* dW, float* const* db, void* ws_, void* stream_) {
    if (!XY || !enc_W || !dims || !d_out || !dW || !ws_ || rows <= 0 || n_enc <= 0 || n_enc > IWVI_MAX_ENC) { set_error("iwvi_encoder_backward: bad argument"); return IWVI_ERR_ARG; }
    for (int l = 0; l <= n_enc; ++l) if (dims[l] <= 0 || dims[l] > 64) { set_error("iwvi_encoder_backward: encoder width %d out of range (1..64)", dims[l]); return IWVI_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream_;
    size_t ao[IWVI_MAX_ENC + 1], d_off[IWVI_MAX_ENC], po, pf;
    const size_t total = enc_bwd_layout(rows, dims, n_enc, ao, d_off, &po, &pf);
    char* base = (char*)ws_;
    float* one = (float*)(base + total - 256);
    EncBwdArgs a{};
    a.XY = XY; a.rows = rows; a.n = n_enc; a.d_out = d_out;
    for (int l = 0; l <= n_enc; ++l) { a.dims[l] = dims[l]; a.acts[l] = (float*)(base + ao[l]); }
    for (int l = 0; l < n_enc; ++l) {
        if (!enc_W[l] || !dW[l]) { set_error("iwvi_encoder_backward: null weight %d", l); return IWVI_ERR_ARG; }
        a.W[l] = enc_W[l]; a.b[l] = enc_b ? enc_b[l] : nullptr; a.delta[l] = (float*)(base + d_off[l]);
    }
    hipLaunchKernelGGL(k_set_one, dim3(1), dim3(64), 0, st, one);
    hipLaunchKernelGGL(k_enc_bwd, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, st, a);
    int rc;
    if ((rc = check_launch("k_enc_bwd")) != IWVI_OK) return rc;
    float* part = (float*)(base + po);
    for (int l = 0; l < n_enc; ++l) {
        const int din = dims[l], dout = dims[l + 1];
        GemmArgs q{};                                   // dW_l = acts_l^T delta_l
        q.A = a.acts[l]; q.a_sm = 1; q.a_sk = din; q.B = a.delta[l]; q.b_sk = dout; q.b_sn = 1; q.M = din; q.N = dout; q.K = (int)rows;
        if ((rc = gemm(st, q, part, pf, dW[l], nullptr, dout, 1.0, 0.0, 0)) != IWVI_OK) return rc;
        if (db && db[l]) {
            q.A = a.delta[l]; q.a_sm = 1; q.a_sk = dout; q.B = one; q.b_sk = 0; q.b_sn = 0; q.M = dout; q.N = 1;
            if ((rc = gemm(st, q, part, pf, db[l], nullptr, 1, 1.0, 0.0, 0)) != IWVI_OK) return rc;
        }
    }
    return IWVI_OK;
}

extern "C" size_t iwvi_natgrad_ws_bytes(int M) {
    if (M <= 0 || M > IWVI_MAX_M) return 0;
    return align256(sizeof(double) * (size_t)M * M) * 8 + align256(sizeof(double) * M) * 4 + align256(iwvi_chol_ws_bytes(M)) + 256;
}

// GPflow 1.x NatGradOptimizer (natural parameterisation, XiNat) on a whitened (q_mu [M, R], q_sqrt [R, M, M]):
//   eta = (m, S + m m^T), theta = (S^-1 m, -1/2 S^-1);  theta <- theta - gamma dLoss/d eta;  back through
//   natural_to_meanvarsqrt (cholesky(-2 theta_2), its inverse, S = X^T X, mu = S theta_1, cholesky(S)).
// dq_mu / dq_sqrt are gradients of the objective that is MAXIMISED (the ELBO): loss = -ELBO.
extern "C" int iwvi_natgrad_step(float* q_mu, float* q_sqrt, const float* dq_mu, const float* dq_sqrt,
                                 int M, int R, double gamma, void* ws_, void* stream_) {
    if (!q_mu || !q_sqrt || !dq_mu || !dq_sqrt || !ws_ || M <= 0 || M > IWVI_MAX_M || R <= 0 || R > IWVI_MAX_R) { set_error("iwvi_natgrad_step: bad argument"); return IWVI_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream_;
    char* base = (char*)ws_; size_t o = 0;
    auto mat = [&]() { double* p = (double*)(base + o); o += align256(sizeof(double) * (size_t)M * M); return p; };
    auto vec = [&]() { double* p = (double*)(base + o); o += align256(sizeof(double) * M); return p; };
    double *L = mat(), *Lbar = mat(), *Linv = mat(), *T1 = mat(), *T2 = mat(), *Sbar = mat(), *Sinv = mat(), *Pn = mat();
    double *m = vec(), *mbar = vec(), *th1 = vec(), *tmp = vec();
    void* cws = base + o;
    const int nb = (M * M + 255) / 256;
    int rc;
    for (int r = 0; r < R; ++r) {
        hipLaunchKernelGGL(k_f2d, dim3(nb), dim3(256), 0, st, (const float*)(q_sqrt + (size_t)r * M * M), (long long)M, L, M, M, 1.0, 1);
        hipLaunchKernelGGL(k_f2d, dim3(nb), dim3(256), 0, st, dq_sqrt + (size_t)r * M * M, (long long)M, Lbar, M, M, -1.0, 1);
        hipLaunchKernelGGL(k_f2d, dim3((M + 255) / 256), dim3(256), 0, st, (const float*)(q_mu + r), (long long)R, m, M, 1, 1.0, 0);
        hipLaunchKernelGGL(k_f2d, dim3((M + 255) / 256), dim3(256), 0, st, dq_mu + r, (long long)R, mbar, M, 1, -1.0, 0);
        hipLaunchKernelGGL(k_tri_inv, dim3((M + 63) / 64), dim3(64), 0, st, (const double*)L, Linv, M);
        // dLoss/dS (symmetric) from dLoss/dL: Sbar = sym(L^-T Phi(L^T Lbar) L^-1)
        dmm(st, L, 1, M, Lbar, M, 1, T1, M, M, M, M, 1.0, nullptr, 0, 0.0, 1);
        dmm(st, Linv, 1, M, T1, M, 1, T2, M, M, M, M);
        dmm(st, T2, M, 1, Linv, M, 1, Sbar, M, M, M, M);
        hipLaunchKernelGGL(k_symmetrise, dim3(nb), dim3(256), 0, st, Sbar, M);
        // g1 = mbar - 2 Sbar m  (d/d eta_1),  g2 = Sbar  (d/d eta_2)
        dmm(st, Sbar, M, 1, m, 1, 1, tmp, 1, M, 1, M, -2.0, mbar, 1, 1.0);
        // theta_1 = S^-1 m, theta_2 = -1/2 S^-1, S^-1 = L^-T L^-1
        dmm(st, Linv, 1, M, Linv, M, 1, Sinv, M, M, M, M);
        dmm(st, Sinv, M, 1, m, 1, 1, th1, 1, M, 1, M);
        // theta_1' = theta_1 - gamma g1 (into mbar);  -2 theta_2' = S^-1 + 2 gamma Sbar (into Pn)
        hipLaunchKernelGGL(k_axpby, dim3((M + 255) / 256), dim3(256), 0, st, (const double*)th1, 1.0, (const double*)tmp, -gamma, mbar, M);
        hipLaunchKernelGGL(k_axpby, dim3(nb), dim3(256), 0, st, (const double*)Sinv, 1.0, (const double*)Sbar, 2.0 * gamma, Pn, M * M);
        // natural_to_meanvarsqrt
        if ((rc = iwvi_chol_factor(Pn, T1, M, cws, stream_)) != IWVI_OK) return rc;
        hipLaunchKernelGGL(k_tri_inv, dim3((M + 63) / 64), dim3(64), 0, st, (const double*)T1, T2, M);
        dmm(st, T2, 1, M, T2, M, 1, Sbar, M, M, M, M);                                   // S' = X^T X
        dmm(st, Sbar, M, 1, mbar, 1, 1, m, 1, M, 1, M);                                  // mu' = S' theta_1'
        if ((rc = iwvi_chol_factor(Sbar, L, M, cws, stream_)) != IWVI_OK) return rc;
        hipLaunchKernelGGL(k_d2f, dim3(nb), dim3(256), 0, st, (const double*)L, q_sqrt + (size_t)r * M * M, (long long)M, M, M, 1);
        hipLaunchKernelGGL(k_d2f, dim3((M + 255) / 256), dim3(256), 0, st, (const double*)m, q_mu + r, (long long)R, M, 1, 0);
        if ((rc = check_launch("iwvi_natgrad_step")) != IWVI_OK) return rc;
    }
    return IWVI_OK;
}

// TensorFlow AdamOptimizer (lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t); x -= lr_t m / (sqrt v + eps)) on the
// unconstrained variables; `maximise` != 0: the gradients are of an objective to maximise (the ELBO).
// init != 0: x <- transform^-1(param), m = v = 0 (no step).
extern "C" int iwvi_adam_step(const iwvi_adam_tensor* tensors, int n_tensors, double lr, double beta1, double beta2,
                              double eps, int64_t t, int maximise, int init, void* stream_) {
    if (!tensors || n_tensors <= 0 || n_tensors > ADAM_MAX || (!init && t < 1)) { set_error("iwvi_adam_step: bad argument (at most %d tensors)", ADAM_MAX); return IWVI_ERR_ARG; }
    AdamArgs a{};
    long long nmax = 1;
    for (int i = 0; i < n_tensors; ++i) {
        const iwvi_adam_tensor& s = tensors[i];
        if (!s.param || !s.x || !s.m || !s.v || (!init && !s.grad) || s.n <= 0 || (s.transform != 0 && s.transform != 1)) { set_error("iwvi_adam_step: bad tensor %d", i); return IWVI_ERR_ARG; }
        a.t[i] = AdamTensor{s.param, s.grad, s.x, s.m, s.v, (long long)s.n, s.transform};
        if (s.n > nmax) nmax = s.n;
    }
    a.n = n_tensors; a.init = init;
    a.lr_t = init ? 0.f : (float)(lr * sqrt(1.0 - pow(beta2, (double)t)) / (1.0 - pow(beta1, (double)t)));
    a.b1 = (float)beta1; a.b2 = (float)beta2; a.eps = (float)eps; a.sign = maximise ? -1.f : 1.f;
    long long blocks = (nmax + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks, n_tensors), dim3(256), 0, (hipStream_t)stream_, a);
    return check_launch("k_adam");
}

// Backward pass of the IW-ELBO path (the reference obtains these gradients from
// TensorFlow's autodiff of the graph built by models.py:112-150, experiments/build_models.py:284-304).
// gfx950 only.  Layer by layer in reverse; the forward (one fused launch) leaves a, u_r, the draws and the per-latent moments
// in HBM.  float32 MFMA products for everything that sums over samples, float64 for the Cholesky adjoint, every sum over
// samples in a fixed order (bit-reproducible gradients).  Per layer: a main chain that produces dF for the layer below
// (k_bw_mid where its shapes allow, else heads / segmented GEMM / GEMM / kernel adjoint), then a parameter branch -- split-K and
// thin products parked in the workspace, ONE deferred reduction, Cholesky adjoint, assembly -- that can run on a side stream
// beside the layer below.  Also here: the ELBO tail, latent-variable layer and encoder adjoints, and the two optimiser steps
// of the reference (NatGrad, Adam).  Measured state and what is next: DESIGN.md section 5b.
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
#include <cstdlib>

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
    // fast path only.  K = nseg segments of kseg: segment s reads A + s*a_seg, B + s*b_seg, scale + s*s_seg (a sum of
    // products in one launch).  nbatch independent products (blockIdx.z = batch*nsplit + split): batch b reads
    // B + b*b_batch, scale + b*s_batch and writes part + b*nsplit*M*N.
    int kseg; long long a_seg, b_seg, s_seg;
    int nbatch; long long b_batch, s_batch;
    // optional epilogue of a row product (replaces beta * C): + sum_r e_dmu[m, r] e_qmu[n, r] - 2 e_sdv[m] e_A[m * e_lda + n]
    const float* e_dmu; const float* e_qmu; const float* e_sdv; const float* e_A; long long e_lda; int e_R;
    int tri_out;                        // split-K products whose strict upper triangle is discarded: those tiles are skipped
    int b_lower_kn;                     // B(k, n) = 0 for k < n (a lower-triangular matrix indexed [k][n]): k starts at the tile's n0
};
constexpr int GT = 64, GK = 16, GLD = GT + 4;

__device__ __forceinline__ float gemm_epilogue(const GemmArgs& g, int m, int n, float acc) {
    float v = g.alpha * acc;
    if (g.e_dmu) {
        float e = -2.f * g.e_sdv[m] * g.e_A[m * g.e_lda + n];
        for (int r = 0; r < g.e_R; ++r) e = fmaf(g.e_dmu[(long long)m * g.e_R + r], g.e_qmu[(long long)n * g.e_R + r], e);
        return v + e;
    }
    return v + (g.beta != 0.f ? g.beta * g.C[m * g.ldc + n] : 0.f);
}

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
            else g.C[m * g.ldc + n] = gemm_epilogue(g, m, n, acc[j][v]);
        }
}


// Fast path: full 64x64 tiles, 32-deep stages, float4 global loads (two per operand per thread per stage: a row of a
// k-contiguous operand contributes 128 contiguous bytes), LDS double buffer (one barrier per stage, the next stage's
// loads in flight under the MFMAs), 2x2 MFMA tiles per wave.
// A_KC: A's k index is contiguous (else its m index); B_NC: B's n index is contiguous (else its k index).
constexpr int GKF = 32, KLD = GKF + 4, OPB = 64 * KLD;     // (OPB floats per operand buffer: 64 x 36 >= 32 x 68)
template <bool A_KC, bool B_NC>
__global__ __launch_bounds__(256) void k_gemm_fast(GemmArgs g) {
    // a k-contiguous operand is kept [row][k] (row stride KLD: float4 stores, no transposition; the MFMA's reads -- 16 rows x
    // 4 k per instruction -- land 2 per bank, the minimum for 64 lanes), a row-contiguous one [k][row] (row stride GLD)
    __shared__ __attribute__((aligned(16))) float Asm[2 * OPB];
    __shared__ __attribute__((aligned(16))) float Bsm[2 * OPB];
    auto AS = [&](int buf, int k, int m) -> float& { return A_KC ? Asm[buf * OPB + m * KLD + k] : Asm[buf * OPB + k * GLD + m]; };
    auto BS = [&](int buf, int k, int n) -> float& { return B_NC ? Bsm[buf * OPB + k * GLD + n] : Bsm[buf * OPB + n * KLD + k]; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    if (g.tri_out && n0 > m0) return;                                     // strictly above the diagonal: discarded by the reduction
    const int batch = (int)blockIdx.z / g.nsplit, split = (int)blockIdx.z - batch * g.nsplit;
    const int kb = split * g.kchunk;
    const int ke = (kb + g.kchunk < g.K) ? kb + g.kchunk : g.K;
    const float* Bb = g.B + batch * g.b_batch;
    const float* Sb = g.scale ? g.scale + batch * g.s_batch : nullptr;
    // this thread's two slots in a stage (j = 0, 1).  k-contiguous operand: (row = tid/8 + 32 j, k = 4*(tid%8)..+3);
    // row-contiguous operand: (k = tid/16 + 16 j, row = 4*(tid%16)..+3)
    const int am = A_KC ? tid >> 3 : (tid & 15) * 4, ak = A_KC ? (tid & 7) * 4 : tid >> 4;
    const int bn = B_NC ? (tid & 15) * 4 : tid >> 3, bk = B_NC ? tid >> 4 : (tid & 7) * 4;
    constexpr int AMJ = A_KC ? 32 : 0, AKJ = A_KC ? 0 : 16, BNJ = B_NC ? 0 : 32, BKJ = B_NC ? 16 : 0;
    f32x4 ra[2], rb[2];
    auto fetch = [&](int k0) {
        const int seg = k0 / g.kseg, kl = k0 - seg * g.kseg;             // a stage never straddles a segment (kseg % 32 == 0)
        const float* A = g.A + seg * g.a_seg; const float* B = Bb + seg * g.b_seg;
        const float* S = Sb ? Sb + seg * g.s_seg : nullptr;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + am + AMJ * j, ka = kl + ak + AKJ * j, n = n0 + bn + BNJ * j, kq = kl + bk + BKJ * j;
            ra[j] = *reinterpret_cast<const f32x4*>(A + (long long)m * g.a_sm + (long long)ka * g.a_sk);
            rb[j] = *reinterpret_cast<const f32x4*>(B + (long long)kq * g.b_sk + (long long)n * g.b_sn);
            if (S) {
                if (g.scale_on_k) {
                    if (A_KC) for (int e = 0; e < 4; ++e) ra[j][e] *= S[(long long)(ka + e) * g.s_stride];
                    else ra[j] *= S[(long long)ka * g.s_stride];
                } else {
                    if (A_KC) ra[j] *= S[(long long)m * g.s_stride];
                    else for (int e = 0; e < 4; ++e) ra[j][e] *= S[(long long)(m + e) * g.s_stride];
                }
            }
            if (g.b_keep_n_ge_k)
                for (int e = 0; e < 4; ++e) { const int nn = n + (B_NC ? e : 0), kk = kq + (B_NC ? 0 : e); if (nn < kk) rb[j][e] = 0.f; }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = am + AMJ * j, ka = ak + AKJ * j, n = bn + BNJ * j, kq = bk + BKJ * j;
            *reinterpret_cast<f32x4*>(&AS(buf, ka, m)) = ra[j];            // 4 consecutive k (A_KC) or 4 consecutive rows: contiguous either way
            *reinterpret_cast<f32x4*>(&BS(buf, kq, n)) = rb[j];
        }
    };
    f32x4 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // stage list.  Split-K: [kb, ke) of the one segment.  Row products: per segment only the k range where the
    // triangular B operand is non-zero for this tile's columns (k < n0 + 64 for L^T-type, k >= n0 for L-type).
    int lo = kb, ns = (ke - kb) / GKF;
    if (g.nsplit == 1) {
        lo = g.b_lower_kn ? n0 : 0;
        const int hi = g.b_keep_n_ge_k ? (n0 + GT < g.kseg ? n0 + GT : g.kseg) : g.kseg;
        ns = hi > lo ? (hi - lo) / GKF : 0;
    }
    const int nstage = (g.nsplit == 1) ? ns * (g.K / g.kseg) : ns;
    auto stage_k = [&](int i) { if (g.nsplit > 1) return lo + GKF * i; const int seg = i / ns; return seg * g.kseg + lo + GKF * (i - seg * ns); };
    if (nstage > 0) { fetch(stage_k(0)); stash(0); }
    __syncthreads();
    int buf = 0;
    for (int i = 0; i < nstage; ++i, buf ^= 1) {
        const bool more = i + 1 < nstage;
        if (more) fetch(stage_k(i + 1));
#pragma unroll
        for (int kk = 0; kk < GKF / 4; ++kk) {
            const int kr = 4 * kk + (lane >> 4);
            const float a0 = AS(buf, kr, wm + (lane & 15)), a1 = AS(buf, kr, wm + 16 + (lane & 15));
            const float b0 = BS(buf, kr, wn + (lane & 15)), b1 = BS(buf, kr, wn + 16 + (lane & 15));
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
    }
    float* part = g.part ? g.part + ((size_t)batch * g.nsplit + split) * g.M * g.N : nullptr;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int v = 0; v < 4; ++v) {
        const int m = m0 + wm + 16 * i + 4 * (lane >> 4) + v, n = n0 + wn + 16 * j + (lane & 15);
        if (part) part[(size_t)m * g.N + n] = acc[i][j][v];
        else g.C[m * g.ldc + n] = gemm_epilogue(g, m, n, acc[i][j][v]);
    }
}
// Larger tiles for the same job when M and N allow it: 128x128 per workgroup, 16-deep stages, each wave a 64x64 quadrant as
// 2x2 v_mfma_f32_32x32x2_f32 tiles (64 accumulator registers): four times the MFMA work per barrier and per byte staged of
// the 64x64 kernel, whose MFMA pipe was measured 20 % busy.
constexpr int BT = 128, BK = 16, BLD = BT + 4;
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <bool A_KC, bool B_NC>
__global__ __launch_bounds__(256) void k_gemm_big(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][BK][BLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int m0 = blockIdx.y * BT, n0 = blockIdx.x * BT;
    if (g.tri_out && n0 > m0) return;
    const int batch = (int)blockIdx.z / g.nsplit, split = (int)blockIdx.z - batch * g.nsplit;
    const int kb = split * g.kchunk;
    const int ke = (kb + g.kchunk < g.K) ? kb + g.kchunk : g.K;
    const float* Bb = g.B + batch * g.b_batch;
    const float* Sb = g.scale ? g.scale + batch * g.s_batch : nullptr;
    // two slots per operand per stage (j = 0, 1).  k-contiguous: (row = tid/4 + 64 j, k = 4*(tid%4)..+3);
    // row-contiguous: (k = tid/32 + 8 j, row = 4*(tid%32)..+3)
    const int am = A_KC ? tid >> 2 : (tid & 31) * 4, ak = A_KC ? (tid & 3) * 4 : tid >> 5;
    const int bn = B_NC ? (tid & 31) * 4 : tid >> 2, bk = B_NC ? tid >> 5 : (tid & 3) * 4;
    constexpr int AMJ = A_KC ? 64 : 0, AKJ = A_KC ? 0 : 8, BNJ = B_NC ? 0 : 64, BKJ = B_NC ? 8 : 0;
    f32x4 ra[2], rb[2];
    auto fetch = [&](int k0) {
        const int seg = k0 / g.kseg, kl = k0 - seg * g.kseg;
        const float* A = g.A + seg * g.a_seg; const float* B = Bb + seg * g.b_seg;
        const float* S = Sb ? Sb + seg * g.s_seg : nullptr;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + am + AMJ * j, ka = kl + ak + AKJ * j, n = n0 + bn + BNJ * j, kq = kl + bk + BKJ * j;
            ra[j] = *reinterpret_cast<const f32x4*>(A + (long long)m * g.a_sm + (long long)ka * g.a_sk);
            rb[j] = *reinterpret_cast<const f32x4*>(B + (long long)kq * g.b_sk + (long long)n * g.b_sn);
            if (S) {
                if (g.scale_on_k) {
                    if (A_KC) for (int e = 0; e < 4; ++e) ra[j][e] *= S[(long long)(ka + e) * g.s_stride];
                    else ra[j] *= S[(long long)ka * g.s_stride];
                } else {
                    if (A_KC) ra[j] *= S[(long long)m * g.s_stride];
                    else for (int e = 0; e < 4; ++e) ra[j][e] *= S[(long long)(m + e) * g.s_stride];
                }
            }
            if (g.b_keep_n_ge_k)
                for (int e = 0; e < 4; ++e) { const int nn = n + (B_NC ? e : 0), kk = kq + (B_NC ? 0 : e); if (nn < kk) rb[j][e] = 0.f; }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = am + AMJ * j, ka = ak + AKJ * j, n = bn + BNJ * j, kq = bk + BKJ * j;
            if (A_KC) for (int e = 0; e < 4; ++e) As[buf][ka + e][m] = ra[j][e];
            else *reinterpret_cast<f32x4*>(&As[buf][ka][m]) = ra[j];
            if (B_NC) *reinterpret_cast<f32x4*>(&Bs[buf][kq][n]) = rb[j];
            else for (int e = 0; e < 4; ++e) Bs[buf][kq + e][n] = rb[j][e];
        }
    };
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    int lo = kb, ns = (ke - kb) / BK;
    if (g.nsplit == 1) {
        lo = g.b_lower_kn ? n0 : 0;
        const int hi = g.b_keep_n_ge_k ? (n0 + BT < g.kseg ? n0 + BT : g.kseg) : g.kseg;
        ns = hi > lo ? (hi - lo) / BK : 0;
    }
    const int nstage = (g.nsplit == 1) ? ns * (g.K / g.kseg) : ns;
    auto stage_k = [&](int i) { if (g.nsplit > 1) return lo + BK * i; const int seg = i / ns; return seg * g.kseg + lo + BK * (i - seg * ns); };
    if (nstage > 0) { fetch(stage_k(0)); stash(0); }
    __syncthreads();
    int buf = 0;
    for (int i = 0; i < nstage; ++i, buf ^= 1) {
        const bool more = i + 1 < nstage;
        if (more) fetch(stage_k(i + 1));
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int kr = 2 * kk + (lane >> 5);
            const float a0 = As[buf][kr][wm + (lane & 31)], a1 = As[buf][kr][wm + 32 + (lane & 31)];
            const float b0 = Bs[buf][kr][wn + (lane & 31)], b1 = Bs[buf][kr][wn + 32 + (lane & 31)];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
    }
    float* part = g.part ? g.part + ((size_t)batch * g.nsplit + split) * g.M * g.N : nullptr;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int v = 0; v < 16; ++v) {
        // 32x32 accumulator: register v of lane l = row 8*(v/4) + 4*(l/32) + v%4, column l%32
        const int m = m0 + wm + 32 * i + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3), n = n0 + wn + 32 * j + (lane & 31);
        if (part) part[(size_t)m * g.N + n] = acc[i][j][v];
        else g.C[m * g.ldc + n] = gemm_epilogue(g, m, n, acc[i][j][v]);
    }
}
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static bool use_big_tiles(const GemmArgs& g) {
    if (g.M % BT || g.N % BT) return false;
    return (long long)(g.M / BT) * (g.N / BT) * g.nsplit * g.nbatch >= 512;
}
// launch on the fast path if the shapes allow it; returns false otherwise (caller falls back to k_gemm)
static bool launch_fast(hipStream_t st, const GemmArgs& g) {
    const bool a_kc = g.a_sk == 1, a_mc = g.a_sm == 1, b_nc = g.b_sn == 1, b_kc = g.b_sk == 1;
    if (!(a_kc || a_mc) || !(b_nc || b_kc)) return false;
    if (g.M % GT || g.N % GT || g.kseg % GKF || g.K % g.kseg || g.kchunk % GKF || (g.K % g.kchunk && g.nsplit > 1) || g.K % GKF) return false;
    if (g.kseg != g.K && g.kchunk % g.kseg && g.kseg % g.kchunk) return false;
    const long long a_ld = a_kc ? g.a_sm : g.a_sk, b_ld = b_nc ? g.b_sk : g.b_sn;
    if (a_ld % 4 || b_ld % 4 || g.a_seg % 4 || g.b_seg % 4 || g.b_batch % 4 || !aligned16(g.A) || !aligned16(g.B)) return false;
    // 128x128 tiles only when they still give every CU two workgroups (measured: at M = 128 / T = 20480 -- 160 big tiles --
    // they are 10 % slower than the 64x64 ones, at M = 256 / T = 204800 2 % faster)
    if (use_big_tiles(g)) {
        const dim3 gridb(g.N / BT, g.M / BT, g.nsplit * g.nbatch), blockb(256);
        if (a_kc && b_nc) hipLaunchKernelGGL((k_gemm_big<true, true>), gridb, blockb, 0, st, g);
        else if (a_kc) hipLaunchKernelGGL((k_gemm_big<true, false>), gridb, blockb, 0, st, g);
        else if (b_nc) hipLaunchKernelGGL((k_gemm_big<false, true>), gridb, blockb, 0, st, g);
        else hipLaunchKernelGGL((k_gemm_big<false, false>), gridb, blockb, 0, st, g);
        return true;
    }
    const dim3 grid(g.N / GT, g.M / GT, g.nsplit * g.nbatch), block(256);
    if (a_kc && b_nc) hipLaunchKernelGGL((k_gemm_fast<true, true>), grid, block, 0, st, g);
    else if (a_kc) hipLaunchKernelGGL((k_gemm_fast<true, false>), grid, block, 0, st, g);
    else if (b_nc) hipLaunchKernelGGL((k_gemm_fast<false, true>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((k_gemm_fast<false, false>), grid, block, 0, st, g);
    return true;
}

// out[m, n] = alpha * sum_s part[s][m][n] (+ beta * out); tri: entries above the diagonal become 0.  Either output
// may be null (float / double).
struct ReduceArgs { const float* part; int S, M, N; float* out; double* out64; long long ldo; double alpha, beta; int tri; long long out_batch;
                    const float* add; double add_coef; int add_diag_inv;      // + add_coef * (add[.] - (m == n ? 1 / add[.] : 0)): the KL gradient rides along
                    int frag; };                                              // > 0 (= M / 16, square, tri): the shares hold only the LOWER 16x16 blocks, each
                                                                              // as one accumulator image (k_bw_chain, phase 5): block (bi, bk <= bi) at
                                                                              // 256 (bi (bi + 1) / 2 + bk), float 4 lane + e of it = entry [16 bi + 4 (lane >> 4) + e][16 bk + (lane & 15)]
// 64 outputs per workgroup; the S partials of an output are summed by 4 threads (contiguous quarters, in order), then combined in a fixed
// order: deterministic whatever the launch geometry.  Output index -> (share offset, m, n): row-major, or by blocks for a `frag` job (the
// blocks above the diagonal are written as zeros, like the discarded upper entries of a row-major `tri` job).
__device__ __forceinline__ void reduce_body(const ReduceArgs& r, int b, int block_x, double (*red)[64]) {
    const int o = threadIdx.x & 63, gq = threadIdx.x >> 6;
    const int idx = block_x * 64 + o;
    int m, n;
    size_t MN, off;
    bool live;
    if (r.frag > 0) {
        const int blk = idx >> 8, within = idx & 255, bi = blk / r.frag, bk = blk - bi * r.frag, ln = within >> 2, e = within & 3;
        m = 16 * bi + 4 * (ln >> 4) + e; n = 16 * bk + (ln & 15);
        MN = (size_t)(r.frag * (r.frag + 1) / 2) * 256;
        off = (size_t)(bi * (bi + 1) / 2 + bk) * 256 + within;
        live = idx < r.M * r.N && bk <= bi;
    } else {
        m = idx / r.N; n = idx - m * r.N;
        MN = (size_t)r.M * r.N; off = (size_t)idx;
        live = idx < r.M * r.N && !(r.tri && n > m);                         // (discarded upper entries are not read: they may be unwritten)
    }
    const float* part = r.part + (size_t)b * r.S * MN;
    double s = 0.0;
    if (live) {
        const int per = (r.S + 3) / 4, k0 = gq * per, k1 = (k0 + per < r.S) ? k0 + per : r.S;
#pragma unroll 8
        for (int k = k0; k < k1; ++k) s += (double)part[(size_t)k * MN + off];
    }
    red[gq][o] = s;
    __syncthreads();
    if (gq != 0 || idx >= r.M * r.N) return;
    s = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
    s *= r.alpha;
    if (r.add) { const double v = (double)r.add[b * r.out_batch + m * r.ldo + n]; s += r.add_coef * (v - ((r.add_diag_inv && m == n) ? 1.0 / v : 0.0)); }
    if (r.tri && n > m) s = 0.0;
    if (r.out) { float* q = r.out + b * r.out_batch + m * r.ldo + n; *q = (float)(s + (r.beta != 0.0 ? r.beta * (double)*q : 0.0)); }
    if (r.out64) { double* q = r.out64 + b * r.out_batch + m * r.ldo + n; *q = s + (r.beta != 0.0 ? r.beta * *q : 0.0); }
}
__global__ __launch_bounds__(256) void k_reduce_parts(ReduceArgs r) {
    __shared__ double red[4][64];
    reduce_body(r, blockIdx.y, blockIdx.x, red);
}

// Deferred reductions: the split-K / thin products of one layer park their partial sums in disjoint slices of the
// workspace and ONE launch sums them all (grid.y = job), instead of one small launch behind every product.
constexpr int RQ_MAX = 12;
struct ReduceJobs { ReduceArgs j[RQ_MAX]; int nb[RQ_MAX]; int n, maxnb; };
__global__ __launch_bounds__(256) void k_reduce_multi(ReduceJobs q) {
    __shared__ double red[4][64];
    const int job = blockIdx.y / q.maxnb, b = blockIdx.y - job * q.maxnb;  // (job, batch)
    if (job >= q.n) return;
    const ReduceArgs& r = q.j[job];
    if (b >= q.nb[job] || (int)blockIdx.x * 64 >= r.M * r.N) return;
    reduce_body(r, b, blockIdx.x, red);
}
struct ReduceQueue {
    float* base; size_t cap, used; ReduceJobs q; int maxblk;
    ReduceQueue(float* b, size_t c) : base(b), cap(c), used(0), maxblk(0) { q.n = 0; q.maxnb = 1; }
    float* take(size_t n) { if (used + n > cap) return nullptr; float* p = base + used; used += (n + 63) & ~size_t(63); return p; }
    bool push(const ReduceArgs& r, int nbatch) {
        if (q.n >= RQ_MAX || nbatch > 32) return false;
        q.j[q.n] = r; q.nb[q.n] = nbatch; ++q.n;
        if (nbatch > q.maxnb) q.maxnb = nbatch;
        const int nb = (r.M * r.N + 63) / 64; if (nb > maxblk) maxblk = nb;
        return true;
    }
    int flush(hipStream_t st) {
        if (q.n == 0) return IWVI_OK;
        hipLaunchKernelGGL(k_reduce_multi, dim3(maxblk, q.maxnb * q.n), dim3(256), 0, st, q);
        q.n = 0; q.maxnb = 1; maxblk = 0; used = 0;
        return check_launch("k_reduce_multi");
    }
};

// samples per split: 512, or the smallest multiple of 512 that keeps the number of splits <= 256 -- one that divides K when
// there is one nearby (the fast GEMM path needs whole chunks)
static int splitk_chunk(long long K) {
    long long q = (K + 512LL * 256 - 1) / (512LL * 256);
    if (q < 1) q = 1;
    for (long long t = q; t < q + 64; ++t) if (K % (512 * t) == 0) return (int)(512 * t);
    return (int)(512 * q);
}
// split-K product(s) summed over many samples: out (+ b*out_batch) = alpha * A^T-style product, reduced in float64
static int gemm(hipStream_t st, GemmArgs g, float* part_ws, size_t part_floats, float* out, double* out64, long long ldo,
                double alpha, double beta, int tri, int nbatch = 1, long long b_batch = 0, long long s_batch = 0, long long out_batch = 0,
                ReduceQueue* rq = nullptr, const float* add = nullptr, double add_coef = 0.0, int add_diag_inv = 0) {
    const int kchunk = splitk_chunk(g.K);                   // 512 samples per split, more once that would exceed 256 splits
    g.nsplit = (g.K + kchunk - 1) / kchunk; g.kchunk = kchunk;
    if (g.nsplit < 2) { g.nsplit = 2; g.kchunk = round_up((g.K + 1) / 2, GK); if (g.kchunk < GK) g.kchunk = GK; }
    g.kseg = g.K; g.a_seg = g.b_seg = g.s_seg = 0; g.nbatch = nbatch; g.b_batch = b_batch; g.s_batch = s_batch;
    if (rq) {                                              // own slice of the workspace; summed by the queue's one launch
        part_ws = rq->take((size_t)g.nsplit * nbatch * g.M * g.N);
        if (!part_ws) { set_error("backward: split-K workspace too small"); return IWVI_ERR_ARG; }
    } else if ((size_t)g.nsplit * nbatch * g.M * g.N > part_floats) { set_error("backward: split-K workspace too small"); return IWVI_ERR_ARG; }
    g.part = part_ws;
    if (!launch_fast(st, g)) {
        for (int b = 0; b < nbatch; ++b) {
            GemmArgs q = g;
            q.B = g.B + b * b_batch; if (g.scale) q.scale = g.scale + b * s_batch;
            q.part = part_ws + (size_t)b * g.nsplit * g.M * g.N;
            hipLaunchKernelGGL(k_gemm, dim3((g.N + GT - 1) / GT, (g.M + GT - 1) / GT, g.nsplit), dim3(256), 0, st, q);
        }
    }
    ReduceArgs r{part_ws, g.nsplit, g.M, g.N, out, out64, ldo, alpha, beta, tri, out_batch, add, add_coef, add_diag_inv};
    if (rq && rq->push(r, nbatch)) return check_launch("k_gemm (split-K)");
    hipLaunchKernelGGL(k_reduce_parts, dim3((g.M * g.N + 63) / 64, nbatch), dim3(256), 0, st, r);
    return check_launch("k_gemm (split-K)");
}
// many rows, short K: direct store.  nseg > 1: the sum of nseg products (segment strides a_seg / b_seg / s_seg)
static int gemm_rows(hipStream_t st, GemmArgs g, int nseg = 1, long long a_seg = 0, long long b_seg = 0, long long s_seg = 0) {
    g.nsplit = 1; g.part = nullptr; g.nbatch = 1; g.b_batch = g.s_batch = 0;
    g.kseg = g.K; g.K *= nseg; g.a_seg = a_seg; g.b_seg = b_seg; g.s_seg = s_seg; g.kchunk = round_up(g.K, GK);
    if (launch_fast(st, g)) return check_launch("k_gemm_fast (rows)");
    const float beta = g.beta;
    for (int sgi = 0; sgi < nseg; ++sgi) {
        GemmArgs q = g;
        q.K = g.kseg; q.kchunk = round_up(q.K, GK); q.A = g.A + sgi * a_seg; q.B = g.B + sgi * b_seg; if (g.scale) q.scale = g.scale + sgi * s_seg;
        q.beta = sgi == 0 ? beta : 1.f;
        if (sgi > 0) q.e_dmu = nullptr;                    // the epilogue terms enter once
        hipLaunchKernelGGL(k_gemm, dim3((g.N + GT - 1) / GT, (g.M + GT - 1) / GT, 1), dim3(256), 0, st, q);
    }
    return check_launch("k_gemm (rows)");
}

// (thin() is defined after k_thin)
// ------------------------------------------------------------------------------------------------------------
// per-sample heads: one wave per sample
// ------------------------------------------------------------------------------------------------------------
struct HeadArgs {
    const float* A; const float* U; const float* eps; const float* W; const float* mfA;
    const float* dFs; const float* dFm; const float* dFv;
    float* DMU; float* DV2; float* SDV; float* dF;
    long long T; int M, Mp, D, R, P, mf_type; float variance;
    const float* var_dev;               // optional device scalar read instead of `variance`
    const float* q_mu; float* GMV;      // optional [T, 3R] = (g_r | mu_r | v_r) per sample, for the mixing matrix's gradient
    const float* gmv_in;                // optional: the same block as the forward left it (then A and U are not read here)
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
    if (!h.gmv_in) {
        for (int m = lane; m < h.M; m += 64) aa = fmaf(a[m], a[m], aa);
        aa = wave_sum(aa);
    }
    float mine_dv = 0.f, mine_dmu = 0.f;                  // lane r keeps latent r's heads
    for (int r = 0; r < h.R; ++r) {
        float uu = 0.f;
        if (!h.gmv_in) {
            const float* u = h.U + ((size_t)r * h.T + t) * h.Mp;
            for (int m = lane; m < h.M; m += 64) uu = fmaf(u[m], u[m], uu);
            uu = wave_sum(uu);
        }
        if (h.GMV && !h.gmv_in) {
            float mu = 0.f;
            for (int m = lane; m < h.M; m += 64) mu = fmaf(a[m], h.q_mu[m * h.R + r], mu);
            mu = wave_sum(mu);
            if (lane == 0) {
                const float vr = fmaxf((h.var_dev ? *h.var_dev : h.variance) - aa + uu, 0.f);
                float* o = h.GMV + t * 3 * h.R;
                o[r] = mu + (h.eps ? h.eps[t * h.R + r] : 0.f) * sqrtf(vr); o[h.R + r] = mu; o[2 * h.R + r] = vr;
            }
        }
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
        const float v = h.gmv_in ? h.gmv_in[t * 3 * h.R + 2 * h.R + r] : (h.var_dev ? *h.var_dev : h.variance) - aa + uu;
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

// kernel value and its derivative w.r.t. the squared scaled distance (GPflow 1.x Stationary kernels, as csrc/precompute.hip):
//   RBF       k = s2 exp(-d2 / 2)                                   dk/dd2 = -k / 2
//   Matern52  k = s2 (1 + a r + a^2 r^2 / 3) exp(-a r), a = sqrt 5  dk/dd2 = -(5/6) s2 (1 + a r) exp(-a r),  r = sqrt(d2 + 1e-12)
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ double fast_exp(double x) { return exp(x); }
template <class T>
__device__ __forceinline__ void kern_and_grad(T d2, int type, T var, T& k, T& g) {
    if (type == IWVI_KERN_MATERN52) {
        const T a = (T)2.23606797749978969641, r = sqrt(d2 + (T)1e-12), e = fast_exp(-a * r);
        k = var * ((T)1 + a * r + (T)(5.0 / 3.0) * r * r) * e;
        g = -var * (T)(5.0 / 6.0) * ((T)1 + a * r) * e;
    } else {
        k = var * fast_exp((T)-0.5 * d2);
        g = (T)-0.5 * k;
    }
}

// One wave per sample: K_uf entries again (RBF, direct differences), c = -1/2 k dk written over DA, then
// dx~ = 2 x~ sum_m c_m - 2 sum_m c_m z~_m, dF += dx~ * invls, and the per-sample row of column-sum inputs
// Qx[t] = (dx~ o x [D] | sum_r dv_r | sum_m k dk).
struct KernArgs { const float* F; const float* Zt; const float* invls; const float* DK; float* C; const float* SDV; float* dF; float* Qx;
                  long long T; int M, D; float variance; int kern_type; const float* var_dev; };
template <int DM>                       // D <= DM: the per-dimension arrays stay in registers
__global__ __launch_bounds__(256) void k_bw_kernel(KernArgs a) {
    // 16 lanes per sample (16 samples per workgroup): lane `sub` takes columns m0 + 4 sub .. + 3 of every 64-column step
    // (one float4 of dK in, one of c out: a sample's row moves as full 256-byte segments), and the D + 2 per-sample sums
    // are 4-step reductions inside the 16-lane group.
    const int sub = threadIdx.x & 15;
    const long long t = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = t < a.T;
    const long long tt = live ? t : a.T - 1;
    auto gsum = [](float v) { for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; };
    float xt[DM], cz[DM];
#pragma unroll
    for (int d = 0; d < DM; ++d) { xt[d] = d < a.D ? a.F[tt * a.D + d] * a.invls[d] : 0.f; cz[d] = 0.f; }
    float sc = 0.f, skd = 0.f;
    const bool vec = (a.M & 3) == 0;
    for (int m0 = 0; m0 < a.M; m0 += 64) {
        const int mb = m0 + 4 * sub;
        float dk[4] = {0.f, 0.f, 0.f, 0.f}, c[4];
        if (vec) { if (mb < a.M) { const f32x4 v = *reinterpret_cast<const f32x4*>(a.DK + tt * a.M + mb); dk[0] = v[0]; dk[1] = v[1]; dk[2] = v[2]; dk[3] = v[3]; } }
        else for (int e = 0; e < 4; ++e) if (mb + e < a.M) dk[e] = a.DK[tt * a.M + mb + e];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = (mb + e < a.M) ? mb + e : a.M - 1;
            float d2 = 0.f, z[DM];
#pragma unroll
            for (int d = 0; d < DM; ++d) { z[d] = d < a.D ? a.Zt[m * a.D + d] : 0.f; const float q = xt[d] - z[d]; d2 = fmaf(q, q, d2); }
            float kv, kg;
            kern_and_grad<float>(d2, a.kern_type, a.var_dev ? *a.var_dev : a.variance, kv, kg);
            const float kd = (mb + e < a.M) ? kv * dk[e] : 0.f;
            c[e] = (mb + e < a.M) ? kg * dk[e] : 0.f;
            sc += c[e]; skd += kd;
#pragma unroll
            for (int d = 0; d < DM; ++d) cz[d] = fmaf(c[e], z[d], cz[d]);
        }
        if (live) {
            if (vec) { if (mb < a.M) *reinterpret_cast<f32x4*>(a.C + t * a.M + mb) = f32x4{c[0], c[1], c[2], c[3]}; }
            else for (int e = 0; e < 4; ++e) if (mb + e < a.M) a.C[t * a.M + mb + e] = c[e];
        }
    }
    sc = gsum(sc); skd = gsum(skd);
    const int W = a.D + 2;
#pragma unroll
    for (int d = 0; d < DM; ++d) {
        if (d < a.D) {
            const float czd = gsum(cz[d]);
            if (live && sub == 0) {
                const float dxt = 2.f * xt[d] * sc - 2.f * czd;
                if (a.dF) a.dF[t * a.D + d] = fmaf(dxt, a.invls[d], a.dF[t * a.D + d]);
                a.Qx[t * W + d] = dxt * a.F[t * a.D + d];
            }
        }
    }
    if (live && sub == 0) { a.Qx[t * W + a.D] = a.SDV[t]; a.Qx[t * W + a.D + 1] = skd; }
}

// ------------------------------------------------------------------------------------------------------------
// The per-sample chain of one layer in ONE launch (M a multiple of 64 up to 256, T a multiple of 64, the forward's gmv
// block at hand): a workgroup owns 64 samples and carries them through
//   heads -> DA = DMU q_mu^T - 2 SDV o A + sum_r (2 dv_r) o (U_r L_r^T)  (tile kept in LDS)
//         -> DK = DA Lm^-1 (A operand straight from that tile)  -> kernel adjoint c = -1/2 k o dk, dx~, dF,
// so that DA never touches HBM, dK is read back from LDS, and four launches (and their ramps) become one.
// Outputs for the sums over samples that follow: DMU, DV2, SDV, DK, C (into the DA buffer), Qx.
// ------------------------------------------------------------------------------------------------------------
struct MidArgs {
    const float* GMV; const float* eps; const float* W; const float* mfA; const float* dFs; const float* dFm; const float* dFv;
    float* DMU; float* DV2; float* SDV; float* dF; int P, mf_type;
    const float* U; const float* A; int Mp; const float* q_sqrt; const float* q_mu;
    const float* LinvF; float* DK;
    const float* F; const float* Zt; const float* invls; float* C; float* Qx;
    long long T; int M, D, R; float variance; int kern_type; const float* var_dev;
};
template <int NP, int DM>                // M = 64 NP;  D <= DM
__global__ __launch_bounds__(256) void k_bw_mid(MidArgs a) {
    extern __shared__ __attribute__((aligned(16))) float msm[];
    const int M = 64 * NP, LDT = M + 4;
    float* tile = msm;                                       // [64][M + 4]: DA, then dK
    float* As = tile + 64 * LDT;                             // 2 operand buffers of OPB floats
    float* Bs = As + 2 * OPB;
    float* dmu_s = Bs + 2 * OPB;                             // [64][R]
    float* dv2_s = dmu_s + 64 * a.R;                         // [64][R]
    float* sdv_s = dv2_s + 64 * a.R;                         // [64]
    float* dfi_s = sdv_s + 64;                               // [64][D]: the mean function's share of dF
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long t0 = (long long)blockIdx.x * 64;
    const int R = a.R, D = a.D, P = a.P;
    auto AK = [&](int buf, int k, int m) -> float& { return As[buf * OPB + m * KLD + k]; };      // k-contiguous operands: [row][k]
    auto BK = [&](int buf, int k, int n) -> float& { return Bs[buf * OPB + n * KLD + k]; };
    auto BS = [&](int buf, int k, int n) -> float& { return Bs[buf * OPB + k * GLD + n]; };      // row-contiguous: [k][row]

    // ---- heads (layers.py:46-48, temp_workaround.py:85-91, :142-145), one thread per sample
    if (tid < 64) {
        const long long t = t0 + tid;
        float sdv = 0.f;
        for (int r = 0; r < R; ++r) {
            float dg = 0.f, dm = 0.f, dvv = 0.f;
            if (a.W) {
                for (int p = 0; p < P; ++p) {
                    const float w = a.W[p * R + r];
                    if (a.dFs) dg = fmaf(w, a.dFs[t * P + p], dg);
                    if (a.dFm) dm = fmaf(w, a.dFm[t * P + p], dm);
                    if (a.dFv) dvv = fmaf(w * w, a.dFv[t * P + p], dvv);
                }
            } else {
                if (a.dFs) dg = a.dFs[t * P + r];
                if (a.dFm) dm = a.dFm[t * P + r];
                if (a.dFv) dvv = a.dFv[t * P + r];
            }
            const float v = a.GMV[t * 3 * R + 2 * R + r];
            float dv = dvv;
            if (v > 0.f) { if (a.eps) dv += dg * a.eps[t * R + r] * 0.5f / sqrtf(v); } else dv = 0.f;
            dmu_s[tid * R + r] = dg + dm; dv2_s[tid * R + r] = 2.f * dv; sdv += dv;
            a.DMU[t * R + r] = dg + dm; a.DV2[t * R + r] = 2.f * dv;
        }
        sdv_s[tid] = sdv; a.SDV[t] = sdv;
    }
    for (int idx = tid; idx < 64 * D; idx += 256) {
        const int j = idx / D, d = idx - j * D;
        const long long t = t0 + j;
        float acc = 0.f;
        if (a.mf_type == IWVI_MF_LINEAR) {
            for (int p = 0; p < P; ++p) {
                float up = 0.f;
                if (a.dFs) up += a.dFs[t * P + p];
                if (a.dFm) up += a.dFm[t * P + p];
                acc = fmaf(a.mfA[d * P + p], up, acc);
            }
        } else if (a.mf_type == IWVI_MF_IDENTITY) {
            if (a.dFs) acc += a.dFs[t * P + d];
            if (a.dFm) acc += a.dFm[t * P + d];
        }
        dfi_s[idx] = acc;
    }
    __syncthreads();

    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int sm_ = tid >> 3, sk_ = (tid & 7) * 4;          // slot of a k-contiguous operand: (row = tid/8 + 32 j, k = 4 (tid%8) ..)
    const int nk_ = tid >> 4, nn_ = (tid & 15) * 4;         // slot of a row-contiguous operand: (k = tid/16 + 16 j, col = 4 (tid%16) ..)
    // ---- DA tile: for each 64-column pass, R segments of the contraction (k < n0 + 64 only: L_r is lower triangular)
    for (int np = 0; np < NP; ++np) {
        const int n0 = 64 * np, hi = n0 + 64, ns = hi / GKF, nstage = ns * R;
        f32x4 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 ra[2], rb[2];
        auto fetch = [&](int s) {
            const int r = s / ns, kl = (s - r * ns) * GKF;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = sm_ + 32 * j, k = kl + sk_;
                ra[j] = *reinterpret_cast<const f32x4*>(a.U + ((size_t)r * a.T + t0 + m) * a.Mp + k);
                ra[j] *= dv2_s[m * R + r];
                const int n = n0 + sm_ + 32 * j;
                rb[j] = *reinterpret_cast<const f32x4*>(a.q_sqrt + ((size_t)r * M + n) * M + k);      // B(k, n) = L_r[n][k], n >= k
                for (int e = 0; e < 4; ++e) if (n < k + e) rb[j][e] = 0.f;
            }
        };
        auto stash = [&](int buf) {
#pragma unroll
            for (int j = 0; j < 2; ++j) { *reinterpret_cast<f32x4*>(&AK(buf, sk_, sm_ + 32 * j)) = ra[j]; *reinterpret_cast<f32x4*>(&BK(buf, sk_, sm_ + 32 * j)) = rb[j]; }
        };
        fetch(0); stash(0);
        __syncthreads();
        int buf = 0;
        for (int s = 0; s < nstage; ++s, buf ^= 1) {
            const bool more = s + 1 < nstage;
            if (more) fetch(s + 1);
#pragma unroll
            for (int kk = 0; kk < GKF / 4; ++kk) {
                const int kr = 4 * kk + (lane >> 4);
                const float a0 = AK(buf, kr, wm + (lane & 15)), a1 = AK(buf, kr, wm + 16 + (lane & 15));
                const float b0 = BK(buf, kr, wn + (lane & 15)), b1 = BK(buf, kr, wn + 16 + (lane & 15));
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            if (more) stash(buf ^ 1);
            __syncthreads();
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int v = 0; v < 4; ++v) {
            const int m = wm + 16 * i + 4 * (lane >> 4) + v, n = n0 + wn + 16 * j + (lane & 15);
            float e = -2.f * sdv_s[m] * a.A[(t0 + m) * a.Mp + n];
            for (int r = 0; r < R; ++r) e = fmaf(dmu_s[m * R + r], a.q_mu[n * R + r], e);
            tile[m * LDT + n] = acc[i][j][v] + e;
        }
    }
    __syncthreads();

    // ---- dK = DA Lm^-1: A operand from the tile, B = Lm^-1 rows k >= n0 staged through LDS; all passes' accumulators are
    //      kept so that the tile can be overwritten only once every pass has read it
    f32x4 acc2[NP][2][2];
    for (int np = 0; np < NP; ++np) {
        const int n0 = 64 * np, nstage = (M - n0) / GKF;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc2[np][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 rb[2];
        auto fetch = [&](int s) {
            const int kl = n0 + s * GKF;
#pragma unroll
            for (int j = 0; j < 2; ++j) rb[j] = *reinterpret_cast<const f32x4*>(a.LinvF + (size_t)(kl + nk_ + 16 * j) * M + n0 + nn_);
        };
        auto stash = [&](int buf) {
#pragma unroll
            for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(&BS(buf, nk_ + 16 * j, nn_)) = rb[j];
        };
        fetch(0); stash(0);
        __syncthreads();
        int buf = 0;
        for (int s = 0; s < nstage; ++s, buf ^= 1) {
            const bool more = s + 1 < nstage;
            if (more) fetch(s + 1);
            const int kb = n0 + s * GKF;
#pragma unroll
            for (int kk = 0; kk < GKF / 4; ++kk) {
                const int kr = 4 * kk + (lane >> 4);
                const float a0 = tile[(wm + (lane & 15)) * LDT + kb + kr], a1 = tile[(wm + 16 + (lane & 15)) * LDT + kb + kr];
                const float b0 = BS(buf, kr, wn + (lane & 15)), b1 = BS(buf, kr, wn + 16 + (lane & 15));
                acc2[np][0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc2[np][0][0], 0, 0, 0);
                acc2[np][0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc2[np][0][1], 0, 0, 0);
                acc2[np][1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc2[np][1][0], 0, 0, 0);
                acc2[np][1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc2[np][1][1], 0, 0, 0);
            }
            if (more) stash(buf ^ 1);
            __syncthreads();
        }
    }
    for (int np = 0; np < NP; ++np)
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int v = 0; v < 4; ++v) {
            const int m = wm + 16 * i + 4 * (lane >> 4) + v, n = 64 * np + wn + 16 * j + (lane & 15);
            tile[m * LDT + n] = acc2[np][i][j][v];
            a.DK[(t0 + m) * M + n] = acc2[np][i][j][v];
        }
    __syncthreads();

    // ---- kernel adjoint (RBF, direct differences), 16 lanes per sample, 16 samples per round
    const int sub = tid & 15;
    auto gsum = [](float v) { for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; };
    for (int round = 0; round < 4; ++round) {
        const int j = 16 * round + (tid >> 4);
        const long long t = t0 + j;
        float xt[DM], cz[DM];
#pragma unroll
        for (int d = 0; d < DM; ++d) { xt[d] = d < D ? a.F[t * D + d] * a.invls[d] : 0.f; cz[d] = 0.f; }
        float sc = 0.f, skd = 0.f;
        for (int m0 = 0; m0 < M; m0 += 64) {
            const int mb = m0 + 4 * sub;
            const f32x4 dk = *reinterpret_cast<const f32x4*>(&tile[j * LDT + mb]);
            f32x4 c;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float d2 = 0.f, z[DM];
#pragma unroll
                for (int d = 0; d < DM; ++d) { z[d] = d < D ? a.Zt[(mb + e) * D + d] : 0.f; const float q = xt[d] - z[d]; d2 = fmaf(q, q, d2); }
                float kv, kg;
                kern_and_grad<float>(d2, a.kern_type, a.var_dev ? *a.var_dev : a.variance, kv, kg);
                const float kd = kv * dk[e];
                c[e] = kg * dk[e];
                sc += c[e]; skd += kd;
#pragma unroll
                for (int d = 0; d < DM; ++d) cz[d] = fmaf(c[e], z[d], cz[d]);
            }
            *reinterpret_cast<f32x4*>(a.C + t * M + mb) = c;
        }
        sc = gsum(sc); skd = gsum(skd);
        const int Wq = D + 2;
#pragma unroll
        for (int d = 0; d < DM; ++d) {
            if (d < D) {
                const float czd = gsum(cz[d]);
                if (sub == 0) {
                    const float dxt = 2.f * xt[d] * sc - 2.f * czd;
                    if (a.dF) a.dF[t * D + d] = fmaf(dxt, a.invls[d], dfi_s[j * D + d]);
                    a.Qx[t * Wq + d] = dxt * a.F[t * D + d];
                }
            }
        }
        if (sub == 0) { a.Qx[t * Wq + D] = sdv_s[j]; a.Qx[t * Wq + D + 1] = skd; }
    }
}
template <int NP>
static void launch_mid_d(hipStream_t st, const MidArgs& a, dim3 grid, size_t lds) {
    if (a.D <= 8) hipLaunchKernelGGL((k_bw_mid<NP, 8>), grid, dim3(256), lds, st, a);
    else if (a.D <= 16) hipLaunchKernelGGL((k_bw_mid<NP, 16>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_bw_mid<NP, 32>), grid, dim3(256), lds, st, a);
}
// returns 1 if the fused kernel was launched, 0 if the shapes do not allow it, < 0 on error
static int launch_mid(hipStream_t st, const MidArgs& a) {
    const int M = a.M;
    if (!a.GMV || M % 64 || M > 256 || M == 192 || a.T % 64 || a.Mp != M) return 0;
    // measured (configs[1] / [2] / [3]): -3.5 % of the whole evaluation at M = 128, T = 20480; +7 % at T = 5120 (80 workgroups
    // for 256 CUs) and +10 % at M = 256 (105 KB of LDS: one workgroup per CU) -- so only where it wins, unless forced
    if (!dbg_opt("IWVI_BW_FUSED") && (M > 128 || a.T < 16384)) return 0;
    if (!aligned16(a.U) || !aligned16(a.q_sqrt) || !aligned16(a.LinvF) || !aligned16(a.C)) return 0;
    const size_t lds = sizeof(float) * ((size_t)64 * (M + 4) + 4 * OPB + (size_t)64 * (2 * a.R + 1) + (size_t)64 * a.D);
    static bool done = false;
    if (!done) {
        const size_t most = sizeof(float) * ((size_t)64 * 260 + 4 * OPB + (size_t)64 * (2 * IWVI_MAX_R + 1) + (size_t)64 * IWVI_MAX_D);
        const void* fns[] = {(const void*)k_bw_mid<1, 8>, (const void*)k_bw_mid<1, 16>, (const void*)k_bw_mid<1, 32>,
                             (const void*)k_bw_mid<2, 8>, (const void*)k_bw_mid<2, 16>, (const void*)k_bw_mid<2, 32>,
                             (const void*)k_bw_mid<4, 8>, (const void*)k_bw_mid<4, 16>, (const void*)k_bw_mid<4, 32>};
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)most) != hipSuccess) { set_error("hipFuncSetAttribute(k_bw_mid)"); return IWVI_ERR_LAUNCH; }
        done = true;
    }
    const dim3 grid((unsigned)(a.T / 64));
    if (M == 64) launch_mid_d<1>(st, a, grid, lds);
    else if (M == 128) launch_mid_d<2>(st, a, grid, lds);
    else launch_mid_d<4>(st, a, grid, lds);
    const int rc = check_launch("k_bw_mid");
    return rc == IWVI_OK ? 1 : rc;
}

// ------------------------------------------------------------------------------------------------------------
// The per-sample chain, streaming form (M a multiple of 16 up to 128, T a multiple of 16): the structure of the fused
// forward kernel applied to the adjoint.  Two identities remove the saved u_r = L_r^T a altogether:
//     sum_r 2 dv_r L_r u_r = sum_r 2 dv_r S_r a,   S_r = L_r L_r^T            (this kernel: da)
//     dL_r = tril(A^T diag(2 dv_r) U_r) = tril(G_r L_r),   G_r = A^T diag(2 dv_r) A   (a weighted Gram of A: split-K SYRK + k_gl_tril)
// so the forward keeps only a (T x M floats instead of (R + 1) T x M), and the adjoint's largest product reads its A
// operand -- the packed 16x16 blocks of S_r, prepared once per evaluation by k_pack_bw -- straight from L2 in MFMA
// fragment order while the chunk's a tile sits in LDS in the forward's B-operand layout:
//   phase 0  heads, a rows HBM -> LDS tile [(bk*4 + g) * (NSAMP + 4) + sample] (float4 = 4 consecutive m of one sample); a workgroup
//            (8 waves) owns 16 NS samples, NS chosen like the forward's so that every CU gets one workgroup
//   phase 1  da(bi) = sum_r 2dv_r o [sum_bk S_r(bi, bk) a(bk)] - 2 (sum_r dv_r) a(bi) + sum_r q_mu_r(bi) dmu_r
//            one wave per output row-block: R * nbk packed blocks streamed back to back, 4 NS MFMAs each
//   phase 2  dk(bi) = sum_{bk >= bi} Lm^-T(bi, bk) da(bk)      packed upper blocks of Lm^-T; row-blocks paired (bi, nbk-1-bi)
//   phase 3  kernel adjoint c = dk o dk/dr2..., dx~, dF, Qx rows (as k_bw_mid)
// ------------------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) f32x4* bw_gptr4;
// asynchronous global -> LDS copy of n floats by a 512-thread workgroup (LDS-DMA: no VGPR round trip; complete after the issuing
// wave's s_waitcnt vmcnt(0) and a barrier).  One wave-instruction moves 64 consecutive floats; dst + i0 is wave-uniform.
__device__ __forceinline__ void bw_async_copy(const float* __restrict__ src, float* lds_dst, int n, int tid) {
    const int lane = tid & 63;
    for (int i0 = (tid & ~63); i0 < n; i0 += 512) {
        const int i = i0 + lane;
        if (i < n)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i),
                                             (__attribute__((address_space(3))) void*)(lds_dst + i0), 4, 0, 0);
    }
}
// products over samples on f16 operands (k_bw_chain, phase 5): the sample the e-th f16 of lane group g stands for in k-step ks.
// A full step covers 32 samples as 4e + g; the half step that ends an odd NS covers 16 as 2e + g on lane groups 0 and 1.
template <int NS>
__device__ __forceinline__ int chain_smp(int ks, int e, int g) {
    return ((NS & 1) && ks == NS / 2) ? 32 * ks + 2 * e + (g & 1) : 32 * ks + 4 * e + g;
}
// maximum of a non-negative value over the wave, wave-uniform: four DPP steps inside the rows of 16 lanes, then one lane of each row
__device__ __forceinline__ float chain_wave_max(float v) {
    int x = __float_as_int(v);
    x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false))));    // quad_perm [1,0,3,2]
    x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false))));    // quad_perm [2,3,0,1]
    x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false))));   // row_half_mirror
    x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false))));   // row_mirror
    const float m0 = __int_as_float(__builtin_amdgcn_readlane(x, 0)), m1 = __int_as_float(__builtin_amdgcn_readlane(x, 16));
    const float m2 = __int_as_float(__builtin_amdgcn_readlane(x, 32)), m3 = __int_as_float(__builtin_amdgcn_readlane(x, 48));
    return fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
}
struct ChainArgs {
    const float* GMV; const float* eps; const float* W; const float* mfA; const float* dFs; const float* dFm; const float* dFv;
    float* DMU; float* DV2; float* SDV; float* dF; int P, mf_type;
    const float* A; int Mp; const float* q_mu;
    const float* SP;                     // packed S_r blocks, [bi][r][bk], 256 floats each
    const float* LinvTP;                 // packed upper blocks of Lm^-T, row-block major (tri_upper_off)
    float* DK;
    const float* F; const float* Zt; const float* invls; float* C; float* Qx;
    long long T; int M, D, R, nbk; float variance; int kern_type; int dbg_exit;
    const float* var_dev;                // optional device scalar read instead of `variance`
    int dsz;                             // floats of the da tile's LDS region (it also stages the heads' and the kernel adjoint's inputs)
    // this workgroup's share of the thin sums over samples, each job [workgroups][len] at its own base (summed by k_reduce_multi):
    float* p_qmu;                        // [M][R]      sum_j a[m][j] dmu[j][r]                      -> dq_mu
    float* p_ctf;                        // [M][D + 1]  sum_j c[j][m] [F[j][d] | 1]                  -> dZ~ terms
    float* p_q;                          // [D + 2]     sum_j (dx~ o x | sum_r dv_r | sum_m k dk)   -> dls, dvariance terms
    float* p_w;                          // 3 [P][R]    dFs^T G | dFm^T MU | dFv^T V                 -> dW        (or NULL)
    float* p_a;                          // 2 [D][P]    F^T dFs | F^T dFm                            -> dmf_A     (or NULL)
    // this workgroup's share of the two M x M products over samples (lower 16x16 blocks only, dense [M][M] addressing):
    float* p_lm;                         // [S][M][M]      sum_j dk[m][j] a[n][j]                    -> dLm = -tril(.)   (reduced with alpha = -1)
    float* p_g;                          // [R][S][M][M]   sum_j 2dv_r[j] a[m][j] a[n][j]            -> G_r (dL_r = tril(G_r L_r))
    int S;                               // workgroups of the launch (stride of a batch in p_g)
    int q_only;                          // only dq_mu / dq_sqrt are wanted (the natural-gradient op): heads, dq_mu shares, G_r shares
    int ts;                              // float4 per tile row (16 NS, + 4 of padding where it fits)
    const float* SP16; const float* spf; int s16;   // phase 1 on split-f16 operands (a third tile holds a as two f16 planes)
    int p5h;                             // phase 5 (products over samples) on split-f16 operands: with s16 unless the development switch IWVI_BW_P5_F32 is set
    int z_lds;                           // the scaled inducing inputs are staged in LDS for the kernel adjoint (M <= 256; beyond: read from L2)
    const float* ZtP; const float* cst; int nsteps;   // the state's K_uf operand (A-fragment order), its constant block (1/ls | centre | extent), k-steps
};
template <int NS, int DM>               // 16 NS samples per workgroup (8 waves);  D <= DM
__global__ __launch_bounds__(512) void k_bw_chain(ChainArgs a) {
    constexpr int NSAMP = 16 * NS;
    const int TS = a.ts;                 // float4 per tile row: NSAMP samples (+ 4 of padding where the LDS allows it, so that the TRANSPOSED
                                         // scalar reads of a tile -- products over samples: lanes 4 rows x 4 samples x 4 entries -- spread over all banks)
    extern __shared__ __attribute__((aligned(16))) float csm[];
    const int M = a.M, nbk = a.nbk, R = a.R, D = a.D, P = a.P;
    // two tiles in the forward's B-operand layout, float4 [(bk*4 + g) * TS + sample] = 4 consecutive m of one sample:
    float* tileA = csm;                                      // a (kept to the end: the products over samples read it)
    float* tileK = tileA + TS * M;                        // da, then dk (in place: phase 2), then c = dk o dk/dd2 (kernel adjoint)
    float* tileH = tileK + TS * M;                        // (s16) a as two f16 planes, rows interleaved: float4 row 8kc + 2g = h1, + 1 = h2 of a[32kc + 8g .. + 7]
    float* tileD = tileH + (a.s16 ? TS * M : 0);          // staging: the heads' inputs before, the kernel adjoint's operands after
    float* qmu_s = tileD + a.dsz;                            // [M][R]
    float* dmu_s = qmu_s + M * R;                            // [NSAMP][R]
    float* dv2_s = dmu_s + NSAMP * R;                        // [NSAMP][R]
    float* sdv_s = dv2_s + NSAMP * R;                        // [NSAMP]
    float* dfi_s = sdv_s + NSAMP;                            // [NSAMP][D]
    float* fr = dfi_s + NSAMP * D;                           // [NSAMP][DM]  raw input rows (zero beyond D)
    float* il = fr + NSAMP * DM;                             // [DM]         1 / lengthscales
    float* qx_s = il + DM;                                   // [NSAMP][D + 2]  per-sample rows of the column sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gq = lane >> 4, jq = lane & 15;
    const long long t0 = (long long)blockIdx.x * NSAMP;
    f32x4* tA4 = reinterpret_cast<f32x4*>(tileA);
    f32x4* tK4 = reinterpret_cast<f32x4*>(tileK);

    // ---- phase 0: a rows -> LDS (B-operand layout), q_mu -> LDS; the heads' inputs (rows of the upstream gradients, of the
    //      forward's variances and draws, W, mfA) staged in the not-yet-used da tile with coalesced loads, then the heads from
    //      LDS by all threads (a thread-per-sample loop over global memory is a chain of ~R*P dependent round trips) --------
    float* ups_s = tileD;                                    // [3][NSAMP][P]: dFs | dFm | dFv rows (0 where absent)
    float* W_s = ups_s + 3 * NSAMP * P;                         // [P][R]
    float* gmv_s = W_s + P * R;                              // [NSAMP][3R]: (g | mu | v) of the latent GPs
    float* eps_s = gmv_s + NSAMP * 3 * R;                    // [NSAMP][R]
    float* mfA_s = eps_s + NSAMP * R;                           // [D][P]
    {
        // every input of the chunk in ONE round trip: the contiguous pieces by LDS-DMA (no register staging, nothing waits), the two
        // that change layout (a rows -> B-operand order, F rows -> DM-padded) through registers with all their loads issued first.
        // (Before: one loop per piece, each ending in s_waitcnt vmcnt(0) -- a dozen dependent round trips, most of this phase.)
        const int q4 = M >> 2;                               // float4 per row
        constexpr int AB = 8;                                // a-row float4 per thread and batch
        const float* up[3] = {a.dFs, a.dFm, a.dFv};
        bw_async_copy(a.q_mu, qmu_s, M * R, tid);
#pragma unroll
        for (int u = 0; u < 3; ++u) if (up[u]) bw_async_copy(up[u] + (size_t)t0 * P, ups_s + u * NSAMP * P, NSAMP * P, tid);
        if (a.W) bw_async_copy(a.W, W_s, P * R, tid);
        bw_async_copy(a.GMV + (size_t)t0 * 3 * R, gmv_s, NSAMP * 3 * R, tid);
        if (a.eps) bw_async_copy(a.eps + (size_t)t0 * R, eps_s, NSAMP * R, tid);
        if (a.mf_type == IWVI_MF_LINEAR) bw_async_copy(a.mfA, mfA_s, D * P, tid);
        float fv[(5 * 16 * DM + 511) / 512];
#pragma unroll
        for (int u = 0; u < (5 * 16 * DM + 511) / 512; ++u) {
            const int idx = tid + 512 * u, j = idx / DM, d = idx - j * DM;
            fv[u] = (idx < NSAMP * DM && d < D) ? a.F[(size_t)(t0 + j) * D + d] : 0.f;
        }
        const float ilv = (tid < D) ? a.invls[tid] : 0.f;
        for (int b0 = 0; b0 < NSAMP * q4; b0 += AB * 512) {
            f32x4 av[AB];
#pragma unroll
            for (int u = 0; u < AB; ++u) {
                const int idx = b0 + tid + 512 * u, j = idx / q4, q = idx - j * q4;
                if (idx < NSAMP * q4) av[u] = *reinterpret_cast<const f32x4*>(a.A + (size_t)(t0 + j) * a.Mp + 4 * q);
            }
#pragma unroll
            for (int u = 0; u < AB; ++u) {
                const int idx = b0 + tid + 512 * u, j = idx / q4, q = idx - j * q4;
                if (idx < NSAMP * q4) tA4[q * TS + j] = av[u];
            }
        }
#pragma unroll
        for (int u = 0; u < (5 * 16 * DM + 511) / 512; ++u) if (tid + 512 * u < NSAMP * DM) fr[tid + 512 * u] = fv[u];
        if (tid < DM) il[tid] = ilv;
        // the pieces a caller may leave out
#pragma unroll
        for (int u = 0; u < 3; ++u) if (!up[u]) for (int idx = tid; idx < NSAMP * P; idx += 512) ups_s[u * NSAMP * P + idx] = 0.f;
        if (!a.W) for (int idx = tid; idx < P * R; idx += 512) W_s[idx] = ((idx / R) == (idx % R) ? 1.f : 0.f);
        if (!a.eps) for (int idx = tid; idx < NSAMP * R; idx += 512) eps_s[idx] = 0.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the LDS-DMA copies of this wave have landed
    }
    __syncthreads();
    if (a.s16 && !a.q_only) {                                // the a tile once more as split f16 (scaled by 2^ea): phase 1's B operand
        using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
        const float sa = a.cst[IWVI_CST_SA];
        f32x4* tH4 = reinterpret_cast<f32x4*>(tileH);
        const int nvec = nbk * 2 * NSAMP;
        for (int v = tid; v < nvec; v += 512) {
            const int j = v % NSAMP, kg = v / NSAMP, row = (2 * kg) * TS + j;
            const f32x4 x0 = tA4[row], x1 = tA4[row + TS];
            f16x8 h1, h2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a0 = x0[e] * sa, a1 = x1[e] * sa;
                h1[e] = (_Float16)a0; h2[e] = (_Float16)(a0 - (float)h1[e]);
                h1[4 + e] = (_Float16)a1; h2[4 + e] = (_Float16)(a1 - (float)h1[4 + e]);
            }
            tH4[row] = __builtin_bit_cast(f32x4, h1); tH4[row + TS] = __builtin_bit_cast(f32x4, h2);
        }
        // (made visible by the barrier that ends the heads)
    }
    if (a.dbg_exit == 10) return;
    for (int idx = tid; idx < NSAMP * R; idx += 512) {       // heads: one thread per (sample, latent GP)
        const int j = idx / R, r = idx - j * R;
        float dg = 0.f, dm = 0.f, dvv = 0.f;
        for (int p = 0; p < P; ++p) {
            const float w = W_s[p * R + r];
            dg = fmaf(w, ups_s[j * P + p], dg);
            dm = fmaf(w, ups_s[NSAMP * P + j * P + p], dm);
            dvv = fmaf(w * w, ups_s[2 * NSAMP * P + j * P + p], dvv);
        }
        const float v = gmv_s[j * 3 * R + 2 * R + r];
        float dv = dvv;
        if (v > 0.f) dv += dg * eps_s[idx] * 0.5f / sqrtf(v); else dv = 0.f;
        dmu_s[idx] = dg + dm; dv2_s[idx] = 2.f * dv;
        a.DV2[(size_t)t0 * R + idx] = 2.f * dv;             // (the G_r product's per-sample weights)
    }
    for (int idx = tid; idx < NSAMP * D; idx += 512) {       // the mean function's share of dF
        const int j = idx / D, d = idx - j * D;
        float acc = 0.f;
        if (a.mf_type == IWVI_MF_LINEAR) {
            for (int p = 0; p < P; ++p) acc = fmaf(mfA_s[d * P + p], ups_s[j * P + p] + ups_s[NSAMP * P + j * P + p], acc);
        } else if (a.mf_type == IWVI_MF_IDENTITY) acc = ups_s[j * P + d] + ups_s[NSAMP * P + j * P + d];
        dfi_s[idx] = acc;
    }
    __syncthreads();
    if (a.dbg_exit == 11) return;
    // this workgroup's share of the thin sums whose operands are at hand now (fixed summation order).  dq_mu's share
    // sum_j a[m][j] dmu[j][r] is an M x 16 MFMA product over the chunk's samples (a read transposed, one row-block per wave);
    // the two small ones (dW, dmf_A) run beside it on different threads.
    {
        float* pq = a.p_qmu + (size_t)blockIdx.x * M * R;
        const int nrb = (R + 15) >> 4;
        for (int job = wave; job < nbk * nrb; job += 8) {
            const int bi = job / nrb, rb = job - bi * nrb, rcol = 16 * rb + jq;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4 * NS; ++q) {
                const int smp = 4 * q + gq;
                const float av = tileA[((size_t)(bi * 4 + (jq >> 2)) * TS + smp) * 4 + (jq & 3)];
                const float bv = rcol < R ? dmu_s[smp * R + rcol] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
            }
            if (rcol < R) {
#pragma unroll
                for (int e = 0; e < 4; ++e) pq[(size_t)(16 * bi + 4 * gq + e) * R + rcol] = acc[e];
            }
        }
        const int n_w = a.p_w ? 3 * P * R : 0, n_a = a.p_a ? 2 * D * P : 0;
        for (int idx = tid; idx < n_w + n_a; idx += 512) {
            float acc = 0.f;
            if (idx < n_w) {                                     // dW shares: dFs^T G | dFm^T MU | dFv^T V
                const int u = idx / (P * R), pr = idx - u * P * R, p_ = pr / R, r = pr - p_ * R;
#pragma unroll 16
                for (int j = 0; j < NSAMP; ++j) acc = fmaf(ups_s[u * NSAMP * P + j * P + p_], gmv_s[j * 3 * R + u * R + r], acc);
                a.p_w[(size_t)blockIdx.x * 3 * P * R + idx] = acc;
            } else {                                             // dmf_A shares: F^T dFs | F^T dFm
                const int i2 = idx - n_w, u = i2 / (D * P), dp = i2 - u * D * P, d = dp / P, p_ = dp - d * P;
#pragma unroll 16
                for (int j = 0; j < NSAMP; ++j) acc = fmaf(fr[j * DM + d], ups_s[u * NSAMP * P + j * P + p_], acc);
                a.p_a[(size_t)blockIdx.x * 2 * D * P + i2] = acc;
            }
        }
    }
    if (tid < NSAMP) {                                       // sum_r dv_r per sample, in index order
        float sdv = 0.f;
        for (int r = 0; r < R; ++r) sdv += 0.5f * dv2_s[tid * R + r];
        sdv_s[tid] = sdv;
    }
    __syncthreads();
    if (a.dbg_exit == 1) return;
    if (a.q_only && !a.p_g) return;                          // (G_r then comes from the split-K GEMM over the saved a rows)

    // ---- phase 1 on split-f16 operands: S_r (bi, :) as nbk / 2 slabs per latent GP, three v_mfma_f32_16x16x32_f16 per slab and sub-tile
    if (a.s16 && !a.q_only) {
        using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
        const f32x4* tH4 = reinterpret_cast<const f32x4*>(tileH);
        const int nkc = nbk >> 1;
        for (int bi = wave; bi < nbk; bi += 8) {
            bw_gptr4 Pb = (bw_gptr4)a.SP16 + (size_t)bi * R * nkc * 128 + lane;
            const int nsl = R * nkc;
            f32x4 tot[NS], acc[NS];
#pragma unroll
            for (int t = 0; t < NS; ++t) { tot[t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            f32x4 r1[4], r2[4];
#pragma unroll
            for (int u = 0; u < 3; ++u) { const size_t o_ = (size_t)(u < nsl ? u : nsl - 1) * 128; r1[u] = Pb[o_]; r2[u] = Pb[o_ + 64]; }
            int kc = 0, r = 0;
            f32x4 b1[NS], b2[NS];
#pragma unroll
            for (int t = 0; t < NS; ++t) b1[t] = tH4[(2 * gq) * TS + 16 * t + jq];
            for (int q0 = 0; q0 < nsl; q0 += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = q0 + u;
                    if (q < nsl) {
                        const size_t nx = (size_t)(q + 3 < nsl ? q + 3 : nsl - 1) * 128;
                        r1[(u + 3) & 3] = Pb[nx]; r2[(u + 3) & 3] = Pb[nx + 64];
                        const f16x8 a1 = __builtin_bit_cast(f16x8, r1[u]), a2 = __builtin_bit_cast(f16x8, r2[u]);
#pragma unroll
                        for (int t = 0; t < NS; ++t) b2[t] = tH4[(8 * kc + 2 * gq + 1) * TS + 16 * t + jq];
                        if (kc == 0) {                       // first slab of a latent GP: onto the constant 0 (nothing cleared between them)
                            const f32x4 Z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(f16x8, b1[t]), Z, 0, 0, 0);
                        } else {
#pragma unroll
                        for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(f16x8, b1[t]), acc[t], 0, 0, 0);
                        }
#pragma unroll
                        for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, __builtin_bit_cast(f16x8, b1[t]), acc[t], 0, 0, 0);
                        const int kc_n = kc + 1 == nkc ? 0 : kc + 1;
#pragma unroll
                        for (int t = 0; t < NS; ++t) b1[t] = tH4[(8 * kc_n + 2 * gq) * TS + 16 * t + jq];
#pragma unroll
                        for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(f16x8, b2[t]), acc[t], 0, 0, 0);
                        if (++kc == nkc) {                   // S_r a done for this row-block: back to its scale, weight by 2 dv_r per sample
                            const float fr = a.spf[r];
#pragma unroll
                            for (int t = 0; t < NS; ++t) {
                                const float w = dv2_s[(16 * t + jq) * R + r] * fr;
#pragma unroll
                                for (int e = 0; e < 4; ++e) tot[t][e] = fmaf(w, acc[t][e], tot[t][e]);
                            }
                            kc = 0; ++r;
                        }
                    }
                }
            }
            // (+ q_mu dmu: per latent GP the 4 + NS operands read once -- per element it was 2 R LDS reads for R multiply-adds; same order)
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                const int j = 16 * t + jq;
                const f32x4 av = tA4[(bi * 4 + gq) * TS + j];
                const float m2 = -2.f * sdv_s[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) tot[t][e] = fmaf(m2, av[e], tot[t][e]);
            }
            for (int rr = 0; rr < R; ++rr) {
                float qm[4], dm[NS];
#pragma unroll
                for (int e = 0; e < 4; ++e) qm[e] = qmu_s[(16 * bi + 4 * gq + e) * R + rr];
#pragma unroll
                for (int t = 0; t < NS; ++t) dm[t] = dmu_s[(16 * t + jq) * R + rr];
#pragma unroll
                for (int t = 0; t < NS; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) tot[t][e] = fmaf(dm[t], qm[e], tot[t][e]);
            }
#pragma unroll
            for (int t = 0; t < NS; ++t) tK4[(bi * 4 + gq) * TS + 16 * t + jq] = tot[t];
        }
    } else
    // ---- phase 1: da row-blocks bi = wave, wave + 4 -------------------------------------------------------------------
    for (int bi = wave; bi < nbk && !a.q_only; bi += 8) {
        bw_gptr4 Pb = (bw_gptr4)a.SP + (size_t)bi * R * nbk * 64 + lane;      // this row-block's R * nbk blocks, contiguous
        const int nblocks = R * nbk;
        f32x4 tot[NS], acc[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) { tot[t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        f32x4 a_nx = Pb[0], a_n2 = Pb[(size_t)(1 < nblocks ? 1 : 0) * 64];
        int bk = 0, r = 0;
        for (int q = 0; q < nblocks; ++q) {
            const f32x4 a_cur = a_nx;
            a_nx = a_n2;
            a_n2 = Pb[(size_t)(q + 2 < nblocks ? q + 2 : nblocks - 1) * 64];
            f32x4 b[NS];
#pragma unroll
            for (int t = 0; t < NS; ++t) b[t] = tA4[(bk * 4 + gq) * TS + 16 * t + jq];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s], b[t][s], acc[t], 0, 0, 0);
            }
            if (++bk == nbk) {                               // S_r a done for this row-block: weight by 2 dv_r per sample
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                    const float w = dv2_s[(16 * t + jq) * R + r];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { tot[t][e] = fmaf(w, acc[t][e], tot[t][e]); acc[t][e] = 0.f; }
                }
                bk = 0; ++r;
            }
        }
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int j = 16 * t + jq;
            const f32x4 av = tA4[(bi * 4 + gq) * TS + j];
            const float m2 = -2.f * sdv_s[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) tot[t][e] = fmaf(m2, av[e], tot[t][e]);
        }
        for (int rr = 0; rr < R; ++rr) {
            float qm[4], dm[NS];
#pragma unroll
            for (int e = 0; e < 4; ++e) qm[e] = qmu_s[(16 * bi + 4 * gq + e) * R + rr];
#pragma unroll
            for (int t = 0; t < NS; ++t) dm[t] = dmu_s[(16 * t + jq) * R + rr];
#pragma unroll
            for (int t = 0; t < NS; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) tot[t][e] = fmaf(dm[t], qm[e], tot[t][e]);
        }
#pragma unroll
        for (int t = 0; t < NS; ++t) tK4[(bi * 4 + gq) * TS + 16 * t + jq] = tot[t];
    }
    __syncthreads();
    if (a.dbg_exit == 2) return;
    // (s16) the third tile is free now: a once more, TRANSPOSED, as the f16 operand of the products over samples (phase 5) -- per column
    // block bk and k-step of 32 samples two 1-KiB planes h1 | h2 of 2^ea a, lane 16g + n holding a[16bk + n][sample(ks, e, g)], e = 0..7
    // (chain_smp: any bijection between a k-step's 32 samples and (g, e) serves as long as both operands use it; this one keeps the
    // transposed scalar reads of a tile off each other's banks).  An odd NS ends in a half step: lanes g < 2 only, half the bytes.
    if (a.p5h && (a.p_lm || a.p_g)) {
        using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
        constexpr int KS = (NS + 1) / 2, PB = (NS / 2) * 128 + (NS & 1) * 64;      // k-steps; float4 per column block
        const float sa = a.cst[IWVI_CST_SA];
        f32x4* tT4 = reinterpret_cast<f32x4*>(tileH);
        for (int v = tid; v < nbk * KS * 64; v += 512) {
            const int ln = v & 63, bs = v >> 6, bk = bs / KS, ks = bs - bk * KS, n = ln & 15, g = ln >> 4;
            const bool half_step = (NS & 1) && ks == NS / 2;
            if (half_step && g >= 2) continue;
            f16x8 h1, h2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int smp = chain_smp<NS>(ks, e, g);
                const float x = tileA[((size_t)(bk * 4 + (n >> 2)) * TS + smp) * 4 + (n & 3)] * sa;
                h1[e] = (_Float16)x; h2[e] = (_Float16)(x - (float)h1[e]);
            }
            f32x4* dst = tT4 + (size_t)bk * PB + (half_step ? (NS / 2) * 128 : ks * 128);
            dst[ln] = __builtin_bit_cast(f32x4, h1); dst[(half_step ? 32 : 64) + ln] = __builtin_bit_cast(f32x4, h2);
        }
        // (visible behind the barriers that end phase 2)
    }

    // ---- phase 2: dk(bi) = sum_{bk >= bi} Lm^-T(bi, bk) da(bk); row-blocks paired so that every wave streams nbk + 1 blocks
    //      IN PLACE over da: every wave holds its (at most two) result row-blocks in registers until all have read da.
    const int npair = (nbk + 1) / 2;
    const bool split8 = nbk <= 8;                            // (four pairs at most: a pair's two row-blocks go to the waves w and w + 4 -- one SIMD, nbk + 1
                                                             //  blocks between them as before, but two waves to hide each other's operand latency)
    f32x4 res[4][NS];                                        // (nbk <= 32: two pairs = four row-blocks per wave at most)
    int rbi[4] = {-1, -1, -1, -1};
    if (!a.q_only)
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const int p_ = split8 ? (wave & 3) + 4 * pp : wave + 8 * pp;
        if (p_ >= npair) continue;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (split8 && pass != (wave >> 2)) continue;
            const int bi = pass == 0 ? p_ : nbk - 1 - p_;
            if (pass == 1 && bi <= p_) continue;                // (the middle row-block of an odd nbk: once)
            const int slot = 2 * pp + pass;
            bw_gptr4 Pb = (bw_gptr4)a.LinvTP + (size_t)tri_upper_off(nbk, bi) * 64 + lane;
            const int nblocks = nbk - bi;
#pragma unroll
            for (int t = 0; t < NS; ++t) res[slot][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 a_nx = Pb[0];
            for (int q = 0; q < nblocks; ++q) {
                const f32x4 a_cur = a_nx;
                a_nx = Pb[(size_t)(q + 1 < nblocks ? q + 1 : q) * 64];
                const int bk = bi + q;
                f32x4 b[NS];
#pragma unroll
                for (int t = 0; t < NS; ++t) b[t] = tK4[(bk * 4 + gq) * TS + 16 * t + jq];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int t = 0; t < NS; ++t) res[slot][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s], b[t][s], res[slot][t], 0, 0, 0);
                }
            }
            rbi[slot] = bi;
        }
    }
    __syncthreads();                                         // every read of da is done
#pragma unroll
    for (int slot = 0; slot < 4; ++slot) {
        if (rbi[slot] < 0) continue;
        const int bi = rbi[slot];
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int j = 16 * t + jq;
            tK4[(bi * 4 + gq) * TS + j] = res[slot][t];
            if (a.DK) *reinterpret_cast<f32x4*>(a.DK + (size_t)(t0 + j) * M + 16 * bi + 4 * gq) = res[slot][t];   // (only for the split-K GEMM path of dLm)
        }
    }
    __syncthreads();
    if (a.dbg_exit == 3) return;

    // ---- phase 5 (before the kernel adjoint overwrites dk): this workgroup's share of the two products over samples,
    //      P_lm(bi, bk) = sum_j dk(bi)[., j] a(bk)[., j]^T and P_g[r](bi, bk) = sum_j 2dv_r[j] a(bi)[., j] a(bk)[., j]^T, lower blocks.
    //      The contraction runs over the chunk's samples (4 per MFMA step), so both operands are read TRANSPOSED from the tiles
    //      (scalar LDS reads); the A side of a (row-block, product) pair -- NSAMP / 4 registers, for G_r scaled by 2dv_r -- is
    //      read once and reused for every bk <= bi.  Waves: row-block pair (w & 3, nbk-1-(w & 3)) x half of the 1 + R products.
    const size_t SHF = (size_t)(nbk * (nbk + 1) / 2) * 256;     // floats of one share: the lower blocks, 256 each
    if (a.p5h) {
        // split-f16 form (round 6): the contraction index is the sample, so a k-step of v_mfma_f32_16x16x32_f16 takes 32 of them; x = h1 + h2 on
        // both sides, three products (h1 h1' on one accumulator, h2 h1' + h1 h2' on another).  B = the transposed planes of 2^ea a built behind
        // phase 1; A = the row-block of dk (item 0) or of 2dv_r o a, read transposed from the fp32 tiles once per (row-block, item), scaled by the
        // power of two that brings ITS largest entry to [2^13, 2^14) (a wave-wide maximum: 4 DPP steps + 4 v_readlane) and split in registers.
        // 9 MFMAs of 16 clocks per block product at NS = 5, where the fp32 form issues 20 of 32.
        using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
        constexpr int KS = (NS + 1) / 2, PB = (NS / 2) * 128 + (NS & 1) * 64;
        const f32x4* tT4 = reinterpret_cast<const f32x4*>(tileH);
        const float sa = a.cst[IWVI_CST_SA];
        const int pr = wave & 3, half = wave >> 2;
        const int it0 = a.p_lm ? 0 : 1, nit = R + 1 - it0;
        const int i_lo = it0 + (half == 0 ? 0 : (nit + 1) / 2), i_hi = it0 + (half == 0 ? (nit + 1) / 2 : nit);
        for (int pass = 0; pass < 2 && (a.p_lm || a.p_g); ++pass) {
            const int bi = pass == 0 ? pr : nbk - 1 - pr;
            if (bi < 0 || bi >= nbk) continue;
            if (pass == 0 ? (pr > nbk - 1 - pr) : (nbk - 1 - pr <= pr)) continue;
            for (int it = i_lo; it < i_hi; ++it) {
                if (it > 0 && !a.p_g) continue;
                const float* src = it == 0 ? tileK : tileA;
                float x[KS][8];
                float mx = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bool live = !((NS & 1) && ks == NS / 2) || gq < 2;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int smp = chain_smp<NS>(ks, e, gq);
                        float v = 0.f;
                        if (live) {
                            v = src[((size_t)(bi * 4 + (jq >> 2)) * TS + smp) * 4 + (jq & 3)];
                            if (it > 0) v *= dv2_s[smp * R + (it - 1)];
                        }
                        x[ks][e] = v;
                        mx = fmaxf(mx, fabsf(v));
                    }
                }
                mx = chain_wave_max(mx);
                int ex = ((__float_as_int(mx) >> 23) & 0xff) - 127;
                ex = mx > 0.f ? (ex < -100 ? -100 : (ex > 100 ? 100 : ex)) : 13;
                const float sc = __int_as_float((127 + 13 - ex) << 23);                       // |x| sc < 2^14
                const float back = __int_as_float((127 - 13 + ex) << 23) / sa;              // (both powers of two: exact)
                f16x8 A1[KS], A2[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xs = x[ks][e] * sc;
                        A1[ks][e] = (_Float16)xs; A2[ks][e] = (_Float16)(xs - (float)A1[ks][e]);
                    }
                // (a share = the lower blocks as accumulator images, 1 KiB each: one 16-byte store per lane -- whole cache lines -- where the dense
                //  [M][M] addressing took four 4-byte stores into 64-byte row segments; ReduceArgs.frag)
                f32x4* outp = reinterpret_cast<f32x4*>(it == 0 ? a.p_lm + (size_t)blockIdx.x * SHF
                                                               : a.p_g + ((size_t)(it - 1) * a.S + blockIdx.x) * SHF) + (size_t)(bi * (bi + 1) / 2) * 64 + lane;
                for (int bk = 0; bk <= bi; ++bk) {
                    const f32x4* Tb = tT4 + (size_t)bk * PB;
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, cor = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        f32x4 b1 = {0.f, 0.f, 0.f, 0.f}, b2 = {0.f, 0.f, 0.f, 0.f};
                        if ((NS & 1) && ks == NS / 2) {
                            if (lane < 32) { b1 = Tb[(NS / 2) * 128 + lane]; b2 = Tb[(NS / 2) * 128 + 32 + lane]; }
                        } else { b1 = Tb[ks * 128 + lane]; b2 = Tb[ks * 128 + 64 + lane]; }
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1[ks], __builtin_bit_cast(f16x8, b1), acc, 0, 0, 0);
                        cor = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[ks], __builtin_bit_cast(f16x8, b1), cor, 0, 0, 0);
                        cor = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1[ks], __builtin_bit_cast(f16x8, b2), cor, 0, 0, 0);
                    }
                    f32x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = (acc[e] + cor[e]) * back;
                    outp[(size_t)bk * 64] = o4;
                }
            }
        }
    } else {
        const int pr = wave & 3, half = wave >> 2;
        const int it0 = a.p_lm ? 0 : 1, nit = R + 1 - it0;                                                     // item 0 = dLm, 1 + r = G_r
        const int i_lo = it0 + (half == 0 ? 0 : (nit + 1) / 2), i_hi = it0 + (half == 0 ? (nit + 1) / 2 : nit);
        for (int pass = 0; pass < 2 && (a.p_lm || a.p_g); ++pass) {
            const int bi = pass == 0 ? pr : nbk - 1 - pr;
            if (bi < 0 || bi >= nbk) continue;
            if (pass == 0 ? (pr > nbk - 1 - pr) : (nbk - 1 - pr <= pr)) continue;
            for (int it = i_lo; it < i_hi; ++it) {
                if (it > 0 && !a.p_g) continue;              // (G_r not asked for)
                const float* src = it == 0 ? tileK : tileA;
                float av[4 * NS];
#pragma unroll
                for (int q = 0; q < 4 * NS; ++q) {           // A[i = jq][k = sample 4q + gq] = src[row 16bi + jq][sample]
                    const int smp = 4 * q + gq;
                    float v = src[((size_t)(bi * 4 + (jq >> 2)) * TS + smp) * 4 + (jq & 3)];
                    if (it > 0) v *= dv2_s[smp * R + (it - 1)];
                    av[q] = v;
                }
                f32x4* outp = reinterpret_cast<f32x4*>(it == 0 ? a.p_lm + (size_t)blockIdx.x * SHF
                                                               : a.p_g + ((size_t)(it - 1) * a.S + blockIdx.x) * SHF) + (size_t)(bi * (bi + 1) / 2) * 64 + lane;
                // (measured and rejected, round 4: two column blocks at a time with even / odd k-steps on their own accumulators -- four
                //  independent MFMA chains instead of one of 20 dependent ones -- 35.5 -> 40 us for the two launches of configs[2])
                for (int bk = 0; bk <= bi; ++bk) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 4 * NS; ++q) {       // B[k = sample 4q + gq][j = jq] = a[row 16bk + jq][sample]
                        const float bv = tileA[((size_t)(bk * 4 + (jq >> 2)) * TS + 4 * q + gq) * 4 + (jq & 3)];
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv, acc, 0, 0, 0);
                    }
                    outp[(size_t)bk * 64] = acc;
                }
            }
        }
    }
    if (a.q_only || a.dbg_exit == 4) return;
    __syncthreads();                                         // phase 3 rewrites the dk tile

    // ---- phases 3 + 4 for the RBF kernel, on the matrix cores.  dK/dd2 = -K/2, so the adjoint needs K(z_m, x_j) itself: it is
    //      REBUILT exactly as the forward builds it (x~ rows, then exp2 of one small MFMA product with the state's Z~ operand --
    //      or the differenced form when the inducing cloud is wide, csrc/dgp_forward.hip "Gram form"), in the dk tile's layout, so
    //      c = -K dk / 2 is one multiply per entry.  sum_m c[m] z~[m][d] (-> dx~) and sum_j c[j][m] [F | 1] (-> dZ~ shares) are
    //      MFMA products over the tile.  Before: ~3 D + 10 VALU instructions per (sample, inducing point), 27 + 5 us of the
    //      117 us of this kernel at configs[2]'s first layer.
    if (a.kern_type == IWVI_KERN_RBF) {
        const int nsteps = a.nsteps, XS = 4 * nsteps;
        float* xt = tileD;                                   // [NSAMP][XS]  x~ = ((x/ls - centre) | -|.|^2/2 | 1 | 0..)
        float* zs = xt + NSAMP * XS;                         // [M][DM]      z/ls (zero beyond D); M > 256: not staged, read from L2
        const bool z_lds = a.z_lds != 0;
        float* red = zs + (z_lds ? M * DM : 0);              // [2][8][NSAMP] per-wave shares of sum_m c and sum_m K dk
        const float* cst = a.cst;
        if (z_lds) for (int idx = tid; idx < M * DM; idx += 512) { const int m = idx / DM, d = idx - m * DM; zs[idx] = d < D ? a.Zt[m * D + d] : 0.f; }
        if (tid < NSAMP) {
            float n2 = 0.f;
            for (int d = 0; d < D; ++d) { const float v = fmaf(fr[tid * DM + d], cst[d], -cst[32 + d]); xt[tid * XS + d] = v; n2 = fmaf(v, v, n2); }
            xt[tid * XS + D] = -0.5f * n2; xt[tid * XS + D + 1] = 1.f;
            for (int d = D + 2; d < XS; ++d) xt[tid * XS + d] = 0.f;
        }
        __syncthreads();
        const bool gram_mfma = __float_as_int(cst[64]) <= __float_as_int(4.0f);
        float sc[NS], skd[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) { sc[t] = 0.f; skd[t] = 0.f; }
        for (int bi = wave; bi < nbk; bi += 8) {
            f32x4 acc[NS];
#pragma unroll
            for (int t = 0; t < NS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (gram_mfma) {
                const float* zp = a.ZtP + (size_t)bi * nsteps * 64 + lane;
                for (int s_ = 0; s_ < nsteps; ++s_) {
                    const float av = zp[s_ * 64];
#pragma unroll
                    for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xt[(16 * t + jq) * XS + 4 * s_ + gq], acc[t], 0, 0, 0);
                }
            } else {
                const float sx = 1.4426950408889634f;
                for (int d = 0; d < D; ++d) {
                    const f32x4 z4 = *reinterpret_cast<const f32x4*>(a.ZtP + ((size_t)bi * nsteps + (d >> 2)) * 64 + 16 * (d & 3) + 4 * gq);
#pragma unroll
                    for (int t = 0; t < NS; ++t) {
                        const float xv = sx * xt[(16 * t + jq) * XS + d];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float df = xv - z4[e]; acc[t][e] = fmaf(df, df, acc[t][e]); }
                    }
                }
                const float of = __log2f(a.var_dev ? *a.var_dev : a.variance);
#pragma unroll
                for (int t = 0; t < NS; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][e] = fmaf(acc[t][e], -0.5f / 1.4426950408889634f, of);
            }
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                const int slot = (bi * 4 + gq) * TS + 16 * t + jq;
                const f32x4 dk = tK4[slot];
                f32x4 c;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float kd = __builtin_amdgcn_exp2f(acc[t][e]) * dk[e];     // (log2 of the variance is folded into Z~)
                    c[e] = -0.5f * kd;
                    sc[t] += c[e]; skd[t] += kd;
                }
                tK4[slot] = c;                                       // c over dk (same lane, same slot)
            }
        }
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            sc[t] += __shfl_xor(sc[t], 16, 64); sc[t] += __shfl_xor(sc[t], 32, 64);
            skd[t] += __shfl_xor(skd[t], 16, 64); skd[t] += __shfl_xor(skd[t], 32, 64);
            if (gq == 0) { red[wave * NSAMP + 16 * t + jq] = sc[t]; red[(8 + wave) * NSAMP + 16 * t + jq] = skd[t]; }
        }
        __syncthreads();
        const int Wq = D + 2, ndb = (D + 15) >> 4;
        for (int job = wave; job < NS * ndb; job += 8) {     // CZ[d][j] = sum_m z~[m][d] c[j][m]: a 16 x 16 block per (sub-tile, 16 dims)
            const int t = job / ndb, db = job - t * ndb, dcol = 16 * db + jq;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int bk = 0; bk < nbk; ++bk) {
                const f32x4 b = tK4[(bk * 4 + gq) * TS + 16 * t + jq];
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    const int m_ = 16 * bk + 4 * gq + s_;
                    const float av = z_lds ? (dcol < DM ? zs[m_ * DM + dcol] : 0.f) : (dcol < D ? a.Zt[m_ * D + dcol] : 0.f);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[s_], acc, 0, 0, 0);
                }
            }
            const int j = 16 * t + jq;
            float scj = 0.f;
#pragma unroll
            for (int w_ = 0; w_ < 8; ++w_) scj += red[w_ * NSAMP + j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = 16 * db + 4 * gq + e;
                if (d < D) {
                    const float x_ = fr[j * DM + d], dxt = 2.f * (x_ * il[d]) * scj - 2.f * acc[e];
                    if (a.dF) a.dF[(t0 + j) * D + d] = fmaf(dxt, il[d], dfi_s[j * D + d]);
                    qx_s[j * Wq + d] = dxt * x_;
                }
            }
        }
        if (tid < NSAMP) {
            float s_ = 0.f;
#pragma unroll
            for (int w_ = 0; w_ < 8; ++w_) s_ += red[(8 + w_) * NSAMP + tid];
            qx_s[tid * Wq + D] = sdv_s[tid]; qx_s[tid * Wq + D + 1] = s_;
        }
        __syncthreads();
        if (a.dbg_exit == 5) return;
        {   // this workgroup's share of C^T [F | 1]: P[m][col] = sum_j c[j][m] [F | 1][j][col], a 16 x 16 block per (row-block, 16 columns)
            const int W1 = D + 1, ncb = (W1 + 15) >> 4;
            float* pc = a.p_ctf + (size_t)blockIdx.x * M * W1;
            for (int job = wave; job < nbk * ncb; job += 8) {
                const int bi = job / ncb, cb = job - bi * ncb, col = 16 * cb + jq;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4 * NS; ++q) {
                    const int smp = 4 * q + gq;
                    const float av = tileK[((size_t)(bi * 4 + (jq >> 2)) * TS + smp) * 4 + (jq & 3)];
                    const float bv = col < D ? fr[smp * DM + col] : (col == D ? 1.f : 0.f);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
                }
                if (col < W1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) pc[(size_t)(16 * bi + 4 * gq + e) * W1 + col] = acc[e];
                }
            }
            if (tid < Wq) {
                float acc = 0.f;
                for (int j = 0; j < NSAMP; ++j) acc += qx_s[j * Wq + tid];
                a.p_q[(size_t)blockIdx.x * Wq + tid] = acc;
            }
        }
        return;
    }
    // ---- phase 3 (Matern-5/2): kernel adjoint (direct differences), 16 lanes per sample, 16 samples per round.  The scaled inducing inputs
    //      and the chunk's input rows are staged in the (now free) da tile, padded to DM columns: per-element global loads in the
    //      inner loops were a chain of dependent L1 round trips (50-70 us of this kernel) ------------------------------------
    float* zs = tileD;                                       // [M][DM]   (zero beyond D); M > 256: read from L2 instead
    const bool z_lds = a.z_lds != 0;
    if (z_lds) for (int idx = tid; idx < M * DM; idx += 512) { const int m = idx / DM, d = idx - m * DM; zs[idx] = d < D ? a.Zt[m * D + d] : 0.f; }
    __syncthreads();
    const int sub = tid & 15;
    auto gsum = [](float v) { for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; };
    for (int round = 0; round < (NSAMP + 31) / 32; ++round) {
        const int j = 32 * round + (tid >> 4);
        if (j >= NSAMP) break;                                // (uniform over each 16-lane group)
        const long long t = t0 + j;
        float xt[DM], cz[DM];
#pragma unroll
        for (int d = 0; d < DM; ++d) { xt[d] = fr[j * DM + d] * il[d]; cz[d] = 0.f; }
        float sc = 0.f, skd = 0.f;
        for (int m0 = 0; m0 < M; m0 += 64) {
            const int mb = m0 + 4 * sub;
            if (mb < M) {
                const f32x4 dk = tK4[(mb >> 2) * TS + j];
                f32x4 c;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float d2 = 0.f, z[DM];
#pragma unroll
                    for (int d = 0; d < DM; ++d) { z[d] = z_lds ? zs[(mb + e) * DM + d] : (d < D ? a.Zt[(mb + e) * D + d] : 0.f); const float q = xt[d] - z[d]; d2 = fmaf(q, q, d2); }
                    float kv, kg;
                    kern_and_grad<float>(d2, a.kern_type, a.var_dev ? *a.var_dev : a.variance, kv, kg);
                    const float kd = kv * dk[e];
                    c[e] = kg * dk[e];
                    sc += c[e]; skd += kd;
#pragma unroll
                    for (int d = 0; d < DM; ++d) cz[d] = fmaf(c[e], z[d], cz[d]);
                }
                tK4[(mb >> 2) * TS + j] = c;                          // c over dk (same thread, same slot): the C^T F sums below
            }
        }
        sc = gsum(sc); skd = gsum(skd);
        const int Wq = D + 2;
#pragma unroll
        for (int d = 0; d < DM; ++d) {
            if (d < D) {
                const float czd = gsum(cz[d]);
                if (sub == 0) {
                    const float dxt = 2.f * xt[d] * sc - 2.f * czd;
                    if (a.dF) a.dF[t * D + d] = fmaf(dxt, il[d], dfi_s[j * D + d]);
                    qx_s[j * Wq + d] = dxt * fr[j * DM + d];
                }
            }
        }
        if (sub == 0) { qx_s[j * Wq + D] = sdv_s[j]; qx_s[j * Wq + D + 1] = skd; }
    }
    __syncthreads();
    if (a.dbg_exit == 5) return;
    // ---- phase 4: this workgroup's share of C^T [F | 1] and of the column sums of Qx (fixed order: sample index) -----------
    {
        const int W1 = D + 1;
        float* pc = a.p_ctf + (size_t)blockIdx.x * M * W1;
        for (int idx = tid; idx < M * W1; idx += 512) {
            const int m = idx / W1, d = idx - m * W1;
            float acc = 0.f;
            if (d < D) {
#pragma unroll 16
                for (int j = 0; j < NSAMP; ++j) acc = fmaf(tileK[((size_t)(m >> 2) * TS + j) * 4 + (m & 3)], fr[j * DM + d], acc);
            } else {
#pragma unroll 16
                for (int j = 0; j < NSAMP; ++j) acc += tileK[((size_t)(m >> 2) * TS + j) * 4 + (m & 3)];
            }
            pc[idx] = acc;
        }
        const int Wq = D + 2;
        if (tid < Wq) {
            float acc = 0.f;
            for (int j = 0; j < NSAMP; ++j) acc += qx_s[j * Wq + tid];
            a.p_q[(size_t)blockIdx.x * Wq + tid] = acc;
        }
    }
}
// operands of k_bw_chain, once per evaluation: packed S_r = L_r L_r^T blocks [bi][r][bk] and the packed upper blocks of Lm^-T.
// One workgroup per 16x16 block, one thread per entry (M^3 R flops in all: negligible, latency-bound, off the critical path).
// SP16 (or NULL): the split-f16 image of the same S_r blocks for v_mfma_f32_16x16x32_f16 (iwvi_common.h: s16_*): slab (bi, r, kc) = the
// blocks bk = 2kc, 2kc + 1, two planes of 64 lanes x 8 halves; scaled by 2^es_r with M max|L_r|^2 2^es_r <= 2^14 (max|L_r| taken from L_r
// here, rounded as the precompute's q(u) role rounds it); spf[r] = 2^-(es_r + ea), what an accumulated row-block is multiplied by.
__device__ __forceinline__ void pack_bw_body(int b, const float* __restrict__ q_sqrt, const double* __restrict__ Linv64, int Mp, int M, int R, int nbk,
                                             float* __restrict__ SP, float* __restrict__ LinvTP,
                                             unsigned short* __restrict__ SP16, const float* __restrict__ cst, float* __restrict__ spf, int own_qscale) {
    __shared__ float Ls[2][16][129];                         // the 16 rows of L_r of row-block bi / bk, 128 columns at a time
    const int i = threadIdx.x >> 4, k = threadIdx.x & 15;
    const int off = (16 * (k >> 2) + i) * 4 + (k & 3);       // A-fragment order: lane 16g + i holds G[i][4g + s]
    const int nS = R * nbk * nbk;
    if (b < nS) {
        const int bi = b / (R * nbk), r = (b / nbk) % R, bk = b % nbk;
        const int bm = bi < bk ? bi : bk, ncol = 16 * bm + 16;
        const float* Lr = q_sqrt + (size_t)r * M * M;
        // The split-f16 image's scale: max |L_r| rounded up to a power of two -- as the precompute's q(u) role left it in the state's constant
        // block, or (IWVI_BW_OWN_QSCALE, round 6) from L_r ITSELF, formed here by every workgroup (16 K floats, one coalesced pass; a maximum
        // is order-free) and rounded exactly as that role rounds it: the same scale bit for bit, without tying this launch behind a precompute
        // of the CURRENT q(u) -- a training step's second op waited a cross-queue join for it.  (+2 us of this launch: only where it is
        // off the critical path.)  Always read from the state: the factorisation's own 2^ea (cst[IWVI_CST_SA]).
        float max_l = 0.f;
        if (SP16 && !own_qscale) max_l = 16384.f * cst[IWVI_CST_FR + r] * cst[IWVI_CST_SA];     // max|L_r| < 2^(14 - e_r)
        if (SP16 && own_qscale) {
            __shared__ float rs[4];
            float mx = 0.f;
            {   // 16 bytes per load, the row / column of an entry stepped along instead of divided out (M is a multiple of 16 here)
                const float4* L4 = reinterpret_cast<const float4*>(Lr);
                int row = (int)(threadIdx.x * 4) / M, col = (int)(threadIdx.x * 4) - row * M;
                for (int i4 = threadIdx.x; i4 < (M * M) >> 2; i4 += 256) {
                    const float4 v = L4[i4];
                    if (col <= row) mx = fmaxf(mx, fabsf(v.x));
                    if (col + 1 <= row) mx = fmaxf(mx, fabsf(v.y));
                    if (col + 2 <= row) mx = fmaxf(mx, fabsf(v.z));
                    if (col + 3 <= row) mx = fmaxf(mx, fabsf(v.w));
                    col += 1024;
                    while (col >= M) { col -= M; ++row; }
                }
            }
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            if ((threadIdx.x & 63) == 0) rs[threadIdx.x >> 6] = mx;
            __syncthreads();
            mx = fmaxf(fmaxf(rs[0], rs[1]), fmaxf(rs[2], rs[3]));
            __syncthreads();
            const int er = mx > 0.f ? 13 - ilogbf(mx) : 0;                                   // (precompute_dev.h: role_pack_body)
            max_l = ldexpf(1.f, 14 - er);                                                    // max |L_r| < 2^(14 - e_r)
        }
        float acc = 0.f;
        for (int c0 = 0; c0 < ncol; c0 += 128) {
            const int nc = ncol - c0 < 128 ? ncol - c0 : 128;
            if (c0) __syncthreads();
            for (int idx = threadIdx.x; idx < 16 * nc; idx += 256) {
                const int rr = idx / nc, c = c0 + idx - rr * nc;
                Ls[0][rr][c - c0] = (c <= 16 * bi + rr) ? Lr[(size_t)(16 * bi + rr) * M + c] : 0.f;     // tril: zero above the diagonal
                Ls[1][rr][c - c0] = (c <= 16 * bk + rr) ? Lr[(size_t)(16 * bk + rr) * M + c] : 0.f;
            }
            __syncthreads();
            for (int j = 0; j < nc; ++j) acc = fmaf(Ls[0][i][j], Ls[1][k][j], acc);
        }
        SP[(size_t)b * 256 + off] = acc;
        if (SP16) {
            const int es = 14 - (int)ceilf(log2f(fmaxf((float)M * max_l * max_l, 1e-30f)));
            const float x = acc * ldexpf(1.f, es);
            const _Float16 h1 = (_Float16)x, h2 = (_Float16)(x - (float)h1);
            const int kc = bk >> 1, k32 = 16 * (bk & 1) + k, ln = 16 * (k32 >> 3) + i, jj = k32 & 7;
            _Float16* dst = reinterpret_cast<_Float16*>(SP16) + ((size_t)(bi * R + r) * (nbk >> 1) + kc) * 1024;
            dst[ln * 8 + jj] = h1; dst[512 + ln * 8 + jj] = h2;
            if (bi == 0 && bk == 0 && threadIdx.x == 0) spf[r] = ldexpf(1.f, -es) / cst[IWVI_CST_SA];
        }
        return;
    }
    b -= nS;                                                  // upper block (bi, bk >= bi) of Lm^-T: entry [i][k] = Lm^-1[16bk + k][16bi + i]
    int bi = 0;
    while (b >= nbk - bi) { b -= nbk - bi; ++bi; }
    const int bk = bi + b;
    LinvTP[((size_t)tri_upper_off(nbk, bi) + (bk - bi)) * 256 + off] = (float)Linv64[(size_t)(16 * bk + k) * Mp + 16 * bi + i];
}
__global__ __launch_bounds__(256) void k_pack_bw(const float* __restrict__ q_sqrt, const double* __restrict__ Linv64, int Mp, int M, int R, int nbk,
                                                 float* __restrict__ SP, float* __restrict__ LinvTP,
                                                 unsigned short* __restrict__ SP16, const float* __restrict__ cst, float* __restrict__ spf, int own_qscale) {
    pack_bw_body((int)blockIdx.x, q_sqrt, Linv64, Mp, M, R, nbk, SP, LinvTP, SP16, cst, spf, own_qscale);
}
static int chain_ns_cap(long long T, int cap) {             // samples per workgroup / 16: as the forward's (every CU a workgroup), then down to a divisor of T
    int ns = (int)((T + 16 * 256 - 1) / (16 * 256));
    if (ns > cap) ns = cap;
    if (ns < 1) ns = 1;
    while (ns > 1 && T % (16 * ns)) --ns;
    return ns;
}
// the conservative choice (two [M x 16 NS] float tiles + staging in 160 KB of LDS for any D, R, P): what the workspace is sized for
// phase 1 of the chain on split-f16 operands: an even number of 16-row blocks (the state then carries the scales); a third tile holds
// the a planes, so beyond M = 256 only 16 samples fit a workgroup -- still faster than the fp32 phase 1 at 32 (configs[4]: 277 -> 240 ms per
// value + gradient)
// IWVI_BW_F32_CHAIN of the descriptor being served by this thread's current call (set at the entry points that take a descriptor;
// the sizing functions, which take none, see the default -- the split-f16 chain needs the larger workspace)
static thread_local bool t_bw_f32_chain = false;
struct BwFlagScope { bool old; explicit BwFlagScope(int flags) : old(t_bw_f32_chain) { t_bw_f32_chain = (flags & IWVI_BW_F32_CHAIN) != 0; } ~BwFlagScope() { t_bw_f32_chain = old; } };
static bool chain_s16(int M, int Mp) { return Mp == M && ((Mp / 16) & 1) == 0 && !t_bw_f32_chain; }
static int chain_ns(long long T, int M = 128) { return chain_ns_cap(T, M <= 128 ? 5 : ((M > 256 && chain_s16(M, M)) ? 1 : 2)); }
static bool chain_ok(int M, int Mp, long long T) {
    // M > 256: only with 32 samples per workgroup (two [M x 32] tiles; the scaled inducing inputs then stay in L2) -- at 16 every packed
    // S_r block (R * 32 * 32 KiB per layer) would be fetched from L2 for 4 MFMAs: measured 383 ms per value + gradient at configs[4]
    // against 359 ms on the GEMM path
    return Mp == M && M <= 512 && (T % 16) == 0 && (M <= 256 || chain_ns(T, M) >= 2 || chain_s16(M, Mp));
}
// floats of the staging region beside the two tiles: what is staged there before (heads) / after (kernel adjoint: x~ rows, z~, shares)
static bool chain_z_lds(int M) { return M <= 256; }
static int chain_dsz(int NSAMP, int M, int D, int R, int P, int DM, bool z_lds = true) {
    const int heads = 3 * NSAMP * P + P * R + 4 * NSAMP * R + D * P, adj = NSAMP * round_up(D + 2, 4) + (z_lds ? M * DM : 0) + 16 * NSAMP;
    return ((heads > adj ? heads : adj) + 3) & ~3;
}
// LDS bytes of k_bw_chain for a layer shape and a tile row stride ts (float4)
static size_t chain_lds_bytes_ts(int NSAMP, int ts, int M, int D, int R, int P) {
    const int DM = D <= 8 ? 8 : (D <= 16 ? 16 : 32);
    return sizeof(float) * ((size_t)(chain_s16(M, round_up(M, 16)) ? 3 : 2) * ts * M + (size_t)chain_dsz(NSAMP, M, D, R, P, DM, M <= 256) + (size_t)M * R + (size_t)NSAMP * (2 * R + 1) + (size_t)NSAMP * D
                            + (size_t)NSAMP * DM + DM + (size_t)NSAMP * (D + 2));
}
// the padded row stride where it fits, else the plain one
static int chain_ts(int NSAMP, int M, int D, int R, int P) { return chain_lds_bytes_ts(NSAMP, NSAMP + 4, M, D, R, P) <= 160 * 1024 ? NSAMP + 4 : NSAMP; }
// samples per workgroup / 16 for a layer shape: 128 < M <= 256 takes 64 samples where the shape's tiles and staging fit (every packed
// S_r block fetched from L2 then feeds 16 MFMAs instead of 8), else the conservative choice.  Never fewer workgroups than chain_ns()
// implies more partial sums: the workspace (sized with chain_ns) covers both.
static int chain_ns_shape(long long T, int M, int D, int R, int P) {
    const int base = chain_ns(T, M);
    if (M > 128 && M <= 256) {
        const int ns4 = chain_ns_cap(T, 4);
        if (ns4 > base && chain_lds_bytes_ts(16 * ns4, 16 * ns4, M, D, R, P) <= 160 * 1024) return ns4;
    }
    return base;
}
static size_t chain_lds_bytes(long long T, int M, int D, int R, int P) {
    const int NSAMP = 16 * chain_ns_shape(T, M, D, R, P);
    return chain_lds_bytes_ts(NSAMP, chain_ts(NSAMP, M, D, R, P), M, D, R, P);
}
static bool chain_fits(long long T, int M, int Mp, int D, int R, int P) { return chain_ok(M, Mp, T) && chain_lds_bytes(T, M, D, R, P) <= 160 * 1024; }
// the two M x M products over samples inside the chain kernel: only while the number of per-workgroup shares stays moderate
static bool chain_products_ok(int M, long long T) {
    return chain_ok(M, round_up(M, 16), T) && M <= 128 && T / (16 * chain_ns(T, M)) <= 1024;
}
template <int NS>
static int launch_chain_ns(hipStream_t st, ChainArgs a) {
    constexpr int NSAMP = 16 * NS;
    const int DM = a.D <= 8 ? 8 : (a.D <= 16 ? 16 : 32);
    a.z_lds = chain_z_lds(a.M) ? 1 : 0;
    a.s16 = chain_s16(a.M, a.Mp) ? 1 : 0;
    a.p5h = (a.s16 && !dbg_opt("IWVI_BW_P5_F32")) ? 1 : 0;
    a.dsz = chain_dsz(NSAMP, a.M, a.D, a.R, a.P, DM, a.M <= 256);
    a.ts = chain_ts(NSAMP, a.M, a.D, a.R, a.P);
    const size_t lds = chain_lds_bytes_ts(NSAMP, a.ts, a.M, a.D, a.R, a.P);
    static bool done = false;
    if (!done) {
        const size_t most = 160 * 1024;
        const void* fns[] = {(const void*)k_bw_chain<NS, 8>, (const void*)k_bw_chain<NS, 16>, (const void*)k_bw_chain<NS, 32>};
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)most) != hipSuccess) { set_error("hipFuncSetAttribute(k_bw_chain)"); return IWVI_ERR_LAUNCH; }
        done = true;
    }
    if (lds > 160 * 1024) { set_error("k_bw_chain: %zu B of LDS", lds); return IWVI_ERR_UNSUPPORTED; }
    const dim3 grid((unsigned)(a.T / NSAMP)), block(512);
    if (a.D <= 8) hipLaunchKernelGGL((k_bw_chain<NS, 8>), grid, block, lds, st, a);
    else if (a.D <= 16) hipLaunchKernelGGL((k_bw_chain<NS, 16>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((k_bw_chain<NS, 32>), grid, block, lds, st, a);
    return check_launch("k_bw_chain");
}
static int launch_chain(hipStream_t st, const ChainArgs& a) {
    switch (chain_ns_shape(a.T, a.M, a.D, a.R, a.P)) {
        case 1: return launch_chain_ns<1>(st, a);
        case 2: return launch_chain_ns<2>(st, a);
        case 3: return launch_chain_ns<3>(st, a);
        case 4: return launch_chain_ns<4>(st, a);
        default: return launch_chain_ns<5>(st, a);
    }
}
// dq_sqrt[r] = tril(sym(G_r) L_r) + add_coef * (L_r - diag(1 / L_ii)),  G_r given by its lower triangle (the reduced split-K SYRK):
// 16x16 output tile per workgroup, operands through LDS, k from the tile's first column (L_r is lower triangular).
__global__ __launch_bounds__(256) void k_gl_tril(const float* __restrict__ G, const float* __restrict__ Lq, float* __restrict__ out, int M, double add_coef) {
    __shared__ float As[16][17], Bs[16][17];
    const int r = blockIdx.z;
    const float* Gr = G + (size_t)r * M * M; const float* Lr = Lq + (size_t)r * M * M; float* o = out + (size_t)r * M * M;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16, i = i0 + ty, j = j0 + tx;
    if (j0 > i0 + 15) { if (i < M && j < M) o[(size_t)i * M + j] = 0.f; return; }
    float s = 0.f;
    for (int k0 = j0; k0 < M; k0 += 16) {
        const int ka = k0 + tx, kb = k0 + ty;
        As[ty][tx] = (i < M && ka < M) ? (ka <= i ? Gr[(size_t)i * M + ka] : Gr[(size_t)ka * M + i]) : 0.f;     // symmetric read
        Bs[ty][tx] = (kb < M && j < M && kb >= j) ? Lr[(size_t)kb * M + j] : 0.f;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) s = fmaf(As[ty][k], Bs[k][tx], s);
        __syncthreads();
    }
    if (i >= M || j >= M) return;
    if (j > i) { o[(size_t)i * M + j] = 0.f; return; }
    const double v = (double)Lr[(size_t)i * M + j];
    o[(size_t)i * M + j] = (float)((double)s + add_coef * (v - (i == j ? 1.0 / v : 0.0)));
}

// Thin sums over samples: part[blk][m][n] = sum_{t in chunk} X[t*ldx + m] * Y(t, n), n < N + ones, N <= 64; the extra
// column (ones) is the plain column sum.  Thread = column m, rows in a fixed order; chunks summed by k_reduce_parts.
constexpr int THIN_ROWS = 64;
struct ThinArgs { const float* X; long long ldx; const float* Y; long long ldy; int N, ones; long long T; int M; float* part; int chunks; };
static int thin_chunks(long long T) { return (int)((T + (long long)THIN_ROWS * 1024 - 1) / ((long long)THIN_ROWS * 1024)); }   // 64-row chunks per workgroup: <= 1024 partials
template <int NM>                       // N <= NM: the accumulators stay in registers
__global__ __launch_bounds__(256) void k_thin(ThinArgs a) {
    // chunk of 64 rows per workgroup: its rows of Y staged in LDS, wave w takes rows 16w..16w+15, lanes over the columns
    // of X (64 at a time), the four waves' partial sums combined through LDS in a fixed order
    __shared__ float ys[THIN_ROWS * (NM > 1 ? NM : 1)];
    __shared__ float red[4][64][NM + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NN = a.N + a.ones;
    for (int mb = 0; mb < a.M; mb += 64) {
        const int m = mb + lane;
        float acc[NM], ones = 0.f;
#pragma unroll
        for (int n = 0; n < NM; ++n) acc[n] = 0.f;
        for (int ch = 0; ch < a.chunks; ++ch) {                      // this workgroup's 64-row chunks, in order
            const long long t0 = ((long long)blockIdx.x * a.chunks + ch) * THIN_ROWS;
            if (t0 >= a.T) break;
            const int nrows = (int)((a.T - t0) < THIN_ROWS ? (a.T - t0) : THIN_ROWS);
            __syncthreads();                                         // the previous chunk's readers are done with ys
            if (a.N > 0) {
                // rows beyond the end are zero-filled: they meet x = 0 below, and 0 * (stale LDS bits) could be NaN
                if (a.ldy == a.N) for (int i = tid; i < THIN_ROWS * a.N; i += 256) ys[i] = i < nrows * a.N ? a.Y[t0 * a.ldy + i] : 0.f;
                else for (int i = tid; i < THIN_ROWS * a.N; i += 256) { const int r = i / a.N; ys[i] = r < nrows ? a.Y[(t0 + r) * a.ldy + (i - r * a.N)] : 0.f; }
            }
            __syncthreads();
            float x[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { const int r = 16 * wave + i; x[i] = (m < a.M && r < nrows) ? a.X[(t0 + r) * a.ldx + m] : 0.f; }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = 16 * wave + i;
#pragma unroll
                for (int n = 0; n < NM; ++n) if (n < a.N) acc[n] = fmaf(x[i], ys[r * a.N + n], acc[n]);
                ones += x[i];
            }
        }
#pragma unroll
        for (int n = 0; n < NM; ++n) red[wave][lane][n] = acc[n];
        red[wave][lane][NM] = ones;
        __syncthreads();
        for (int i = tid; i < 64 * NN; i += 256) {
            const int l = i / NN, n = i - l * NN, c = (n < a.N) ? n : NM;
            if (mb + l < a.M)
                a.part[((size_t)blockIdx.x * a.M + mb + l) * NN + n] = ((red[0][l][c] + red[1][l][c]) + red[2][l][c]) + red[3][l][c];
        }
        __syncthreads();
    }
}

// out[m, n] = sum_t X[t, m] Y[t, n] (n < N) and, with ones, out[m, N] = sum_t X[t, m];  out is [M, N + ones]
static int thin(hipStream_t st, const float* X, long long ldx, int M, const float* Y, long long ldy, int N, int ones, long long T,
                float* part_ws, size_t part_floats, float* out, int ldo = 0, ReduceQueue* rq = nullptr,
                const float* add = nullptr, double add_coef = 0.0) {
    if (ldo == 0) ldo = N + ones;
    if (N > 32) {                                          // columns of Y in two passes (LDS budget of k_thin)
        int rc = thin(st, X, ldx, M, Y, ldy, 32, 0, T, part_ws, part_floats, out, ldo, rq, add, add_coef);
        if (rc != IWVI_OK) return rc;
        return thin(st, X, ldx, M, Y + 32, ldy, N - 32, ones, T, part_ws, part_floats, out + 32, ldo, rq, add ? add + 32 : nullptr, add_coef);
    }
    const int chunks = thin_chunks(T);
    const int nblk = (int)((T + (long long)THIN_ROWS * chunks - 1) / ((long long)THIN_ROWS * chunks)), NN = N + ones;
    if (rq) {
        part_ws = rq->take((size_t)nblk * M * NN);
        if (!part_ws) { set_error("backward: thin-reduction workspace too small"); return IWVI_ERR_ARG; }
    } else if ((size_t)nblk * M * NN > part_floats) { set_error("backward: thin-reduction workspace too small"); return IWVI_ERR_ARG; }
    ThinArgs a{X, ldx, Y, ldy, N, ones, T, M, part_ws, chunks};
    const dim3 grid(nblk), block(256);
    if (N <= 1) hipLaunchKernelGGL(k_thin<1>, grid, block, 0, st, a);
    else if (N <= 8) hipLaunchKernelGGL(k_thin<8>, grid, block, 0, st, a);
    else if (N <= 16) hipLaunchKernelGGL(k_thin<16>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(k_thin<32>, grid, block, 0, st, a);
    ReduceArgs r{part_ws, nblk, M, NN, out, nullptr, ldo, 1.0, 0.0, 0, 0, add, add_coef, 0};
    if (rq && rq->push(r, 1)) return check_launch("k_thin");
    hipLaunchKernelGGL(k_reduce_parts, dim3((M * NN + 63) / 64, 1), dim3(256), 0, st, r);
    return check_launch("k_thin");
}

// ------------------------------------------------------------------------------------------------------------
// float64 side: small dense products for the Cholesky adjoint, K_uu's own gradient, final assembly
// ------------------------------------------------------------------------------------------------------------
// C[i, j] = sum_k A(i, k) B(k, j); post = 1: Phi (strict upper -> 0, diagonal halved).  16x16 tiles through LDS.
__global__ __launch_bounds__(256) void k_dmm(const double* A, long long a_si, long long a_sk, const double* B, long long b_sk, long long b_sj,
                                             double* C, int n, int ldc, int post) {
    __shared__ double As[16][17], Bs[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
    double s = 0.0;
    for (int k0 = 0; k0 < n; k0 += 16) {
        // load so that the contiguous index of each operand runs along tx
        if (a_sk == 1) As[ty][tx] = (i < n && k0 + tx < n) ? A[i * a_si + (k0 + tx)] : 0.0;
        else { const int ii = blockIdx.y * 16 + tx, kk = k0 + ty; As[tx][ty] = (ii < n && kk < n) ? A[ii * a_si + kk * a_sk] : 0.0; }
        if (b_sj == 1) Bs[ty][tx] = (k0 + ty < n && j < n) ? B[(k0 + ty) * b_sk + j] : 0.0;
        else { const int jj = blockIdx.x * 16 + ty, kk = k0 + tx; Bs[tx][ty] = (jj < n && kk < n) ? B[kk * b_sk + jj * b_sj] : 0.0; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) s = fma(As[ty][k], Bs[k][tx], s);
        __syncthreads();
    }
    if (i >= n || j >= n) return;
    if (post == 1) s = (j > i) ? 0.0 : (j == i ? 0.5 * s : s);
    C[(size_t)i * ldc + j] = s;
}
__device__ __forceinline__ void prep_body(int bid, const float* Z, const float* ls, const double* Linv64, int Mp, float* Zt, float* invls, float* LinvF, int M, int D) {
    const int idx = bid * 256 + threadIdx.x;
    if (idx < D) invls[idx] = 1.f / ls[idx];
    if (idx < M * D) Zt[idx] = Z[idx] / ls[idx % D];
    if (idx < M * M) { const int i = idx / M, j = idx - i * M; LinvF[idx] = j <= i ? (float)Linv64[(size_t)i * Mp + j] : 0.f; }
}
__global__ __launch_bounds__(256) void k_prep(const float* Z, const float* ls, const double* Linv64, int Mp, float* Zt, float* invls, float* LinvF, int M, int D) {
    prep_body((int)blockIdx.x, Z, ls, Linv64, Mp, Zt, invls, LinvF, M, D);
}
// the parameter-only operands of EVERY layer's adjoint in one launch (iwvi_gp_layers_backward_prepare): the four small launches of a
// two-layer model queue up behind the layer kernel, which holds every CU -- as one they are over before the ELBO tail's adjoint is
struct PrepOne {
    const float* Z; const float* ls; const double* Linv64; const float* q_sqrt; const float* cst;
    float* Zt; float* invls; float* LinvF; float* SP; float* LinvTP; unsigned short* SP16; float* spf;
    int Mp, M, D, R, nbk, nprep, npack, own_qscale;
};
struct PrepAll { PrepOne L[IWVI_MAX_STACK]; int n; };
__global__ __launch_bounds__(256) void k_prepare_all(const PrepAll a) {
    int b = (int)blockIdx.x;
    for (int li = 0; li < a.n; ++li) {
        const PrepOne& L = a.L[li];
        if (b < L.nprep) { prep_body(b, L.Z, L.ls, L.Linv64, L.Mp, L.Zt, L.invls, L.LinvF, L.M, L.D); return; }
        b -= L.nprep;
        if (b < L.npack) { pack_bw_body(b, L.q_sqrt, L.Linv64, L.Mp, L.M, L.R, L.nbk, L.SP, L.LinvTP, L.SP16, L.cst, L.spf, L.own_qscale); return; }
        b -= L.npack;
    }
}
// row m of K_uu: dZ~_uu[m, :] = 4 sum_n Sbar_mn dK_mn/dd2 (z~_m - z~_n),  dvar_m = sum_n Sbar_mn K_mn / s2,  Sbar = (S + S^T)/2
__global__ __launch_bounds__(256) void k_kuu_bwd(const float* Zt, const double* S, int M, int D, double variance_, const float* var_dev, int kern_type, double* dZt_uu, double* dvar_m) {
    // one WAVE per row m, lanes over n.  The D + 1 sums over n of a row (dZ~[m][d], the variance term) are reduced TOGETHER: every lane parks
    // its D + 1 partial sums in LDS, then lane j < D + 1 adds the 64 partials of sum j in lane order -- one LDS round trip and a chain of 64
    // adds for all of them, where a shuffle tree per sum was 12 cross-lane moves of a double each (D + 1 = 10 of them: 3 us of an 11 us launch).
    // Fixed order: deterministic.  (round 6; the symmetrised S is read as S[m][n] + S[n][m]: the second one strided, from L2)
    __shared__ double part[4][16][65];                        // (16 sums per round: one round up to D = 15)
    const double variance = var_dev ? (double)*var_dev : variance_;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, m = blockIdx.x * 4 + w;
    if (m >= M) return;                                        // (wave-uniform; no workgroup barrier below)
    double acc[IWVI_MAX_D + 1];
#pragma unroll
    for (int d = 0; d <= IWVI_MAX_D; ++d) acc[d] = 0.0;
    for (int n = lane; n < M; n += 64) {
        double d2 = 0.0;
        for (int d = 0; d < D; ++d) { const double e = (double)Zt[m * D + d] - (double)Zt[n * D + d]; d2 += e * e; }
        double k, g;
        kern_and_grad<double>(d2, kern_type, variance, k, g);
        const double sb = 0.5 * (S[(size_t)m * M + n] + S[(size_t)n * M + m]);
        const double wgt = 4.0 * sb * g;                        // both (m, n) and (n, m)
        acc[IWVI_MAX_D] += sb * k / variance;
#pragma unroll
        for (int d = 0; d < IWVI_MAX_D; ++d) if (d < D) acc[d] += wgt * ((double)Zt[m * D + d] - (double)Zt[n * D + d]);
    }
    for (int j0 = 0; j0 <= D; j0 += 16) {                      // sums j0 .. j0 + 15 of (dZ~[m][0..D-1] | variance term)
#pragma unroll
        for (int d = 0; d <= IWVI_MAX_D; ++d) {
            const int j = d == IWVI_MAX_D ? D : d;              // (the variance term's accumulator is the last register, its sum index is D)
            if ((d < D || d == IWVI_MAX_D) && j >= j0 && j < j0 + 16) part[w][j - j0][lane] = acc[d];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS writes have landed (its own region: no workgroup barrier)
        __builtin_amdgcn_wave_barrier();
        const int j = j0 + lane;
        if (lane < 16 && j <= D) {
            double s = 0.0;
            for (int l = 0; l < 64; ++l) s += part[w][lane][l];
            if (j < D) dZt_uu[m * D + j] = s; else dvar_m[m] = s;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}
struct FinalArgsB {
    const float* Z; const float* ls; const float* q_mu; const float* q_sqrt; const float* Zt; const float* invls;
    const float* colsumC; const float* CtF; const float* sums; const float* dinvls_x; const double* dZt_uu; const double* dvar_m;
    float* dZ; float* dls; float* dvariance; float* dq_mu; float* dq_sqrt;
    int M, D, R; double kl_weight, variance; const float* var_dev;
    // mixing matrix / linear mean function (workgroup D + 1, when asked for): dW[p, r] = S1 + S2 + 2 W o S3 with S1 = dFs^T G, S2 = dFm^T MU,
    // S3 = dFv^T V;  dA = F^T dFs + F^T dFm
    const float *s1, *s2, *s3, *W; float* dW; int n_w; const float *a1, *a2; float* dA; int n_a;
};
// workgroup d < D: dZ[:, d] and dls[d]; workgroup D: dvariance.  One wave each, lanes over m, fixed reduction tree.
__global__ __launch_bounds__(64) void k_bw_final(FinalArgsB f) {
    const int d = blockIdx.x, lane = threadIdx.x;
    auto wsum = [&](double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; };
    if (d < f.D) {
        const double il = (double)f.invls[d];
        double s = 0.0;
        for (int m = lane; m < f.M; m += 64) {
            const double zt = f.Zt[m * f.D + d];
            const double dzt = 2.0 * zt * (double)f.colsumC[m * (f.D + 1) + f.D] - 2.0 * il * (double)f.CtF[m * (f.D + 1) + d] + f.dZt_uu[m * f.D + d];
            if (f.dZ) f.dZ[m * f.D + d] = (float)(dzt * il);
            s += dzt * (double)f.Z[m * f.D + d];
        }
        s = wsum(s) + (double)f.dinvls_x[d];
        if (lane == 0 && f.dls) f.dls[d] = (float)(-s * il * il);
    } else if (d > f.D) {                                      // (round 6: the former k_lin_combine launch) dW = S1 + S2 + 2 W o S3, dA = A1 + A2
        for (int i = lane; i < f.n_w || i < f.n_a; i += 64) {
            if (f.dW && i < f.n_w) f.dW[i] = (f.s1 ? f.s1[i] : 0.f) + (f.s2 ? f.s2[i] : 0.f) + (f.s3 ? 2.f * f.W[i] * f.s3[i] : 0.f);
            if (f.dA && i < f.n_a) f.dA[i] = (f.a1 ? f.a1[i] : 0.f) + (f.a2 ? f.a2[i] : 0.f);
        }
    } else if (f.dvariance) {
        double s = 0.0;
        for (int m = lane; m < f.M; m += 64) s += f.dvar_m[m];
        s = wsum(s) + (double)f.sums[0] + (double)f.sums[1] / (f.var_dev ? (double)*f.var_dev : f.variance);     // + sum_t sum_r dv_r  +  sum k dk / s2
        if (lane == 0) f.dvariance[0] = (float)s;
    }
}
struct BwdWs {
    float *DMU, *DV2, *SDV, *DA, *DK, *Qx, *part, *LinvF, *Zt, *invls, *CtF1, *Qsum;
    double *Lbar, *T1, *T2, *S, *dZt_uu, *dvar_m;
    float *GMV, *lin;                   // [T, 3R];  3 [P, R] + 2 [D, P] partial results
    float *SP, *LinvTP, *G;             // k_bw_chain's packed operands; G_r = A^T diag(2 dv_r) A [R, M, M]
    unsigned short* SP16; float* spf;   // split-f16 image of S_r and its per-r factors (even nbk)
    size_t part_floats, bytes;
};
static BwdWs bwd_layout(char* base, long long T, int M, int D, int R) {
    BwdWs w; size_t o = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o = align256(o + bytes); return p; };
    const int nsplit = (int)((T + splitk_chunk(T) - 1) / splitk_chunk(T)) + 2;
    w.DMU = (float*)take(sizeof(float) * T * R); w.DV2 = (float*)take(sizeof(float) * T * R); w.SDV = (float*)take(sizeof(float) * T);
    w.DA = (float*)take(sizeof(float) * T * M); w.DK = (float*)take(sizeof(float) * T * M);
    w.Qx = (float*)take(sizeof(float) * T * (D + 2));
    w.part_floats = (size_t)nsplit * M * M * (R + 1);          // dLm + the R batched dL_r, parked together
    if (chain_products_ok(M, T)) {                               // the chain kernel's per-workgroup shares of dLm and G_r instead
        const size_t need = (size_t)(T / (16 * chain_ns(T, M))) * M * M * (R + 1);
        if (need > w.part_floats) w.part_floats = need;
    }
    if (chain_ok(M, round_up(M, 16), T)) {                        // + its per-workgroup shares of the thin sums
        const size_t S = (size_t)(T / (16 * chain_ns(T, M)));
        w.part_floats += S * ((size_t)M * (R + D + 1) + D + 2 + 3 * (size_t)IWVI_MAX_P * R + 2 * (size_t)D * IWVI_MAX_P) + 64 * 8;
    }
    {   // thin reductions: ceil(T / THIN_ROWS) chunks of at most [max(M, 34)][33]
        const size_t thin = (size_t)((T + (long long)THIN_ROWS * thin_chunks(T) - 1) / ((long long)THIN_ROWS * thin_chunks(T))) * (M > 34 ? M : 34) * 33;
        w.part_floats += 8 * thin + 1024;                      // + up to 8 thin products
    }
    w.part = (float*)take(sizeof(float) * w.part_floats);
    w.LinvF = (float*)take(sizeof(float) * M * M); w.Zt = (float*)take(sizeof(float) * M * D); w.invls = (float*)take(sizeof(float) * IWVI_MAX_D);
    w.CtF1 = (float*)take(sizeof(float) * M * (D + 1)); w.Qsum = (float*)take(sizeof(float) * (IWVI_MAX_D + 2));
    w.Lbar = (double*)take(sizeof(double) * M * M); w.T1 = (double*)take(sizeof(double) * M * M); w.T2 = (double*)take(sizeof(double) * M * M);
    w.S = (double*)take(sizeof(double) * M * M); w.dZt_uu = (double*)take(sizeof(double) * M * D); w.dvar_m = (double*)take(sizeof(double) * M);
    w.GMV = (float*)take(sizeof(float) * T * 3 * R);
    w.lin = (float*)take(sizeof(float) * (3 * IWVI_MAX_P * IWVI_MAX_R + 2 * IWVI_MAX_D * IWVI_MAX_P));
    {
        const int Mp = round_up(M, 16), nbk = Mp / 16;
        w.SP = (float*)take(sizeof(float) * (size_t)R * nbk * nbk * 256);
        w.SP16 = (unsigned short*)take((size_t)R * nbk * ((nbk + 1) / 2) * 2048); w.spf = (float*)take(sizeof(float) * IWVI_MAX_R);
        w.LinvTP = (float*)take(sizeof(float) * (size_t)tri_blocks(nbk) * 256);
        w.G = (float*)take(sizeof(float) * (size_t)R * M * M);
    }
    w.bytes = o;
    return w;
}


// ------------------------------------------------------------------------------------------------------------
// ELBO tail (models.py:134-150), one thread per data point: L_nk, softmax over the K samples, heads of the final layer
// ------------------------------------------------------------------------------------------------------------
struct ElboBwdArgs {
    const float* fmean; const float* fvar; const float* Y; int Dy;
    const float* kl[IWVI_MAX_KL]; int kl_dims[IWVI_MAX_KL]; int n_kl;
    long long B; int K; float lik_var; const float* lik_var_dev; double scale; int mode_vi;
    const float* lse_global; int K_total;   // K-sharded: logsumexp over ALL the job's samples of each point (after the exchange)
    float* w; float* d_mean; float* d_var; double* part;   // part[0..B) = lse - log K, part[B..2B) = d lik_var share
};
__global__ __launch_bounds__(256) void k_elbo_bwd(ElboBwdArgs a) {      // one wave per data point, lanes over its K samples
    const int lane = threadIdx.x & 63;
    const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= a.B) return;
    const float s = a.lik_var_dev ? *a.lik_var_dev : a.lik_var, c0 = -0.5f * logf(6.283185307179586f * s);
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
    double se = 0.0;
    if (a.mode_vi) {                                    // models.py:84: mean over the samples -> uniform weights
        for (int k = lane; k < a.K; k += 64) se += (double)logw(b * a.K + k);
        for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
    } else if (a.lse_global) {                          // weights against the whole job's normaliser: exp(L - LSE)
        mx = a.lse_global[b]; se = 1.0;
    } else {
        for (int k = lane; k < a.K; k += 64) mx = fmaxf(mx, logw(b * a.K + k));
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        for (int k = lane; k < a.K; k += 64) se += (double)__expf(logw(b * a.K + k) - mx);
        for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
    }
    double ds = 0.0;
    for (int k = lane; k < a.K; k += 64) {
        const long long t = b * a.K + k;
        const float wt = a.mode_vi ? (float)(a.scale / (double)a.K) : (float)(a.scale * (double)__expf(logw(t) - mx) / se);
        if (a.w) a.w[t] = wt;
        for (int j = 0; j < a.Dy; ++j) {
            const float e = a.Y[b * a.Dy + j] - a.fmean[t * a.Dy + j], v = a.fvar[t * a.Dy + j];
            if (a.d_mean) a.d_mean[t * a.Dy + j] = wt * e / s;
            if (a.d_var) a.d_var[t * a.Dy + j] = -0.5f * wt / s;
            ds += (double)wt * (-0.5 / (double)s + 0.5 * ((double)e * e + (double)v) / ((double)s * s));
        }
    }
    for (int o = 32; o > 0; o >>= 1) ds += __shfl_xor(ds, o, 64);
    if (lane == 0) { a.part[b] = a.mode_vi ? se / (double)a.K : (double)mx + log(se) - log((double)(a.lse_global ? a.K_total : a.K)); a.part[a.B + b] = ds; }
}
// out[0] = sum part[0..n), out[1] = sum part[n..2n), out[2] = scale * out[0] - sum of the global KL shares (the bound)
struct ElboFinishArgs { const double* part; long long n; double scale; const double* klg[IWVI_MAX_LAYERS]; int kln[IWVI_MAX_LAYERS]; int n_glob; double* out; };
__global__ __launch_bounds__(256) void k_elbo_finish(ElboFinishArgs a) {
    __shared__ double red[256];
    double tot[2];
    for (int i = 0; i < 2; ++i) {
        const double* p = a.part + (size_t)i * a.n;
        double s = 0.0;
        for (long long k = threadIdx.x; k < a.n; k += 256) s += p[k];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
        tot[i] = red[0];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double kl = 0.0;
        for (int i = 0; i < a.n_glob; ++i) for (int q = 0; q < a.kln[i]; ++q) kl += a.klg[i][q];
        a.out[0] = tot[0]; a.out[1] = tot[1]; a.out[2] = a.scale * tot[0] - kl;
    }
}

// LatentVariableLayer (layers.py:83-103): W = mu + eps sigma; d(enc_out) [B, 2 Lw] = sum over the K samples of (dmu | draw)
struct LvBwdArgs {
    const float* mu; const float* sigma; int ld_enc, raw; const float* eps; const float* dFn; int ld, col0;
    const float* w; int Lw; long long B; int K, sampled; float* d_out;
};
// (point b, latent dim l): lanes over the K samples, partials combined by a fixed shuffle tree (deterministic) -> (d mu, d raw) in every lane
__device__ __forceinline__ void lv_bwd_pair(const LvBwdArgs& a, long long b, int l, int lane, float& o_mu, float& o_raw) {
    const float mu = a.mu[b * a.ld_enc + l];
    float sg = a.sigma[b * a.ld_enc + l];
    if (a.raw) sg = softplus_f(sg - 3.f);
    float dmu = 0.f, dsg = 0.f;
    for (int k = lane; k < a.K; k += 64) {
        const long long t = b * a.K + k;
        const float e = a.eps[t * a.Lw + l], W = fmaf(e, sg, mu);
        const float dfw = a.dFn ? a.dFn[t * a.ld + a.col0 + l] : 0.f;
        const float dkl = a.w ? -a.w[t] : 0.f;                 // L_nk contains -kl (models.py:141-142)
        if (a.sampled) { const float dW = dfw + dkl * W; dmu += dW; dsg += dW * e - dkl / sg; }
        else { dmu += dfw + dkl * mu; dsg += dfw * e + dkl * (sg - 1.f / sg); }
    }
    for (int o = 32; o > 0; o >>= 1) { dmu += __shfl_xor(dmu, o, 64); dsg += __shfl_xor(dsg, o, 64); }
    o_mu = dmu;
    o_raw = dsg * (1.f - __expf(-sg));                          // sigma = softplus(raw - 3): d sigma / d raw = 1 - exp(-sigma)
}
__global__ __launch_bounds__(256) void k_lv_bwd(LvBwdArgs a) {
    // one wave per (point, latent dim), lanes over its K samples (a thread per point walking K samples is a chain of K
    // dependent global round trips)
    const int lane = threadIdx.x & 63;
    const long long idx = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= a.B * a.Lw) return;
    const long long b = idx / a.Lw; const int l = (int)(idx - b * a.Lw);
    float dmu, draw;
    lv_bwd_pair(a, b, l, lane, dmu, draw);
    if (lane == 0) {
        a.d_out[b * 2 * a.Lw + l] = dmu;
        a.d_out[b * 2 * a.Lw + a.Lw + l] = draw;
    }
}

// Encoder MLP (layers.py:137-152), one thread per row: activations of every layer -> acts, then deltas (d / d pre-activation)
struct EncBwdArgs {
    const float* XY; long long rows; const float* W[IWVI_MAX_ENC]; const float* b[IWVI_MAX_ENC]; int dims[IWVI_MAX_ENC + 1]; int n;
    const float* d_out; int act;
    float* part; int woff[IWVI_MAX_ENC], boff[IWVI_MAX_ENC], ptot;     // part[workgroup][ptot]: this workgroup's share of (dW_l | db_l)
    LvBwdArgs lv; int fused;                                           // fused: d_out rows come from the latent-variable layer's adjoint, formed here
};
constexpr int ER = 4, ELD = 65;         // rows per workgroup, row stride of an activation tile in LDS
constexpr int EW_MAX = IWVI_MAX_ENC * (64 * 64 + 64);   // LDS copy of the weights and biases (widths <= 64)
// 8 rows per workgroup (128 workgroups at B = 1024: the phase is latency-bound), activations, deltas AND the weights in LDS
// (weights read from global memory inside the dot-product loops cost an L1 round trip per term), threads over (row, unit)
__global__ __launch_bounds__(256) void k_enc_bwd(EncBwdArgs a) {
    extern __shared__ float esm[];
    const int tid = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * ER;
    const int nrows = (int)((a.rows - row0) < ER ? (a.rows - row0) : ER);
    float* acts = esm;                                   // [n + 1][ER][ELD]
    float* cur = esm + (size_t)(a.n + 1) * ER * ELD;     // d / d layer output
    float* dl = cur + ER * ELD;                          // d / d pre-activation
    float* prev = dl + ER * ELD;
    float* wl = prev + ER * ELD;                         // weights | biases of every layer, laid out like a workgroup's share of the
    //                                                      parameter gradients: W_l at woff[l], b_l at boff[l] (ptot floats in all)
    // Prologue: every global read is issued before the first value is used (round 5: layer by layer, each copy loop a round trip of its
    // own -- weights x n, input rows, d_out: six dependent round trips in a 15 us kernel).
    constexpr int CH = 8;                                    // parameters per thread held in registers (n ptot <= 2048: one pass)
    float wv[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int idx = tid + 256 * c;
        wv[c] = 0.f;
        if (idx < a.ptot) {
            int l = 0;
            while (l + 1 < a.n && idx >= a.woff[l + 1]) ++l;
            wv[c] = idx < a.boff[l] ? a.W[l][idx - a.woff[l]] : (a.b[l] ? a.b[l][idx - a.boff[l]] : 0.f);
        }
    }
    float xv[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int idx = tid + 256 * c;
        xv[c] = 0.f;
        if (idx < nrows * a.dims[0]) { const int r = idx / a.dims[0], i = idx - r * a.dims[0]; xv[c] = a.XY[(row0 + r) * a.dims[0] + i]; }
    }
    const int dlast = a.dims[a.n];
    if (a.fused) {
        // d / d (means | raw) of this workgroup's rows: the latent-variable layer's adjoint (k_lv_bwd's arithmetic, same shuffle tree)
        const int lane = tid & 63, wave = tid >> 6;
        for (int p = wave; p < nrows * a.lv.Lw; p += 4) {
            const int r = p / a.lv.Lw, l = p - r * a.lv.Lw;
            float dmu, draw;
            lv_bwd_pair(a.lv, row0 + r, l, lane, dmu, draw);
            if (lane == 0) { cur[r * ELD + l] = dmu; cur[r * ELD + a.lv.Lw + l] = draw; }
        }
    } else {
        for (int idx = tid; idx < nrows * dlast; idx += 256) { const int r = idx / dlast, o = idx - r * dlast; cur[r * ELD + o] = a.d_out[(row0 + r) * dlast + o]; }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) { const int idx = tid + 256 * c; if (idx < a.ptot) wl[idx] = wv[c]; }
    for (int idx = tid + 256 * CH; idx < a.ptot; idx += 256) {      // (wider encoders: the rest, pass by pass)
        int l = 0;
        while (l + 1 < a.n && idx >= a.woff[l + 1]) ++l;
        wl[idx] = idx < a.boff[l] ? a.W[l][idx - a.woff[l]] : (a.b[l] ? a.b[l][idx - a.boff[l]] : 0.f);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int idx = tid + 256 * c;
        if (idx < nrows * a.dims[0]) { const int r = idx / a.dims[0], i = idx - r * a.dims[0]; acts[r * ELD + i] = xv[c]; }
    }
    __syncthreads();
    for (int l = 0; l < a.n; ++l) {
        const int din = a.dims[l], dout = a.dims[l + 1];
        const float* in = acts + (size_t)l * ER * ELD; float* out = acts + (size_t)(l + 1) * ER * ELD;
        for (int idx = tid; idx < nrows * dout; idx += 256) {
            const int r = idx / dout, o = idx - r * dout;
            float acc = wl[a.boff[l] + o];
            for (int i = 0; i < din; ++i) acc = fmaf(in[r * ELD + i], wl[a.woff[l] + i * dout + o], acc);
            if (l < a.n - 1) acc = enc_act(acc, a.act);
            if (din == dout) acc += in[r * ELD + o];
            out[r * ELD + o] = acc;
        }
        __syncthreads();
    }
    // Backward sweep, ONE barrier per layer (round 6; three before -- delta, products, copy -- and the phases between them are a handful of LDS
    // round trips each: the launch is a chain of barriers): the thread that forms d / d input [r][i] of layer l applies the activation's
    // derivative of layer l - 1 to it on the spot (that IS layer l - 1's delta), and the two row tiles alternate instead of being copied.
    float* dl2 = wl + a.ptot;                                // the other delta tile (behind the weights)
    {
        const int dout = a.dims[a.n];                        // the last layer has no activation: its delta is d / d output
        for (int idx = tid; idx < nrows * dout; idx += 256) { const int r = idx / dout, o = idx - r * dout; dl[r * ELD + o] = cur[r * ELD + o]; }
        __syncthreads();
    }
    for (int l = a.n - 1; l >= 0; --l) {
        const int din = a.dims[l], dout = a.dims[l + 1];
        const bool skip = din == dout;
        const float* in = acts + (size_t)l * ER * ELD;
        // this workgroup's rows' share of dW_l = in^T dl and db_l = colsum(dl), rows in a fixed order
        float* pw = a.part + (size_t)blockIdx.x * a.ptot;
        for (int idx = tid; idx < din * dout + dout; idx += 256) {
            float acc = 0.f;
            if (idx < din * dout) {
                const int i = idx / dout, o = idx - i * dout;
                for (int r = 0; r < nrows; ++r) acc = fmaf(in[r * ELD + i], dl[r * ELD + o], acc);
                pw[a.woff[l] + idx] = acc;
            } else {
                const int o = idx - din * dout;
                for (int r = 0; r < nrows; ++r) acc += dl[r * ELD + o];
                pw[a.boff[l] + o] = acc;
            }
        }
        if (l > 0) {
            const bool skip_b = a.dims[l - 1] == din;            // layer l - 1's residual connection
            const float* in_b = acts + (size_t)(l - 1) * ER * ELD;
            for (int idx = tid; idx < nrows * din; idx += 256) {
                const int r = idx / din, i = idx - r * din;
                float acc = skip ? cur[r * ELD + i] : 0.f;
                for (int o = 0; o < dout; ++o) acc = fmaf(dl[r * ELD + o], wl[a.woff[l] + i * dout + o], acc);
                prev[r * ELD + i] = acc;                         // d / d output of layer l - 1 (its residual branch reads it next round)
                const float av = in[r * ELD + i] - (skip_b ? in_b[r * ELD + i] : 0.f);
                dl2[r * ELD + i] = acc * enc_act_grad(av, a.act);
            }
            __syncthreads();
            float* t_ = cur; cur = prev; prev = t_;
            t_ = dl; dl = dl2; dl2 = t_;
        }
    }
}

// every encoder parameter = sum over the workgroups' shares (fixed order), scattered to its tensor
struct EncReduceArgs { const float* part; int nblk, ptot, n; int off[2 * IWVI_MAX_ENC], len[2 * IWVI_MAX_ENC]; float* dst[2 * IWVI_MAX_ENC]; };
__global__ __launch_bounds__(256) void k_enc_reduce(EncReduceArgs a) {
    // 64 outputs per workgroup; the shares of an output are summed by 4 threads (contiguous quarters, in order), then combined
    // in a fixed order (as k_reduce_parts): deterministic, and four times shorter than one serial pass over the workgroups
    __shared__ double red[4][64];
    const int o = threadIdx.x & 63, gq = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + o;
    double s = 0.0;
    if (idx < a.ptot) {
        const int per = (a.nblk + 3) / 4, k0 = gq * per, k1 = (k0 + per < a.nblk) ? k0 + per : a.nblk;
#pragma unroll 8
        for (int b = k0; b < k1; ++b) s += (double)a.part[(size_t)b * a.ptot + idx];
    }
    red[gq][o] = s;
    __syncthreads();
    if (gq != 0 || idx >= a.ptot) return;
    s = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
    for (int j = 0; j < a.n; ++j)
        if (idx >= a.off[j] && idx < a.off[j] + a.len[j]) { if (a.dst[j]) a.dst[j][idx - a.off[j]] = (float)s; return; }
}

// ------------------------------------------------------------------------------------------------------------
// Optimiser steps of experiments/build_models.py:284-304: natural gradient on the final layer's (q_mu, q_sqrt)
// (GPflow NatGradOptimizer, natural parameterisation) and Adam on everything else (TensorFlow AdamOptimizer on
// GPflow's unconstrained variables).  float64 for the natural-gradient algebra, like the reference.
// ------------------------------------------------------------------------------------------------------------
// C[i, j] = alpha * sum_k A(i, k) B(k, j) + beta * E[i, j];  post 1: Phi, post 2: nothing
struct DmmArgs { const double* A; long long a_si, a_sk; const double* B; long long b_sk, b_sj; double* C; long long ldc;
                 int I, J, K; double alpha; const double* E; long long lde; double beta; int post; };
__global__ __launch_bounds__(256) void k_dmm2(DmmArgs a) {               // 16x16 output tile per workgroup, operands through LDS
    __shared__ double As[16][17], Bs[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
    double s = 0.0;
    for (int k0 = 0; k0 < a.K; k0 += 16) {
        if (a.a_sk == 1) As[ty][tx] = (i < a.I && k0 + tx < a.K) ? a.A[i * a.a_si + (k0 + tx)] : 0.0;
        else { const int ii = blockIdx.y * 16 + tx, kk = k0 + ty; As[tx][ty] = (ii < a.I && kk < a.K) ? a.A[ii * a.a_si + kk * a.a_sk] : 0.0; }
        if (a.b_sj == 1 || a.J == 1) Bs[ty][tx] = (k0 + ty < a.K && j < a.J) ? a.B[(k0 + ty) * a.b_sk + j * a.b_sj] : 0.0;
        else { const int jj = blockIdx.x * 16 + ty, kk = k0 + tx; Bs[tx][ty] = (jj < a.J && kk < a.K) ? a.B[kk * a.b_sk + jj * a.b_sj] : 0.0; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) s = fma(As[ty][k], Bs[k][tx], s);
        __syncthreads();
    }
    if (i >= a.I || j >= a.J) return;
    s *= a.alpha;
    if (a.E) s += a.beta * a.E[i * a.lde + j];
    if (a.post == 1) s = (j > i) ? 0.0 : (j == i ? 0.5 * s : s);
    a.C[i * a.ldc + j] = s;
}
// The same product on v_mfma_f64_16x16x4_f64: one WAVE per 16x16 output tile (lane l feeds A[l & 15][4kk + (l >> 4)] and
// B[4kk + (l >> 4)][l & 15]; accumulator register e of lane l is C[(l >> 4) + 4e][l & 15]).  The operands of 32 k-steps are
// requested together (one L2 round trip per 128 of K), so a 128^3 product is ~2 us of latency instead of the LDS kernel's 8
// barriers-and-loads rounds per tile (10-13 us); used where I, J are multiples of 16 and K of 4, J > 1.
using bw_f64x4 = __attribute__((ext_vector_type(4))) double;
__global__ __launch_bounds__(64) void k_dmm_mfma(DmmArgs a) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    bw_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    const double* Ap = a.A + (long long)(i0 + r) * a.a_si + (long long)g * a.a_sk;
    const double* Bp = a.B + (long long)g * a.b_sk + (long long)(j0 + r) * a.b_sj;
    for (int k0 = 0; k0 < a.K; k0 += 128) {                   // 32 k-steps' operands requested together: ONE L2 round trip per 128 of K
        double av[32], bv[32];                               // (8 at a time was four dependent round trips for a 128^3 product: 7 us of latency)
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int k = k0 + 4 * u;
            const bool in = k + g < a.K;
            av[u] = in ? Ap[(long long)k * a.a_sk] : 0.0;
            bv[u] = in ? Bp[(long long)k * a.b_sk] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = i0 + g + 4 * e, j = j0 + r;
        double s = acc[e] * a.alpha;
        if (a.E) s += a.beta * a.E[(long long)i * a.lde + j];
        if (a.post == 1) s = (j > i) ? 0.0 : (j == i ? 0.5 * s : s);
        a.C[(long long)i * a.ldc + j] = s;
    }
}
static void dmm(hipStream_t st, const double* A, long long a_si, long long a_sk, const double* B, long long b_sk, long long b_sj,
                double* C, long long ldc, int I, int J, int K, double alpha = 1.0, const double* E = nullptr, long long lde = 0, double beta = 0.0, int post = 0) {
    DmmArgs a{A, a_si, a_sk, B, b_sk, b_sj, C, ldc, I, J, K, alpha, E, lde, beta, post};
    if (I % 16 == 0 && J % 16 == 0 && K % 4 == 0) {
        hipLaunchKernelGGL(k_dmm_mfma, dim3(J / 16, I / 16), dim3(64), 0, st, a);
        return;
    }
    hipLaunchKernelGGL(k_dmm2, dim3((J + 15) / 16, (I + 15) / 16), dim3(256), 0, st, a);
}
// X = L^-1 for n <= TRI_SMALL in ONE workgroup, 16x16 blocks of X in LDS: the diagonal blocks by substitution (a lane per
// column), then block diagonal d = 1, 2, ...: X(i, i-d) = -X(i, i) * sum_{k=i-d}^{i-1} L(i, k) X(k, i-d); one barrier per d.
constexpr int TRI_SMALL = 128;     // (two copies of the lower blocks in LDS: 2 * 36 * 2 KiB + scratch at n = 128)
__global__ __launch_bounds__(256) void k_tri_inv_small(const double* L, double* X, int n) {
    extern __shared__ double xs[];                       // lower-triangular blocks, row-block major: block (i, j) at tri(i) + j
    const int nb = (n + 15) / 16, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto blk = [&](int i, int j) { return xs + (size_t)(i * (i + 1) / 2 + j) * 256; };
    // L's lower blocks staged in LDS first (same block layout, identity padding): element reads from global memory inside the
    // substitution loops were a chain of dependent L2 round trips (~100 us for n = 128)
    double* ls = xs + (size_t)(nb * (nb + 1) / 2) * 256;
    {   // one element of every block per thread, eight blocks' loads in flight at a time (one load per iteration is a chain of
        // ~36 L2 round trips: most of this kernel's 66 us before)
        const int nblk = nb * (nb + 1) / 2, er = tid >> 4, ec = tid & 15;
        for (int b0 = 0; b0 < nblk; b0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = (b0 + u < nblk) ? b0 + u : nblk - 1;
                int bi = 0; while ((bi + 1) * (bi + 2) / 2 <= b) ++bi;
                const int bj = b - bi * (bi + 1) / 2, r = 16 * bi + er, c = 16 * bj + ec;
                v[u] = (r < n && c < n) ? L[(size_t)r * n + c] : (r == c ? 1.0 : 0.0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) if (b0 + u < nblk) ls[(size_t)(b0 + u) * 256 + tid] = v[u];
        }
    }
    __syncthreads();
    auto Lel = [&](int r, int c) { return ls[(size_t)((r >> 4) * ((r >> 4) + 1) / 2 + (c >> 4)) * 256 + (r & 15) * 16 + (c & 15)]; };
    double* scratch = ls + (size_t)(nb * (nb + 1) / 2) * 256 + wave * 256;
    for (int i = wave; i < nb; i += 4) {
        if (lane < 16) {
            double x[16];
            const int j = lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                double acc = (r == j) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) if (k < r && k >= j) acc = fma(-Lel(16 * i + r, 16 * i + k), x[k], acc);
                x[r] = (r >= j) ? acc / Lel(16 * i + r, 16 * i + r) : 0.0;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) blk(i, i)[r * 16 + j] = x[r];
        }
    }
    __syncthreads();
    // off-diagonal blocks by v_mfma_f64_16x16x4_f64 (lane l feeds A[l & 15][4kk + (l >> 4)] and B[4kk + (l >> 4)][l & 15]; accumulator
    // register e of lane l is C[(l >> 4) + 4e][l & 15]): a 16^3 block product is 4 MFMAs instead of a serial 16 x 4 fma loop per lane
    // (the block diagonals late in the sweep were chains of ~100 dependent LDS round trips per block: 66 us at n = 128)
    auto lblk = [&](int i, int j) { return ls + (size_t)(i * (i + 1) / 2 + j) * 256; };
    const int mr = lane & 15, mg = lane >> 4;
    for (int d = 1; d < nb; ++d) {
        for (int i = d + wave; i < nb; i += 4) {
            const int j = i - d;
            // T = sum_k L(i, k) X(k, j), k = j .. i-1
            bw_f64x4 t = {0.0, 0.0, 0.0, 0.0};
            for (int k = j; k < i; ++k) {
                const double* A = lblk(i, k); const double* B = blk(k, j);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) t = __builtin_amdgcn_mfma_f64_16x16x4f64(A[mr * 16 + 4 * kk + mg], B[(4 * kk + mg) * 16 + mr], t, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) scratch[(mg + 4 * e) * 16 + mr] = t[e];
            __builtin_amdgcn_wave_barrier();
            // X(i, j) = -X(i, i) T
            bw_f64x4 o = {0.0, 0.0, 0.0, 0.0};
            const double* xi = blk(i, i);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) o = __builtin_amdgcn_mfma_f64_16x16x4f64(-xi[mr * 16 + 4 * kk + mg], scratch[(4 * kk + mg) * 16 + mr], o, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) blk(i, j)[(mg + 4 * e) * 16 + mr] = o[e];
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
    }
    for (int idx = tid; idx < n * n; idx += 256) {
        const int r = idx / n, c = idx - r * n;
        X[idx] = (c <= r) ? blk(r >> 4, c >> 4)[(r & 15) * 16 + (c & 15)] : 0.0;
    }
}
static int tri_inverse(hipStream_t st, const double* L, double* X, int n);
// X = L^-1 for lower-triangular L [n, n] (row-major): one wave per column j solves L x = e_j by forward substitution,
// x in LDS, each row's dot product spread over the lanes; zeros above the diagonal
__global__ __launch_bounds__(64) void k_tri_inv(const double* L, double* X, int n) {
    __shared__ double x[IWVI_MAX_M];
    const int j = blockIdx.x, lane = threadIdx.x;
    for (int i = lane; i < j; i += 64) X[(size_t)i * n + j] = 0.0;
    for (int i = j; i < n; ++i) {
        double s = 0.0;
        for (int k = j + lane; k < i; k += 64) s = fma(L[(size_t)i * n + k], x[k], s);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) {
            const double v = ((i == j ? 1.0 : 0.0) - s) / L[(size_t)i * n + i];
            x[i] = v; X[(size_t)i * n + j] = v;
        }
        __syncthreads();
    }
}
static int tri_inverse(hipStream_t st, const double* L, double* X, int n) {
    if (n <= TRI_SMALL) {
        const int nb = (n + 15) / 16;
        const size_t lds = sizeof(double) * ((size_t)(nb * (nb + 1) / 2) * 512 + 4 * 256);
        static bool done = false;
        if (!done) {
            const int nbm = (TRI_SMALL + 15) / 16;
            const size_t most = sizeof(double) * ((size_t)(nbm * (nbm + 1) / 2) * 512 + 4 * 256);
            hipError_t e = hipFuncSetAttribute((const void*)k_tri_inv_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)most);
            if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", most, hipGetErrorString(e)); return IWVI_ERR_LAUNCH; }
            done = true;
        }
        hipLaunchKernelGGL(k_tri_inv_small, dim3(1), dim3(256), lds, st, L, X, n);
    } else {
        hipLaunchKernelGGL(k_tri_inv, dim3(n), dim3(64), 0, st, L, X, n);
    }
    return check_launch("triangular inverse");
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

// Adam on GPflow's unconstrained variables.  transform 1 = positive: p = softplus(x) + 1e-6 (gpflow.transforms.Log1pe)
struct AdamTensor { float* p; const float* g; float* x; float* m; float* v; long long n; int transform; int g64; };
constexpr int ADAM_MAX = 48;
struct AdamArgs { AdamTensor t[ADAM_MAX]; int n; float lr_t, b1, b2, eps, sign; int init; const long long* t_dev; float lr; };
__global__ void k_adam(AdamArgs a) {
    const AdamTensor& T = a.t[blockIdx.y];
    float lr_t = a.lr_t;                                     // (a local: writing to the by-value argument block would copy all of it to scratch)
    if (a.t_dev) {                                           // bias correction from the device-resident step count (this step = *t_dev + 1):
        __shared__ float lr_sh;                              // two float64 pow() per workgroup, not per thread
        if (threadIdx.x == 0) {
            const double t = (double)(*a.t_dev + 1);
            lr_sh = (float)((double)a.lr * sqrt(1.0 - pow((double)a.b2, t)) / (1.0 - pow((double)a.b1, t)));
        }
        __syncthreads();
        lr_t = lr_sh;
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < T.n; i += (long long)gridDim.x * blockDim.x) {
        if (a.init) {
            const float p = T.p[i];
            float x = p;
            if (T.transform == 1) { const float y = p - 1e-6f; x = y > 20.f ? y : logf(expm1f(y)); }
            T.x[i] = x; T.m[i] = 0.f; T.v[i] = 0.f;
            continue;
        }
        float x = T.x[i], g = a.sign * (T.g64 ? (float)reinterpret_cast<const double*>(T.g)[i] : T.g[i]);
        if (T.transform == 1) g *= 1.f - __expf(-(T.p[i] - 1e-6f));          // d softplus(x) / dx = sigmoid(x)
        const float m = a.b1 * T.m[i] + (1.f - a.b1) * g;
        const float v = a.b2 * T.v[i] + (1.f - a.b2) * g * g;
        x -= lr_t * m / (sqrtf(v) + a.eps);
        T.m[i] = m; T.v[i] = v; T.x[i] = x;
        T.p[i] = (T.transform == 1) ? (x > 20.f ? x : log1pf(__expf(x))) + 1e-6f : x;
    }
}
__global__ void k_inc_i64(long long* p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p += 1; }

}  // namespace iwvi

using namespace iwvi;

// The two sizing entries take no descriptor, so they answer for BOTH arithmetic modes a later call may ask for through its descriptor's
// flags (IWVI_BW_F32_CHAIN): u is needed when either mode's adjoint reads it, the workspace is the larger of the two layouts.
extern "C" int iwvi_gp_layer_backward_needs_u(int64_t T, int M, int D, int R, int P) {
    if (T <= 0 || M <= 0 || D <= 0 || R <= 0 || P <= 0) return 1;
    bool fits_s16, fits_f32;
    { const BwFlagScope sc(0); fits_s16 = chain_fits(T, M, round_up(M, 16), D, R, P); }
    { const BwFlagScope sc(IWVI_BW_F32_CHAIN); fits_f32 = chain_fits(T, M, round_up(M, 16), D, R, P); }
    return (fits_s16 && fits_f32) ? 0 : 1;
}

extern "C" size_t iwvi_gp_layer_backward_ws_bytes(int64_t T, int M, int D, int R) {
    if (T <= 0 || M <= 0 || D <= 0 || R <= 0) return 0;
    size_t b_s16, b_f32;
    { const BwFlagScope sc(0); b_s16 = bwd_layout(nullptr, T, M, D, R).bytes; }
    { const BwFlagScope sc(IWVI_BW_F32_CHAIN); b_f32 = bwd_layout(nullptr, T, M, D, R).bytes; }
    return b_s16 > b_f32 ? b_s16 : b_f32;
}

// What the adjoint of a layer needs besides the forward's outputs: scaled inducing inputs, a float32 Lm^-1, and for the
// streaming chain the packed S_r / Lm^-T operands.  Depends only on the parameters and the dense factors, so a caller may
// queue it early on another stream (beside the forward) and pass desc.prepared = 1 to iwvi_gp_layer_backward.
extern "C" int iwvi_gp_layer_backward_prepare(const iwvi_gp_bwd_desc* dp, int64_t T, void* ws_, void* stream_) {
    if (!dp || !ws_ || T <= 0) { set_error("iwvi_gp_layer_backward_prepare: bad argument"); return IWVI_ERR_ARG; }
    const iwvi_gp_bwd_desc& d = *dp;
    const BwFlagScope flag_scope(d.flags);
    if (!d.state || !d.Z || !d.lengthscales || !d.q_sqrt || d.M <= 0 || d.M > IWVI_MAX_M || d.D <= 0 || d.D > IWVI_MAX_D || d.R <= 0 || d.R > IWVI_MAX_R) {
        set_error("iwvi_gp_layer_backward_prepare: null input or size out of range"); return IWVI_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream_;
    const int M = d.M, D = d.D, R = d.R;
    const StateLayout sl = state_layout(M, R);
    const int Mp = sl.Mp;
    const double* Linv64 = (const double*)((const char*)d.state + sl.off_Linv);
    BwdWs w = bwd_layout((char*)ws_, T, M, D, R);
    const int n = M * M > M * D ? M * M : M * D;
    hipLaunchKernelGGL(k_prep, dim3((n + 255) / 256), dim3(256), 0, st, d.Z, d.lengthscales, Linv64, Mp, w.Zt, w.invls, w.LinvF, M, D);
    if (chain_fits(T, M, Mp, D, R, d.P > 0 ? d.P : R)) {
        const int nbk = Mp / 16;
        const bool s16 = chain_s16(M, Mp);
        hipLaunchKernelGGL(k_pack_bw, dim3((unsigned)(R * nbk * nbk + tri_blocks(nbk))), dim3(256), 0, st, d.q_sqrt, Linv64, Mp, M, R, nbk, w.SP, w.LinvTP,
                           s16 ? w.SP16 : (unsigned short*)nullptr, (const float*)((const char*)d.state + sl.off_cst), w.spf, (d.flags & IWVI_BW_OWN_QSCALE) ? 1 : 0);
    }
    return check_launch("iwvi_gp_layer_backward_prepare");
}

extern "C" int iwvi_gp_layers_backward_prepare(const iwvi_gp_bwd_desc* descs, int n, int64_t T, void* const* ws, void* stream_) {
    if (!descs || !ws || n <= 0 || n > IWVI_MAX_STACK || T <= 0) { set_error("iwvi_gp_layers_backward_prepare: bad argument"); return IWVI_ERR_ARG; }
    PrepAll a{};
    a.n = n;
    unsigned grid = 0;
    for (int i = 1; i < n; ++i)                                    // (one launch for all layers: one arithmetic mode)
        if ((descs[i].flags ^ descs[0].flags) & IWVI_BW_F32_CHAIN) { set_error("iwvi_gp_layers_backward_prepare: layer %d asks for another arithmetic mode (IWVI_BW_F32_CHAIN) than layer 0", i); return IWVI_ERR_ARG; }
    const BwFlagScope flag_scope(descs[0].flags);
    for (int i = 0; i < n; ++i) {
        const iwvi_gp_bwd_desc& d = descs[i];
        if (!ws[i] || !d.state || !d.Z || !d.lengthscales || !d.q_sqrt || d.M <= 0 || d.M > IWVI_MAX_M || d.D <= 0 || d.D > IWVI_MAX_D || d.R <= 0 || d.R > IWVI_MAX_R) {
            set_error("iwvi_gp_layers_backward_prepare: layer %d: null input or size out of range", i); return IWVI_ERR_ARG;
        }
        const int M = d.M, D = d.D, R = d.R;
        const StateLayout sl = state_layout(M, R);
        BwdWs w = bwd_layout((char*)ws[i], T, M, D, R);
        PrepOne& L = a.L[i];
        L.Z = d.Z; L.ls = d.lengthscales; L.Linv64 = (const double*)((const char*)d.state + sl.off_Linv); L.q_sqrt = d.q_sqrt;
        L.cst = (const float*)((const char*)d.state + sl.off_cst);
        L.Zt = w.Zt; L.invls = w.invls; L.LinvF = w.LinvF; L.SP = w.SP; L.LinvTP = w.LinvTP; L.spf = w.spf;
        L.Mp = sl.Mp; L.M = M; L.D = D; L.R = R; L.nbk = sl.Mp / 16;
        const int nn = M * M > M * D ? M * M : M * D;
        L.nprep = (nn + 255) / 256;
        const bool chain = chain_fits(T, M, sl.Mp, D, R, d.P > 0 ? d.P : R);
        L.npack = chain ? R * L.nbk * L.nbk + tri_blocks(L.nbk) : 0;
        L.SP16 = (chain && chain_s16(M, sl.Mp)) ? w.SP16 : (unsigned short*)nullptr;
        L.own_qscale = (d.flags & IWVI_BW_OWN_QSCALE) ? 1 : 0;
        grid += (unsigned)(L.nprep + L.npack);
    }
    hipLaunchKernelGGL(k_prepare_all, dim3(grid), dim3(256), 0, (hipStream_t)stream_, a);
    return check_launch("iwvi_gp_layers_backward_prepare");
}

extern "C" int iwvi_gp_layer_backward(const iwvi_gp_bwd_desc* dp, int64_t T, void* ws_, void* stream_) {
    if (!dp || !ws_ || T <= 0) { set_error("iwvi_gp_layer_backward: bad argument"); return IWVI_ERR_ARG; }
    const iwvi_gp_bwd_desc& d = *dp;
    const BwFlagScope flag_scope(d.flags);
    if (!d.state || !d.Z || !d.lengthscales || !d.q_mu || !d.q_sqrt || !d.F || !d.A) { set_error("iwvi_gp_layer_backward: null input"); return IWVI_ERR_ARG; }
    if (d.M <= 0 || d.M > IWVI_MAX_M || d.D <= 0 || d.D > IWVI_MAX_D || d.R <= 0 || d.R > IWVI_MAX_R || d.P <= 0 || d.P > IWVI_MAX_P || T >= (1LL << 31) / (d.M > d.D ? d.M : d.D)) {
        set_error("iwvi_gp_layer_backward: size out of range"); return IWVI_ERR_ARG;
    }
    if (d.kern_type != IWVI_KERN_RBF && d.kern_type != IWVI_KERN_MATERN52) { set_error("iwvi_gp_layer_backward: unknown kernel type %d", d.kern_type); return IWVI_ERR_UNSUPPORTED; }
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
    // Only dq_mu / dq_sqrt asked for (the natural-gradient op, build_models.py:288-295, moves nothing else): on the streaming chain's
    // shapes that is the heads + two sums over samples -- no da / dk, no kernel adjoint, no adjoint of the factorisation, and
    // nothing of iwvi_gp_layer_backward_prepare (the state may then be the packed one: the dense factors are not read).
    const bool q_any = !d.dZ && !d.dls && !d.dvariance && !d.dF && !d.dW && !d.dmf_A;
    const bool q_only = q_any && d.GMV && chain_fits(T, M, Mp, D, R, d.P);
    // desc.phase splits the call at the point where dF is queued: 1 = the per-sample chain only, 2 = the parameter branch only (same
    // descriptor, same workspace; the caller orders 2 after 1 -- on any stream -- and may queue other work in between: a captured
    // graph then keeps the layers' chains on one hardware queue).  Shapes off the streaming chain do everything in phase 1.
    const int phase = (q_only || !(d.GMV && chain_fits(T, M, Mp, D, R, d.P))) ? (d.phase == 2 ? -1 : 0) : d.phase;
    if (phase == -1) return IWVI_OK;
    if (phase < 0 || phase > 2) { set_error("iwvi_gp_layer_backward: phase %d", d.phase); return IWVI_ERR_ARG; }
    if (phase != 2 && !q_any && !d.prepared && (rc = iwvi_gp_layer_backward_prepare(dp, T, ws_, stream_)) != IWVI_OK) return rc;
    const float* gmv = d.GMV ? d.GMV : w.GMV;
    // deferred reductions of this layer: every product over samples parks its partial sums in its own slice of the workspace
    const bool prod = d.GMV && chain_products_ok(M, T);    // dLm and G_r shares come out of the chain kernel
    size_t partA = (size_t)((T + splitk_chunk(T) - 1) / splitk_chunk(T) + 2) * M * M;
    if (prod && (size_t)(T / (16 * chain_ns(T, M))) * M * M > partA) partA = (size_t)(T / (16 * chain_ns(T, M))) * M * M;
    const bool two_q = d.side_stream && d.side_stream2 && d.side_stream2 != d.side_stream;
    ReduceQueue rqA(w.part, two_q ? partA : 0), rqB(w.part + (two_q ? partA : 0), w.part_floats - (two_q ? partA : 0));
    float* s[3] = {nullptr, nullptr, nullptr}; float* a12[2] = {nullptr, nullptr};
    const bool lin_on = (d.dW && d.W) || (d.dmf_A && d.mf_type == IWVI_MF_LINEAR);
    // streaming chain (no saved u_r): M a multiple of 16 up to 128, T a multiple of 64, the forward's gmv block at hand
    const bool chain = d.GMV && chain_fits(T, M, Mp, D, R, d.P);
    if (!chain && !d.U) { set_error("iwvi_gp_layer_backward: this shape (M=%d, T=%lld) takes the GEMM path, which needs the forward's u_out", M, (long long)T); return IWVI_ERR_ARG; }
    if (chain) {
        const int nbk = Mp / 16;
        ChainArgs ca{d.GMV, d.noise, d.W, d.mf_A, d.d_sample, d.d_mean, d.d_var, w.DMU, w.DV2, w.SDV, d.dF, d.P, d.mf_type,
                     d.A, Mp, d.q_mu, w.SP, w.LinvTP, w.DK, d.F, w.Zt, w.invls, w.DA, w.Qx, (long long)T, M, D, R, nbk, d.variance, d.kern_type,
                     dbg_opt("IWVI_CHAIN_EXIT"), d.variance_dev};
        // the thin sums over samples ride in the chain kernel: one partial per workgroup and job, summed with the rest of chain B
        const int S = (int)(T / (16 * chain_ns_shape(T, M, D, R, d.P))), P = d.P;
        auto job = [&](int Mj, int Nj, float* out, const float* add, double add_coef) -> float* {
            float* pp = rqB.take((size_t)S * Mj * Nj);
            if (!pp) return nullptr;
            ReduceArgs r{pp, S, Mj, Nj, out, nullptr, (long long)Nj, 1.0, 0.0, 0, 0, add, add_coef, 0};
            return rqB.push(r, 1) ? pp : nullptr;
        };
        // (- kl_weight * dKL/dq_mu = - kl_weight * q_mu rides in the reduction; temp_workaround.py:186-188)
        ca.p_qmu = job(M, R, d.dq_mu ? d.dq_mu : w.DMU, d.dq_mu ? d.q_mu : nullptr, -d.kl_weight);
        ca.q_only = q_only ? 1 : 0;
        ca.SP16 = (const float*)w.SP16; ca.spf = w.spf;
        ca.ZtP = (const float*)((const char*)d.state + sl.off_ZtP); ca.cst = (const float*)((const char*)d.state + sl.off_cst); ca.nsteps = round_up(D + 2, 4) / 4;
        if (!q_only) { ca.p_ctf = job(M, D + 1, w.CtF1, nullptr, 0.0); ca.p_q = job(D + 2, 1, w.Qsum, nullptr, 0.0); }
        if (d.dW && d.W) { ca.p_w = job(3 * P, R, w.lin, nullptr, 0.0); for (int i = 0; i < 3; ++i) s[i] = w.lin + (size_t)i * P * R; if (!ca.p_w) ca.p_qmu = nullptr; }
        if (d.dmf_A && d.mf_type == IWVI_MF_LINEAR) { ca.p_a = job(2 * D, P, w.lin + 3 * IWVI_MAX_P * IWVI_MAX_R, nullptr, 0.0); for (int i = 0; i < 2; ++i) a12[i] = w.lin + 3 * IWVI_MAX_P * IWVI_MAX_R + (size_t)i * D * P; if (!ca.p_a) ca.p_qmu = nullptr; }
        if (!ca.p_qmu || (!q_only && (!ca.p_ctf || !ca.p_q))) { set_error("backward: workspace too small for the chain kernel's partial sums"); return IWVI_ERR_ARG; }
        ca.S = S;
        if (prod) {                                        // dLm and G_r shares too: no DK / A round trip, no split-K launches
            ReduceQueue& qa = two_q ? rqA : rqB;
            if (!q_only) {
                ca.p_lm = qa.take((size_t)S * M * M);
                ReduceArgs rl{ca.p_lm, S, M, M, nullptr, w.Lbar, (long long)M, -1.0, 0.0, 1, 0, nullptr, 0.0, 0, nbk};      // (shares: lower blocks, accumulator images)
                if (!ca.p_lm || !qa.push(rl, 1)) { set_error("backward: workspace too small for the chain kernel's dLm shares"); return IWVI_ERR_ARG; }
            }
            if (d.dq_sqrt) {
                ca.p_g = rqB.take((size_t)R * S * M * M);
                ReduceArgs rg{ca.p_g, S, M, M, w.G, nullptr, (long long)M, 1.0, 0.0, 1, (long long)M * M, nullptr, 0.0, 0, nbk};
                if (!ca.p_g || !rqB.push(rg, R)) { set_error("backward: workspace too small for the chain kernel's G_r shares"); return IWVI_ERR_ARG; }
            }
            ca.DK = nullptr;
        }
        if (phase != 2 && (rc = launch_chain(st, ca)) != IWVI_OK) return rc;
        if (phase == 1) return IWVI_OK;
        if (q_only) {                                      // the two reductions (+ G_r by split-K GEMM where the chain does not form it), then dL_r
            if (d.dq_sqrt && !prod) {
                GemmArgs q{};
                q.A = d.A; q.a_sm = 1; q.a_sk = Mp; q.B = d.A; q.b_sk = Mp; q.b_sn = 1;
                q.scale = w.DV2; q.s_stride = R; q.scale_on_k = 1; q.M = M; q.N = M; q.K = (int)T; q.tri_out = 1;
                if ((rc = gemm(st, q, w.part, w.part_floats, w.G, nullptr, M, 1.0, 0.0, 1, R, 0, 1, (long long)M * M, &rqB)) != IWVI_OK) return rc;
            }
            if ((rc = rqB.flush(st)) != IWVI_OK) return rc;
            if (d.dq_sqrt) {
                hipLaunchKernelGGL(k_gl_tril, dim3((M + 15) / 16, (M + 15) / 16, R), dim3(256), 0, st, (const float*)w.G, d.q_sqrt, d.dq_sqrt, M, -d.kl_weight);
                if ((rc = check_launch("k_gl_tril")) != IWVI_OK) return rc;
            }
            return IWVI_OK;
        }
    }
    if (q_any && !chain) {
        // the same two gradients on the GEMM path (shapes off the streaming chain, e.g. M = 512): heads, dq_mu = A^T DMU, dL_r = tril(A^T
        // diag(2 dv_r) U_r) -- nothing of the kernel adjoint, the triangular solves or the adjoint of the factorisation
        HeadArgs h{d.A, d.U, d.noise, d.W, d.mf_A, d.d_sample, d.d_mean, d.d_var, w.DMU, w.DV2, w.SDV, nullptr, T, M, Mp, D, R, d.P, d.mf_type, d.variance, d.variance_dev,
                   d.q_mu, nullptr, d.GMV};
        hipLaunchKernelGGL(k_bw_heads, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, st, h);
        if ((rc = check_launch("k_bw_heads")) != IWVI_OK) return rc;
        if (d.dq_mu && (rc = thin(st, d.A, Mp, M, w.DMU, R, R, 0, T, w.part, w.part_floats, d.dq_mu, 0, &rqB, d.q_mu, -d.kl_weight)) != IWVI_OK) return rc;
        if (d.dq_sqrt) {
            GemmArgs q{};
            q.A = d.A; q.a_sm = 1; q.a_sk = Mp; q.B = d.U; q.b_sk = Mp; q.b_sn = 1;
            q.scale = w.DV2; q.s_stride = R; q.scale_on_k = 1; q.M = M; q.N = M; q.K = (int)T; q.tri_out = 1;
            if ((rc = gemm(st, q, w.part, w.part_floats, d.dq_sqrt, nullptr, M, 1.0, 0.0, 1, R, (long long)T * Mp, 1, (long long)M * M, &rqB, d.q_sqrt, -d.kl_weight, 1)) != IWVI_OK) return rc;
        }
        return rqB.flush(st);
    }
    MidArgs ma{d.GMV, d.noise, d.W, d.mf_A, d.d_sample, d.d_mean, d.d_var, w.DMU, w.DV2, w.SDV, d.dF, d.P, d.mf_type,
               d.U, d.A, Mp, d.q_sqrt, d.q_mu, w.LinvF, w.DK, d.F, w.Zt, w.invls, w.DA, w.Qx, (long long)T, M, D, R, d.variance, d.kern_type, d.variance_dev};
    const int fused = chain ? 1 : launch_mid(st, ma);      // heads + DA + dK + kernel adjoint in one launch where the shapes allow
    if (fused < 0) return fused;
    if (!fused) {
        HeadArgs h{d.A, d.U, d.noise, d.W, d.mf_A, d.d_sample, d.d_mean, d.d_var, w.DMU, w.DV2, w.SDV, d.dF, T, M, Mp, D, R, d.P, d.mf_type, d.variance, d.variance_dev,
                   d.q_mu, (d.dW && d.W && !d.GMV) ? w.GMV : nullptr, d.GMV};
        hipLaunchKernelGGL(k_bw_heads, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, st, h);
        if ((rc = check_launch("k_bw_heads")) != IWVI_OK) return rc;
        // DA = DMU q_mu^T - 2 SDV o A + sum_r (2 dv_r) o (U_r L_r^T): ONE launch -- R segments of M along the contraction, the
        // first two terms in the epilogue
        {
            GemmArgs q{};
            q.A = d.U; q.a_sm = Mp; q.a_sk = 1;
            q.B = d.q_sqrt; q.b_sk = 1; q.b_sn = M; q.b_keep_n_ge_k = 1;                          // B(k = j, n = i) = L_r[i][j], i >= j
            q.scale = w.DV2; q.s_stride = R; q.scale_on_k = 0;
            q.C = w.DA; q.ldc = M; q.M = (int)T; q.N = M; q.K = M; q.alpha = 1.f; q.beta = 0.f;
            q.e_dmu = w.DMU; q.e_qmu = d.q_mu; q.e_sdv = w.SDV; q.e_A = d.A; q.e_lda = Mp; q.e_R = R;
            if ((rc = gemm_rows(st, q, R, (long long)T * Mp, (long long)M * M, 1)) != IWVI_OK) return rc;
        }
        // DK = DA Lm^-1
        {
            GemmArgs q{};
            q.A = w.DA; q.a_sm = M; q.a_sk = 1; q.B = w.LinvF; q.b_sk = M; q.b_sn = 1;
            q.C = w.DK; q.ldc = M; q.M = (int)T; q.N = M; q.K = M; q.alpha = 1.f; q.beta = 0.f; q.b_lower_kn = 1;
            if ((rc = gemm_rows(st, q)) != IWVI_OK) return rc;
        }
        // C = -1/2 K o DK (over DA), dx~, dF, per-sample rows of the column sums
        KernArgs ka{d.F, w.Zt, w.invls, w.DK, w.DA, w.SDV, d.dF, w.Qx, T, M, D, d.variance, d.kern_type, d.variance_dev};
        {
            const dim3 grid((unsigned)((T + 15) / 16)), block(256);
            if (D <= 4) hipLaunchKernelGGL(k_bw_kernel<4>, grid, block, 0, st, ka);
            else if (D <= 8) hipLaunchKernelGGL(k_bw_kernel<8>, grid, block, 0, st, ka);
            else if (D <= 16) hipLaunchKernelGGL(k_bw_kernel<16>, grid, block, 0, st, ka);
            else hipLaunchKernelGGL(k_bw_kernel<32>, grid, block, 0, st, ka);
        }
        if ((rc = check_launch("k_bw_kernel")) != IWVI_OK) return rc;

    }
    // Everything the layer below needs (dF) is now queued on `st`.  What follows only produces this layer's parameter
    // gradients: with side streams it runs beside the next layer's adjoint instead of ahead of it, as two chains --
    //   A: dLm -> its reduction -> Cholesky adjoint -> K_uu's gradient          B: every other sum over samples -> one reduction
    // -- that meet in the final assembly (on B's stream).
    hipStream_t stA = st, stB = st;
    hipEvent_t evA = nullptr;
    if (d.side_stream) {
        stA = (hipStream_t)d.side_stream;
        stB = d.side_stream2 ? (hipStream_t)d.side_stream2 : stA;
        hipEvent_t ev;
        // (a stream never waits on its own event: in phase 2 the caller's stream IS the first side stream)
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, st) != hipSuccess ||
            (stA != st && hipStreamWaitEvent(stA, ev, 0) != hipSuccess) || (stB != stA && stB != st && hipStreamWaitEvent(stB, ev, 0) != hipSuccess)) {
            set_error("iwvi_gp_layer_backward: stream fork failed"); return IWVI_ERR_LAUNCH;
        }
        (void)hipEventDestroy(ev);                         // (released once the recorded work has passed it)
    }
    // ---- chain A: dLm = -tril(DK^T A)  (float64, for the adjoint of the factorisation), then S = Lm^-T Phi(Lm^T Lbar) Lm^-1
    const bool two = stB != stA;
    auto chol_adjoint = [&](hipStream_t s_) {
        const dim3 grid((M + 15) / 16, (M + 15) / 16), block(256);
        (void)grid; (void)block;
        dmm(s_, Lm64, 1, Mp, (const double*)w.Lbar, M, 1, w.T1, M, M, M, M, 1.0, nullptr, 0, 0.0, 1);      // T1 = Phi(Lm^T Lbar)
        dmm(s_, Linv64, 1, Mp, (const double*)w.T1, M, 1, w.T2, M, M, M, M);                                // T2 = Lm^-T T1
        dmm(s_, (const double*)w.T2, M, 1, Linv64, Mp, 1, w.S, M, M, M, M);                                 // S  = T2 Lm^-1
        hipLaunchKernelGGL(k_kuu_bwd, dim3((M + 3) / 4), dim3(256), 0, s_, w.Zt, (const double*)w.S, M, D, (double)d.variance, d.variance_dev, d.kern_type, w.dZt_uu, w.dvar_m);
        return check_launch("cholesky adjoint");
    };
    {
        GemmArgs q{};
        q.A = w.DK; q.a_sm = 1; q.a_sk = M; q.B = d.A; q.b_sk = Mp; q.b_sn = 1; q.M = M; q.N = M; q.K = (int)T; q.tri_out = 1;
        if (!(chain && prod) && (rc = gemm(stA, q, w.part, w.part_floats, nullptr, w.Lbar, M, -1.0, 0.0, 1, 1, 0, 0, 0, two ? &rqA : &rqB)) != IWVI_OK) return rc;
        if (two) {                                         // its own reduction and the float64 chain, concurrently with chain B
            if ((rc = rqA.flush(stA)) != IWVI_OK || (rc = chol_adjoint(stA)) != IWVI_OK) return rc;
            if (hipEventCreateWithFlags(&evA, hipEventDisableTiming) != hipSuccess || hipEventRecord(evA, stA) != hipSuccess) {
                set_error("iwvi_gp_layer_backward: stream join failed"); return IWVI_ERR_LAUNCH;
            }
        }
    }
    // ---- chain B
    st = stB;
    // dq_mu = A^T DMU
    // (- kl_weight * dKL/dq_mu = - kl_weight * q_mu rides in the reduction; temp_workaround.py:186-188)
    if (!chain && d.dq_mu && (rc = thin(st, d.A, Mp, M, w.DMU, R, R, 0, T, w.part, w.part_floats, d.dq_mu, 0, &rqB, d.q_mu, -d.kl_weight)) != IWVI_OK) return rc;
    // streaming path: dL_r = tril(G_r L_r) with G_r = A^T diag(2 dv_r) A (lower tiles of a split-K SYRK, all r in one batched launch)
    if (d.dq_sqrt && chain && !prod) {
        GemmArgs q{};
        q.A = d.A; q.a_sm = 1; q.a_sk = Mp; q.B = d.A; q.b_sk = Mp; q.b_sn = 1;
        q.scale = w.DV2; q.s_stride = R; q.scale_on_k = 1; q.M = M; q.N = M; q.K = (int)T; q.tri_out = 1;
        if ((rc = gemm(st, q, w.part, w.part_floats, w.G, nullptr, M, 1.0, 0.0, 1, R, 0, 1, (long long)M * M, &rqB)) != IWVI_OK) return rc;
    }
    // dL_r = tril(A^T diag(2 dv_r) U_r), all r in one batched launch
    if (d.dq_sqrt && !chain) {
        GemmArgs q{};
        q.A = d.A; q.a_sm = 1; q.a_sk = Mp; q.B = d.U; q.b_sk = Mp; q.b_sn = 1;
        q.scale = w.DV2; q.s_stride = R; q.scale_on_k = 1; q.M = M; q.N = M; q.K = (int)T; q.tri_out = 1;
        // (- kl_weight * dKL/dL_r = - kl_weight * (L_r - diag(1 / L_ii)) rides in the reduction)
        if ((rc = gemm(st, q, w.part, w.part_floats, d.dq_sqrt, nullptr, M, 1.0, 0.0, 1, R, (long long)T * Mp, 1, (long long)M * M, &rqB, d.q_sqrt, -d.kl_weight, 1)) != IWVI_OK) return rc;
    }
    // sums over samples: C^T [F | 1]  and the column sums of Qx = (dx~ o x | sum_r dv_r | sum_m k dk)
    if (!chain && (rc = thin(st, w.DA, M, M, d.F, D, D, 1, T, w.part, w.part_floats, w.CtF1, 0, &rqB)) != IWVI_OK) return rc;
    if (!chain && (rc = thin(st, w.Qx, D + 2, D + 2, nullptr, 0, 0, 1, T, w.part, w.part_floats, w.Qsum, 0, &rqB)) != IWVI_OK) return rc;
    // mixing matrix and linear mean function (trainable when the reference runs with fix_linear=False, build_models.py:224-227)
    if (lin_on && !chain) {
        const int P = d.P;
        const float* ups[3] = {d.d_sample, d.d_mean, d.d_var};
        if (d.dW && d.W) for (int i = 0; i < 3; ++i) if (ups[i]) {
            s[i] = w.lin + i * IWVI_MAX_P * IWVI_MAX_R;
            if ((rc = thin(st, ups[i], P, P, gmv + i * R, 3 * R, R, 0, T, w.part, w.part_floats, s[i], 0, &rqB)) != IWVI_OK) return rc;
        }
        if (d.dmf_A && d.mf_type == IWVI_MF_LINEAR) for (int i = 0; i < 2; ++i) if (ups[i]) {
            a12[i] = w.lin + 3 * IWVI_MAX_P * IWVI_MAX_R + i * IWVI_MAX_D * IWVI_MAX_P;
            if ((rc = thin(st, d.F, D, D, ups[i], P, P, 0, T, w.part, w.part_floats, a12[i], 0, &rqB)) != IWVI_OK) return rc;
        }
    }
    if ((rc = rqB.flush(st)) != IWVI_OK) return rc;
    if (d.dq_sqrt && chain) {                              // (- kl_weight * dKL/dL_r = - kl_weight * (L_r - diag(1 / L_ii)) rides along)
        hipLaunchKernelGGL(k_gl_tril, dim3((M + 15) / 16, (M + 15) / 16, R), dim3(256), 0, st, (const float*)w.G, d.q_sqrt, d.dq_sqrt, M, -d.kl_weight);
        if ((rc = check_launch("k_gl_tril")) != IWVI_OK) return rc;
    }
    if (!two && (rc = chol_adjoint(st)) != IWVI_OK) return rc;     // one stream: after the single reduction, as before
    const int n_w = (lin_on && d.dW && d.W) ? d.P * R : 0, n_a = (lin_on && d.dmf_A && d.mf_type == IWVI_MF_LINEAR) ? D * d.P : 0;
    if (evA) {                                             // the assembly needs chain A's dZt_uu / dvar_m
        if (hipStreamWaitEvent(stB, evA, 0) != hipSuccess) { set_error("iwvi_gp_layer_backward: stream join failed"); return IWVI_ERR_LAUNCH; }
        (void)hipEventDestroy(evA);
    }
    FinalArgsB f{d.Z, d.lengthscales, d.q_mu, d.q_sqrt, w.Zt, w.invls, w.CtF1, w.CtF1, w.Qsum + D, w.Qsum, w.dZt_uu, w.dvar_m,
                 d.dZ, d.dls, d.dvariance, d.dq_mu, d.dq_sqrt, M, D, R, d.kl_weight, (double)d.variance, d.variance_dev,
                 s[0], s[1], s[2], d.W, n_w ? d.dW : nullptr, n_w, a12[0], a12[1], n_a ? d.dmf_A : nullptr, n_a};
    hipLaunchKernelGGL(k_bw_final, dim3(D + 1 + (lin_on ? 1 : 0)), dim3(64), 0, st, f);
    return check_launch("k_bw_final");
}

extern "C" int iwvi_iw_elbo_backward(const float* fmean, const float* fvar, const float* Y, int Dy,
                                     const float* const* kl_local, const int32_t* kl_dims, int n_local,
                                     int64_t B, int K, float lik_variance, double scale, int mode_vi,
                                     float* out_w, float* d_mean, float* d_var,
                                     const double* const* kl_global, const int32_t* kl_global_counts, int n_glob,
                                     const float* lse_global, int K_total, double* out_sums, double* ws, void* stream_) {
    return iwvi_iw_elbo_backward_dev(fmean, fvar, Y, Dy, kl_local, kl_dims, n_local, B, K, lik_variance, nullptr, scale, mode_vi, out_w, d_mean, d_var,
                                     kl_global, kl_global_counts, n_glob, lse_global, K_total, out_sums, ws, stream_);
}

extern "C" int iwvi_iw_elbo_backward_dev(const float* fmean, const float* fvar, const float* Y, int Dy,
                                         const float* const* kl_local, const int32_t* kl_dims, int n_local,
                                         int64_t B, int K, float lik_variance, const float* lik_variance_dev, double scale, int mode_vi,
                                         float* out_w, float* d_mean, float* d_var,
                                         const double* const* kl_global, const int32_t* kl_global_counts, int n_glob,
                                         const float* lse_global, int K_total,
                                         double* out_sums /* [3]: sum_n (lse - log K), d/d lik_variance, the bound */, double* ws, void* stream_) {
    if (!fmean || !fvar || !Y || !out_sums || !ws || Dy <= 0 || B <= 0 || K <= 0 || n_local < 0 || n_local > IWVI_MAX_KL || !(lik_variance > 0.f) ||
        n_glob < 0 || n_glob > IWVI_MAX_LAYERS) {
        set_error("iwvi_iw_elbo_backward: bad argument"); return IWVI_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream_;
    ElboBwdArgs a{};
    a.fmean = fmean; a.fvar = fvar; a.Y = Y; a.Dy = Dy; a.n_kl = n_local;
    for (int i = 0; i < n_local; ++i) {
        if (!kl_local || !kl_local[i] || !kl_dims || kl_dims[i] <= 0) { set_error("iwvi_iw_elbo_backward: bad local regulariser %d", i); return IWVI_ERR_ARG; }
        a.kl[i] = kl_local[i]; a.kl_dims[i] = kl_dims[i];
    }
    if (lse_global && (mode_vi || K_total < K)) { set_error("iwvi_iw_elbo_backward: lse_global needs the IW bound and K_total >= K"); return IWVI_ERR_ARG; }
    a.B = B; a.K = K; a.lik_var = lik_variance; a.lik_var_dev = lik_variance_dev; a.scale = scale; a.mode_vi = mode_vi; a.lse_global = lse_global; a.K_total = K_total; a.w = out_w; a.d_mean = d_mean; a.d_var = d_var; a.part = ws;
    hipLaunchKernelGGL(k_elbo_bwd, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, a);
    ElboFinishArgs fa{};
    fa.part = ws; fa.n = B; fa.scale = scale; fa.n_glob = n_glob; fa.out = out_sums;
    for (int i = 0; i < n_glob; ++i) {
        if (!kl_global || !kl_global[i] || !kl_global_counts || kl_global_counts[i] <= 0) { set_error("iwvi_iw_elbo_backward: bad global KL %d", i); return IWVI_ERR_ARG; }
        fa.klg[i] = kl_global[i]; fa.kln[i] = kl_global_counts[i];
    }
    hipLaunchKernelGGL(k_elbo_finish, dim3(1), dim3(256), 0, st, fa);
    return check_launch("k_elbo_bwd");
}

extern "C" int iwvi_lv_layer_backward(const float* mu, const float* sigma, int ld_enc, int sigma_is_raw, const float* noise,
                                      const float* dF_next, int ld_next, int col0, const float* w,
                                      int latent_dim, int64_t B, int K, int sampled_kl, float* d_enc_out, void* stream_) {
    if (!mu || !sigma || !noise || !d_enc_out || latent_dim <= 0 || ld_enc < latent_dim || B <= 0 || K <= 0 || (dF_next && (ld_next < col0 + latent_dim || col0 < 0))) {
        set_error("iwvi_lv_layer_backward: bad argument"); return IWVI_ERR_ARG;
    }
    LvBwdArgs a{mu, sigma, ld_enc, sigma_is_raw, noise, dF_next, ld_next, col0, w, latent_dim, B, K, sampled_kl, d_enc_out};
    hipLaunchKernelGGL(k_lv_bwd, dim3((unsigned)((B * latent_dim + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, a);
    return check_launch("k_lv_bwd");
}

static int enc_param_total(const int32_t* dims, int n) {
    int p = 0;
    for (int l = 0; l < n; ++l) p += dims[l] * dims[l + 1] + dims[l + 1];
    return p;
}
extern "C" size_t iwvi_encoder_backward_ws_bytes(int64_t rows, const int32_t* dims, int n_enc) {
    if (rows <= 0 || !dims || n_enc <= 0 || n_enc > IWVI_MAX_ENC) return 0;
    for (int l = 0; l <= n_enc; ++l) if (dims[l] <= 0 || dims[l] > 64) return 0;
    return align256(sizeof(float) * (size_t)((rows + ER - 1) / ER) * enc_param_total(dims, n_enc)) + 256;
}
extern "C" int iwvi_encoder_backward(const float* XY, int64_t rows, const float* const* enc_W, const float* const* enc_b,
                                     const int32_t* dims, int n_enc, const float* d_out,
                                     float* const* dW, float* const* db, void* ws_, void* stream_) {
    return iwvi_encoder_backward_act(XY, rows, enc_W, enc_b, dims, n_enc, IWVI_ACT_TANH, d_out, dW, db, ws_, stream_);
}

static int enc_bwd_impl(const float* XY, int64_t rows, const float* const* enc_W, const float* const* enc_b,
                        const int32_t* dims, int n_enc, int act, const float* d_out, const LvBwdArgs* lv,
                        float* const* dW, float* const* db, void* ws_, void* stream_);
extern "C" int iwvi_encoder_backward_act(const float* XY, int64_t rows, const float* const* enc_W, const float* const* enc_b,
                                         const int32_t* dims, int n_enc, int act, const float* d_out,
                                         float* const* dW, float* const* db, void* ws_, void* stream_) {
    if (!d_out) { set_error("iwvi_encoder_backward: bad argument"); return IWVI_ERR_ARG; }
    return enc_bwd_impl(XY, rows, enc_W, enc_b, dims, n_enc, act, d_out, nullptr, dW, db, ws_, stream_);
}
// iwvi_lv_layer_backward + iwvi_encoder_backward_act as one launch (+ the reduction): the workgroup that back-propagates eight data rows
// through the encoder forms their d(means | raw) itself
extern "C" int iwvi_lv_encoder_backward(const float* mu, const float* sigma, int ld_enc, int sigma_is_raw, const float* noise,
                                        const float* dF_next, int ld_next, int col0, const float* w,
                                        int latent_dim, int64_t B, int K, int sampled_kl,
                                        const float* XY, const float* const* enc_W, const float* const* enc_b,
                                        const int32_t* dims, int n_enc, int act,
                                        float* const* dW, float* const* db, void* ws_, void* stream_) {
    if (!mu || !sigma || !noise || latent_dim <= 0 || ld_enc < latent_dim || B <= 0 || K <= 0 || (dF_next && (ld_next < col0 + latent_dim || col0 < 0))) {
        set_error("iwvi_lv_encoder_backward: bad argument"); return IWVI_ERR_ARG;
    }
    if (!dims || n_enc <= 0 || n_enc > IWVI_MAX_ENC || dims[n_enc] != 2 * latent_dim) { set_error("iwvi_lv_encoder_backward: the encoder's output width must be 2 * latent_dim"); return IWVI_ERR_ARG; }
    const LvBwdArgs lv{mu, sigma, ld_enc, sigma_is_raw, noise, dF_next, ld_next, col0, w, latent_dim, B, K, sampled_kl, nullptr};
    return enc_bwd_impl(XY, B, enc_W, enc_b, dims, n_enc, act, nullptr, &lv, dW, db, ws_, stream_);
}
static int enc_bwd_impl(const float* XY, int64_t rows, const float* const* enc_W, const float* const* enc_b,
                        const int32_t* dims, int n_enc, int act, const float* d_out, const LvBwdArgs* lv,
                        float* const* dW, float* const* db, void* ws_, void* stream_) {
    if (act < IWVI_ACT_TANH || act > IWVI_ACT_IDENTITY) { set_error("iwvi_encoder_backward: unknown activation %d", act); return IWVI_ERR_UNSUPPORTED; }
    if (!XY || !enc_W || !dims || (!d_out && !lv) || !dW || !ws_ || rows <= 0 || n_enc <= 0 || n_enc > IWVI_MAX_ENC) { set_error("iwvi_encoder_backward: bad argument"); return IWVI_ERR_ARG; }
    for (int l = 0; l <= n_enc; ++l) if (dims[l] <= 0 || dims[l] > 64) { set_error("iwvi_encoder_backward: encoder width %d out of range (1..64)", dims[l]); return IWVI_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream_;
    EncBwdArgs a{};
    EncReduceArgs r{};
    a.XY = XY; a.rows = rows; a.n = n_enc; a.d_out = d_out; a.act = act; a.part = (float*)ws_;
    if (lv) { a.lv = *lv; a.fused = 1; }
    int off = 0;
    for (int l = 0; l <= n_enc; ++l) a.dims[l] = dims[l];
    for (int l = 0; l < n_enc; ++l) {
        if (!enc_W[l] || !dW[l]) { set_error("iwvi_encoder_backward: null weight %d", l); return IWVI_ERR_ARG; }
        a.W[l] = enc_W[l]; a.b[l] = enc_b ? enc_b[l] : nullptr;
        a.woff[l] = off; r.off[2 * l] = off; r.len[2 * l] = dims[l] * dims[l + 1]; r.dst[2 * l] = dW[l]; off += dims[l] * dims[l + 1];
        a.boff[l] = off; r.off[2 * l + 1] = off; r.len[2 * l + 1] = dims[l + 1]; r.dst[2 * l + 1] = db ? db[l] : nullptr; off += dims[l + 1];
    }
    a.ptot = off;
    const int nblk = (int)((rows + ER - 1) / ER);
    r.part = a.part; r.nblk = nblk; r.ptot = off; r.n = 2 * n_enc;
    size_t elds = sizeof(float) * (size_t)(n_enc + 5) * ER * ELD;
    for (int l = 0; l < n_enc; ++l) elds += sizeof(float) * (size_t)(dims[l] * dims[l + 1] + dims[l + 1]);
    int rc;
    {   // once per process: allow the largest encoder (hipFuncSetAttribute is not a stream operation)
        static bool done = false;
        if (!done) {
            const size_t most = sizeof(float) * ((size_t)(IWVI_MAX_ENC + 5) * ER * ELD + EW_MAX);
            hipError_t e = hipFuncSetAttribute((const void*)k_enc_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)most);
            if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", most, hipGetErrorString(e)); return IWVI_ERR_LAUNCH; }
            done = true;
        }
    }
    hipLaunchKernelGGL(k_enc_bwd, dim3((unsigned)nblk), dim3(256), elds, st, a);
    if ((rc = check_launch("k_enc_bwd")) != IWVI_OK) return rc;
    hipLaunchKernelGGL(k_enc_reduce, dim3((off + 63) / 64), dim3(256), 0, st, r);
    return check_launch("k_enc_reduce");
}

extern "C" size_t iwvi_natgrad_ws_bytes(int M) {
    if (M <= 0 || M > IWVI_MAX_M) return 0;
    return align256(sizeof(double) * (size_t)M * M) * 8 + align256(sizeof(double) * M) * 4 + align256(iwvi_chol_ws_bytes(M)) + 256;
}

// GPflow 1.x NatGradOptimizer (natural parameterisation, XiNat) on a whitened (q_mu [M, R], q_sqrt [R, M, M]):
//   eta = (m, S + m m^T), theta = (S^-1 m, -1/2 S^-1);  theta <- theta - gamma dLoss/d eta;  back through
//   natural_to_meanvarsqrt (cholesky(-2 theta_2), its inverse, S = X^T X, mu = S theta_1, cholesky(S)).
// dq_mu / dq_sqrt are gradients of the objective that is MAXIMISED (the ELBO): loss = -ELBO.
//
// The same update without S^-1, with one factorisation instead of two.  With S = L L^T, Lbar = dLoss/dL, mbar = dLoss/dm:
//   dLoss/dS = L^-T sym(Phi) L^-1,  Phi = Phi(L^T Lbar)  (lower triangle, diagonal halved; sym(A) = (A + A^T) / 2)
//   -2 theta_2' = S^-1 + 2 gamma dLoss/dS = L^-T Q L^-1,          Q = I + gamma (Phi + Phi^T)
//   S' = (-2 theta_2')^-1 = L Q^-1 L^T = (L W)(L W)^T             with Q^-1 = W W^T, W lower  =>  L' = chol(S') = L W
//   theta_1' = L^-T (Q y - gamma L^T mbar), y = L^-1 m            =>  mu' = S' theta_1' = m - gamma L' (L'^T mbar)
// Q^-1 = W W^T with W lower is Q = V V^T with V = W^-T UPPER: the Cholesky factorisation of Q with rows and columns reversed
// (J Q J = C C^T, C lower, V = J C J), so W[k][j] = (C^-1)[M-1-j][M-1-k].  Every step is the reference's arithmetic regrouped
// (float64 throughout); an indefinite -2 theta_2' shows up as an indefinite Q (congruent) and ends in NaN as before.
__global__ void k_ng_q(const double* __restrict__ Phi, double gamma, double* __restrict__ Qrev, int M) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * M) return;
    const int i = idx / M, j = idx - i * M, ri = M - 1 - i, rj = M - 1 - j;
    Qrev[idx] = (i == j ? 1.0 : 0.0) + gamma * (Phi[(size_t)ri * M + rj] + Phi[(size_t)rj * M + ri]);
}
static int g_ng_route = 0;                                // diagnostic; see iwvi_debug_last_natgrad_route
extern "C" int iwvi_debug_last_natgrad_route(void) { return g_ng_route; }
extern "C" int iwvi_natgrad_step(float* q_mu, float* q_sqrt, const float* dq_mu, const float* dq_sqrt,
                                 int M, int R, double gamma, void* ws_, void* stream_) {
    return iwvi_natgrad_step_ex(q_mu, q_sqrt, dq_mu, dq_sqrt, M, R, gamma, ws_, iwvi_natgrad_ws_bytes(M), stream_);
}
extern "C" int iwvi_natgrad_step_ex(float* q_mu, float* q_sqrt, const float* dq_mu, const float* dq_sqrt,
                                    int M, int R, double gamma, void* ws_, size_t ws_bytes, void* stream_) {
    if (!q_mu || !q_sqrt || !dq_mu || !dq_sqrt || !ws_ || M <= 0 || M > IWVI_MAX_M || R <= 0 || R > IWVI_MAX_R) { set_error("iwvi_natgrad_step: bad argument"); return IWVI_ERR_ARG; }
    if (ws_bytes < iwvi_natgrad_ws_bytes(M)) { set_error("iwvi_natgrad_step: workspace of %zu B, iwvi_natgrad_ws_bytes(%d) = %zu", ws_bytes, M, iwvi_natgrad_ws_bytes(M)); return IWVI_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream_;
    {   // M <= 128: the whole step in one workgroup per latent GP, or spread over the chip (csrc/precompute.hip: natgrad_small)
        const int rcs = natgrad_small(q_mu, q_sqrt, dq_mu, dq_sqrt, M, R, gamma, st, ws_, ws_bytes);
        if (rcs != 0) { if (rcs > 0) g_ng_route = rcs; return rcs > 0 ? IWVI_OK : rcs; }
    }
    g_ng_route = 0;
    char* base = (char*)ws_; size_t o = 0;
    auto mat = [&]() { double* p = (double*)(base + o); o += align256(sizeof(double) * (size_t)M * M); return p; };
    auto vec = [&]() { double* p = (double*)(base + o); o += align256(sizeof(double) * M); return p; };
    double *L = mat(), *Lbar = mat(), *Phi = mat(), *Qrev = mat(), *C = mat(), *Cinv = mat(), *Lnew = mat(), *unused = mat();
    double *m = vec(), *mbar = vec(), *t = vec(), *mnew = vec();
    (void)unused;
    void* cws = base + o;
    const int nb = (M * M + 255) / 256;
    int rc;
    for (int r = 0; r < R; ++r) {
        hipLaunchKernelGGL(k_f2d, dim3(nb), dim3(256), 0, st, (const float*)(q_sqrt + (size_t)r * M * M), (long long)M, L, M, M, 1.0, 1);
        hipLaunchKernelGGL(k_f2d, dim3(nb), dim3(256), 0, st, dq_sqrt + (size_t)r * M * M, (long long)M, Lbar, M, M, -1.0, 1);
        hipLaunchKernelGGL(k_f2d, dim3((M + 255) / 256), dim3(256), 0, st, (const float*)(q_mu + r), (long long)R, m, M, 1, 1.0, 0);
        hipLaunchKernelGGL(k_f2d, dim3((M + 255) / 256), dim3(256), 0, st, dq_mu + r, (long long)R, mbar, M, 1, -1.0, 0);
        dmm(st, L, 1, M, Lbar, M, 1, Phi, M, M, M, M, 1.0, nullptr, 0, 0.0, 1);          // Phi(L^T Lbar)
        hipLaunchKernelGGL(k_ng_q, dim3(nb), dim3(256), 0, st, (const double*)Phi, gamma, Qrev, M);
        if ((rc = iwvi_chol_factor(Qrev, C, M, cws, stream_)) != IWVI_OK) return rc;
        if ((rc = tri_inverse(st, C, Cinv, M)) != IWVI_OK) return rc;
        // L' = L W,  W[k][j] = Cinv[M-1-j][M-1-k]
        dmm(st, L, M, 1, Cinv + (size_t)(M - 1) * M + (M - 1), -1, -(long long)M, Lnew, M, M, M, M);
        // mu' = m - gamma L' (L'^T mbar)
        dmm(st, Lnew, 1, M, mbar, 1, 1, t, 1, M, 1, M);
        dmm(st, Lnew, M, 1, t, 1, 1, mnew, 1, M, 1, M, -gamma, m, 1, 1.0);
        hipLaunchKernelGGL(k_d2f, dim3(nb), dim3(256), 0, st, (const double*)Lnew, q_sqrt + (size_t)r * M * M, (long long)M, M, M, 1);
        hipLaunchKernelGGL(k_d2f, dim3((M + 255) / 256), dim3(256), 0, st, (const double*)mnew, q_mu + r, (long long)R, M, 1, 0);
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
        if (!s.param || !s.x || !s.m || !s.v || (!init && !s.grad) || s.n <= 0 || ((s.transform & ~IWVI_ADAM_GRAD_F64) != 0 && (s.transform & ~IWVI_ADAM_GRAD_F64) != 1)) { set_error("iwvi_adam_step: bad tensor %d", i); return IWVI_ERR_ARG; }
        a.t[i] = AdamTensor{s.param, s.grad, s.x, s.m, s.v, (long long)s.n, s.transform & ~IWVI_ADAM_GRAD_F64, (s.transform & IWVI_ADAM_GRAD_F64) ? 1 : 0};
        if (s.n > nmax) nmax = s.n;
    }
    a.n = n_tensors; a.init = init;
    a.lr_t = init ? 0.f : (float)(lr * sqrt(1.0 - pow(beta2, (double)t)) / (1.0 - pow(beta1, (double)t)));
    a.b1 = (float)beta1; a.b2 = (float)beta2; a.eps = (float)eps; a.sign = maximise ? -1.f : 1.f;
    long long blocks = (nmax + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks, n_tensors), dim3(256), 0, (hipStream_t)stream_, a);
    return check_launch("k_adam");
}

extern "C" int iwvi_adam_step_dev(const iwvi_adam_tensor* tensors, int n_tensors, double lr, double beta1, double beta2,
                                  double eps, int64_t* t_dev, int maximise, void* stream_) {
    if (!tensors || n_tensors <= 0 || n_tensors > ADAM_MAX || !t_dev) { set_error("iwvi_adam_step_dev: bad argument (at most %d tensors)", ADAM_MAX); return IWVI_ERR_ARG; }
    AdamArgs a{};
    long long nmax = 1;
    for (int i = 0; i < n_tensors; ++i) {
        const iwvi_adam_tensor& s = tensors[i];
        if (!s.param || !s.x || !s.m || !s.v || !s.grad || s.n <= 0 || ((s.transform & ~IWVI_ADAM_GRAD_F64) != 0 && (s.transform & ~IWVI_ADAM_GRAD_F64) != 1)) { set_error("iwvi_adam_step_dev: bad tensor %d", i); return IWVI_ERR_ARG; }
        a.t[i] = AdamTensor{s.param, s.grad, s.x, s.m, s.v, (long long)s.n, s.transform & ~IWVI_ADAM_GRAD_F64, (s.transform & IWVI_ADAM_GRAD_F64) ? 1 : 0};
        if (s.n > nmax) nmax = s.n;
    }
    a.n = n_tensors; a.init = 0; a.lr = (float)lr; a.t_dev = (const long long*)t_dev;
    a.b1 = (float)beta1; a.b2 = (float)beta2; a.eps = (float)eps; a.sign = maximise ? -1.f : 1.f;
    long long blocks = (nmax + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks, n_tensors), dim3(256), 0, (hipStream_t)stream_, a);
    hipLaunchKernelGGL(k_inc_i64, dim3(1), dim3(64), 0, (hipStream_t)stream_, (long long*)t_dev);
    return check_launch("k_adam (device step count)");
}

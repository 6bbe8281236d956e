// Per-step inducing-set factorisation for every GP layer of the model, in float64:
//   Kuu = K(Z,Z) + jitter I  ->  Lm = chol(Kuu)  ->  Lm^-1  ->  MFMA-fragment packed float32 operands
// Replaces (reference file:line) Kuu + tf.cholesky (temp_workaround.py:39,48), the operand side of
// tf.matrix_triangular_solve (:51), tf.matrix_band_part(q_sqrt) (:78) and gauss_kl (:186-188).
//
// Two launches for ALL layers of a model (grid.x = layer):
//   k_kuu_chol : one 1024-thread workgroup per layer; Gram + blocked right-looking Cholesky (NB=16),
//                matrix resident in LDS when Mp <= 128, in L2-resident global memory otherwise;
//                also inverts every 16x16 diagonal block (needed by the panel solve anyway).
//   k_inv_pack : (layer, role) workgroups: 32-wide column blocks of Lm^-1 by blocked forward
//                substitution + packing, one role per latent GP for tril(q_sqrt)^T, one for q_mu^T + KL.
#include "iwvi_common.h"

namespace iwvi {

constexpr int NB = 16;          // Cholesky / inverse block size
constexpr int PLD = NB + 1;     // padded leading dimension of the LDS panels (doubles)

struct PreLayer {
    const float* Z; const float* ls; const float* q_mu; const float* q_sqrt;
    double* Lm; double* Linv; float* LinvP; float* LrTP; float* QmuP; float* Zs; float* invls; double* kl;
    double jitter; float variance;
    int M, D, R, Mp, nb, kern_type;
};
struct PreArgs { PreLayer L[IWVI_MAX_LAYERS]; int n; };

__device__ __forceinline__ double kern_value(double r2, int type, double var) {
    if (type == IWVI_KERN_MATERN52) {
        const double s5 = 2.23606797749978969641;
        double r = sqrt(r2 + 1e-12);
        return var * (1.0 + s5 * r + (5.0 / 3.0) * r * r) * exp(-s5 * r);
    }
    return var * exp(-0.5 * r2);
}

__device__ __forceinline__ void wave_lds_sync() {
    // single-wave producer/consumer through LDS: LDS ops of one wave retire in order; this only
    // has to stop the compiler from moving the reads above the writes.
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// In-LDS factorisation of a 16x16 SPD block by ONE wave (lane & 15 = row), then its inverse.
// Dg: in = SPD block (lower used), out = lower Cholesky factor.  Di: out = inverse of that factor.
__device__ void factor_invert_16(double* Dg, double* Di, int lane) {
    const int i = lane & 15;
    const bool act = lane < 16;
    for (int j = 0; j < NB; ++j) {
        double s = 0.0;
        if (act && i >= j) {
            s = Dg[i * PLD + j];
            for (int k = 0; k < j; ++k) s -= Dg[i * PLD + k] * Dg[j * PLD + k];
            Dg[i * PLD + j] = s;
        }
        wave_lds_sync();
        if (act && i >= j) {
            double djj = sqrt(Dg[j * PLD + j]);
            // every lane i > j scales its own entry; lane j stores the pivot last (after the sync)
            if (i > j) Dg[i * PLD + j] = s / djj;
        }
        wave_lds_sync();
        if (act && i == j) Dg[j * PLD + j] = sqrt(s);
        wave_lds_sync();
    }
    // inverse: lane c owns column c of X = L^-1 (forward substitution on e_c)
    if (act) {
        const int c = i;
        double x[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < r; ++k) s -= Dg[r * PLD + k] * x[k];
            x[r] = (r >= c) ? s / Dg[r * PLD + r] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < NB; ++r) Di[r * PLD + c] = x[r];
    }
    wave_lds_sync();
}

// Blocked right-looking Cholesky of A [n x n], leading dimension ld, lower triangle, in place.
// A may live in LDS or in global memory (generic pointer). Linv receives the inverses of the 16x16
// diagonal blocks (the rest of Linv is filled by k_inv_pack). smem: Dg, Di, P scratch.
__device__ void chol_blocked(double* A, int n, int ld, double* Linv, int ldi,
                             double* Dg, double* Di, double* P, int tid, int nthreads) {
    const int lane = tid & 63, wave = tid >> 6;
    for (int c0 = 0; c0 < n; c0 += NB) {
        if (tid < NB * NB) {
            int r = tid / NB, c = tid % NB;
            Dg[r * PLD + c] = (c <= r) ? A[(size_t)(c0 + r) * ld + c0 + c] : 0.0;
        }
        __syncthreads();
        if (wave == 0) factor_invert_16(Dg, Di, lane);
        __syncthreads();
        if (tid < NB * NB) {
            int r = tid / NB, c = tid % NB;
            if (c <= r) {
                A[(size_t)(c0 + r) * ld + c0 + c] = Dg[r * PLD + c];
                Linv[(size_t)(c0 + r) * ldi + c0 + c] = Di[r * PLD + c];
            }
        }
        const int r0 = c0 + NB, nrem = n - r0;
        // panel: L[i][c0+c] = sum_{k<=c} A[i][c0+k] * Dinv[c][k]
        for (int idx = tid; idx < nrem * NB; idx += nthreads) {
            int ii = idx / NB, c = idx % NB;
            const double* arow = A + (size_t)(r0 + ii) * ld + c0;
            double s = 0.0;
            for (int k = 0; k <= c; ++k) s += arow[k] * Di[c * PLD + k];
            P[ii * PLD + c] = s;
        }
        __syncthreads();
        for (int idx = tid; idx < nrem * NB; idx += nthreads) {
            int ii = idx / NB, c = idx % NB;
            A[(size_t)(r0 + ii) * ld + c0 + c] = P[ii * PLD + c];
        }
        // trailing update (lower triangle incl. diagonal): A[i][k] -= sum_c P[i][c] P[k][c]
        for (int idx = tid; idx < nrem * nrem; idx += nthreads) {
            int ii = idx / nrem, kk = idx % nrem;
            if (kk > ii) continue;
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < NB; ++c) s += P[ii * PLD + c] * P[kk * PLD + c];
            A[(size_t)(r0 + ii) * ld + r0 + kk] -= s;
        }
        __syncthreads();
    }
}

extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

// Gram (float64, from the float32-rounded scaled inducing inputs that Kuf also uses) + Cholesky.
__global__ __launch_bounds__(1024) void k_kuu_chol(PreArgs args) {
    const PreLayer& L = args.L[blockIdx.x];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int M = L.M, D = L.D, Mp = L.Mp;
    double* sm = reinterpret_cast<double*>(smem_raw);
    double* Dg = sm;
    double* Di = Dg + NB * PLD;
    double* P = Di + NB * PLD;
    double* Alds = P + (size_t)(Mp - NB > 0 ? Mp - NB : 1) * PLD;
    const bool in_lds = Mp <= 128;
    double* A = in_lds ? Alds : L.Lm;
    const int ld = Mp;

    for (int idx = tid; idx < Mp * 32; idx += nthreads) {
        int m = idx >> 5, d = idx & 31;
        float v = 0.f;
        if (m < M && d < D) v = (float)((double)L.Z[(size_t)m * D + d] / (double)L.ls[d]);
        L.Zs[idx] = v;
    }
    if (tid < 32) L.invls[tid] = (tid < D) ? (float)(1.0 / (double)L.ls[tid]) : 0.f;
    __syncthreads();   // Zs is read back below by other threads of this workgroup
    for (int idx = tid; idx < Mp * Mp; idx += nthreads) {
        int i = idx / Mp, j = idx % Mp;
        if (j > i) { if (!in_lds) A[idx] = 0.0; else A[idx] = 0.0; continue; }
        double v;
        if (i >= M) v = (i == j) ? 1.0 : 0.0;           // identity padding
        else {
            double r2 = 0.0;
            for (int d = 0; d < D; ++d) {
                double df = (double)L.Zs[i * 32 + d] - (double)L.Zs[j * 32 + d];
                r2 += df * df;
            }
            v = kern_value(r2, L.kern_type, (double)L.variance);
            if (i == j) v += L.jitter;
        }
        A[idx] = v;
    }
    __syncthreads();
    chol_blocked(A, Mp, ld, L.Linv, Mp, Dg, Di, P, tid, nthreads);
    if (in_lds) {
        for (int idx = tid; idx < Mp * Mp; idx += nthreads) L.Lm[idx] = A[idx];
    }
}

// standalone Gram / Cholesky entry points (K1, K2) reuse the same device code
__global__ void k_gram_sym(const float* Z, const float* ls, float variance, double jitter, int type,
                           int M, int D, double* K) {
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < (size_t)M * M;
         idx += (size_t)gridDim.x * blockDim.x) {
        int i = idx / M, j = idx % M;
        double r2 = 0.0;
        for (int d = 0; d < D; ++d) {
            double a = (double)(float)((double)Z[(size_t)i * D + d] / (double)ls[d]);
            double b = (double)(float)((double)Z[(size_t)j * D + d] / (double)ls[d]);
            r2 += (a - b) * (a - b);
        }
        double v = kern_value(r2, type, (double)variance);
        if (i == j) v += jitter;
        K[idx] = v;
    }
}

__global__ __launch_bounds__(1024) void k_chol_only(double* A, int M, double* scratch_inv) {
    // A [M x M] row-major, M multiple of 16 handled by caller via padded copy in `scratch`
    double* sm = reinterpret_cast<double*>(smem_raw);
    double* Dg = sm;
    double* Di = Dg + NB * PLD;
    double* P = Di + NB * PLD;
    chol_blocked(A, M, M, scratch_inv, M, Dg, Di, P, threadIdx.x, blockDim.x);
}

__global__ void k_pad_copy(const double* src, int M, double* dst, int Mp) {
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < (size_t)Mp * Mp;
         idx += (size_t)gridDim.x * blockDim.x) {
        int i = idx / Mp, j = idx % Mp;
        dst[idx] = (i < M && j < M) ? src[(size_t)i * M + j] : ((i == j) ? 1.0 : 0.0);
    }
}
__global__ void k_unpad_copy(const double* src, int Mp, double* dst, int M) {
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < (size_t)M * M;
         idx += (size_t)gridDim.x * blockDim.x) {
        int i = idx / M, j = idx % M;
        dst[idx] = (j <= i) ? src[(size_t)i * Mp + j] : 0.0;
    }
}

// ---------------------------------------------------------------------------------------------
// k_inv_pack roles (blockIdx.y): [0, nb) inverse column block; [nb, nb+R) LrT pack; nb+R: q_mu + KL
// ---------------------------------------------------------------------------------------------
constexpr int XLD = 33;   // leading dimension of the LDS column buffer (doubles)

__device__ void role_inverse(const PreLayer& L, int jb, double* X, double* S) {
    const int tid = threadIdx.x;
    const int Mp = L.Mp, M = L.M;
    const int cbase = 32 * jb;
    const int r = tid >> 4, c2 = tid & 15;            // 16 rows x (2 x 16 columns)
    // rows above the column block are zero
    for (int idx = tid; idx < Mp * 32; idx += blockDim.x) X[(idx >> 5) * XLD + (idx & 31)] = 0.0;
    __syncthreads();
    for (int ib = 2 * jb; ib < Mp / NB; ++ib) {
        const int row = NB * ib + r;
        double s0 = (row == cbase + c2) ? 1.0 : 0.0;
        double s1 = (row == cbase + c2 + 16) ? 1.0 : 0.0;
        const double* lrow = L.Lm + (size_t)row * Mp;
        for (int kk = cbase; kk < NB * ib; ++kk) {
            double l = lrow[kk];
            s0 -= l * X[kk * XLD + c2];
            s1 -= l * X[kk * XLD + c2 + 16];
        }
        S[r * XLD + c2] = s0;
        S[r * XLD + c2 + 16] = s1;
        __syncthreads();
        // X_ib = Dinv_ib * S   (Dinv_ib = lower 16x16 block of Linv on the diagonal)
        const double* dinv = L.Linv + (size_t)(NB * ib + r) * Mp + NB * ib;
        double x0 = 0.0, x1 = 0.0;
        for (int t = 0; t <= r; ++t) {
            double dv = dinv[t];
            x0 += dv * S[t * XLD + c2];
            x1 += dv * S[t * XLD + c2 + 16];
        }
        X[row * XLD + c2] = x0;
        X[row * XLD + c2 + 16] = x1;
        __syncthreads();
    }
    // write the float64 inverse (lower triangle of this column block; the 16x16 diagonal blocks were
    // already written by k_kuu_chol and are rewritten with identical values)
    for (int idx = tid; idx < Mp * 32; idx += blockDim.x) {
        int i = idx >> 5, c = idx & 31;
        L.Linv[(size_t)i * Mp + cbase + c] = (cbase + c <= i) ? X[i * XLD + c] : 0.0;
    }
    // pack blocks (bi >= jb, bk = jb) as float32, masking the identity padding to zero
    const int nb = L.nb;
    for (int bi = jb; bi < nb; ++bi) {
        float* dst = L.LinvP + (size_t)(bi * nb + jb) * 1024;
        for (int idx = tid; idx < 1024; idx += blockDim.x) {
            int q = idx >> 8, lane = (idx >> 2) & 63, e = idx & 3;
            int i = 32 * bi + (lane & 31);
            int kc = 8 * q + 4 * (lane >> 5) + e;
            float v = 0.f;
            if (i < M && cbase + kc < M && cbase + kc <= i) v = (float)X[i * XLD + kc];
            dst[idx] = v;
        }
    }
}

__device__ void role_pack_LrT(const PreLayer& L, int r) {
    const int nb = L.nb, M = L.M;
    const float* q = L.q_sqrt + (size_t)r * M * M;
    float* base = L.LrTP + (size_t)r * nb * nb * 1024;
    for (int bi = 0; bi < nb; ++bi)
        for (int bk = bi; bk < nb; ++bk) {
            float* dst = base + (size_t)(bi * nb + bk) * 1024;
            for (int idx = threadIdx.x; idx < 1024; idx += blockDim.x) {
                int qq = idx >> 8, lane = (idx >> 2) & 63, e = idx & 3;
                int i = 32 * bi + (lane & 31);
                int k = 32 * bk + 8 * qq + 4 * (lane >> 5) + e;
                // (L_r^T)[i][k] = L_r[k][i], non-zero for k >= i
                float v = 0.f;
                if (i < M && k < M && k >= i) v = q[(size_t)k * M + i];
                dst[idx] = v;
            }
        }
}

__device__ void role_qmu_kl(const PreLayer& L, double* red) {
    const int nb = L.nb, M = L.M, R = L.R;
    for (int bk = 0; bk < nb; ++bk) {
        float* dst = L.QmuP + (size_t)bk * 1024;
        for (int idx = threadIdx.x; idx < 1024; idx += blockDim.x) {
            int qq = idx >> 8, lane = (idx >> 2) & 63, e = idx & 3;
            int r = lane & 31;
            int k = 32 * bk + 8 * qq + 4 * (lane >> 5) + e;
            float v = 0.f;
            if (r < R && k < M) v = L.q_mu[(size_t)k * R + r];
            dst[idx] = v;
        }
    }
    // KL = 0.5 * ( sum q_mu^2 - M R - sum log diag(L)^2 + sum tril(L)^2 )
    double acc = 0.0;
    for (size_t idx = threadIdx.x; idx < (size_t)M * R; idx += blockDim.x) {
        double v = L.q_mu[idx];
        acc += v * v;
    }
    for (size_t idx = threadIdx.x; idx < (size_t)R * M * M; idx += blockDim.x) {
        int rc = idx % ((size_t)M * M);
        int i = rc / M, j = rc % M;
        if (j > i) continue;
        double v = L.q_sqrt[idx];
        acc += v * v;
        if (i == j) acc -= log(v * v);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *L.kl = 0.5 * (red[0] - (double)M * R);
}

__global__ __launch_bounds__(256) void k_inv_pack(PreArgs args) {
    const PreLayer& L = args.L[blockIdx.x];
    const int role = blockIdx.y;
    double* sm = reinterpret_cast<double*>(smem_raw);
    if (role < L.nb) {
        double* S = sm;
        double* X = sm + NB * XLD;
        role_inverse(L, role, X, S);
    } else if (role < L.nb + L.R) {
        role_pack_LrT(L, role - L.nb);
    } else if (role == L.nb + L.R) {
        role_qmu_kl(L, sm);
    }
}

__global__ __launch_bounds__(256) void k_gauss_kl(const float* q_mu, const float* q_sqrt, int M, int R,
                                                   double* kl) {
    PreLayer L{};
    L.q_mu = q_mu; L.q_sqrt = q_sqrt; L.M = M; L.R = R; L.nb = 0; L.kl = kl;
    double* sm = reinterpret_cast<double*>(smem_raw);
    role_qmu_kl(L, sm);
}

static int ensure_lds_attr(const void* fn, size_t bytes) {
    // remember the largest size configured per kernel: hipFuncSetAttribute is not a stream operation and
    // must stay out of the steady state (and out of hipGraph capture)
    static const void* fns[8]; static size_t sizes[8]; static int nf = 0;
    for (int i = 0; i < nf; ++i) if (fns[i] == fn) { if (sizes[i] >= bytes) return IWVI_OK; break; }
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(%zu B LDS): %s", bytes, hipGetErrorString(e));
        return IWVI_ERR_LAUNCH;
    }
    int slot = -1;
    for (int i = 0; i < nf; ++i) if (fns[i] == fn) slot = i;
    if (slot < 0 && nf < 8) slot = nf++;
    if (slot >= 0) { fns[slot] = fn; sizes[slot] = bytes; }
    return IWVI_OK;
}

static size_t chol_lds_bytes(int Mp) {
    size_t d = 2 * NB * PLD + (size_t)(Mp - NB > 0 ? Mp - NB : 1) * PLD;
    if (Mp <= 128) d += (size_t)Mp * Mp;
    return d * sizeof(double);
}

}  // namespace iwvi

using namespace iwvi;

extern "C" size_t iwvi_gp_state_bytes(int M, int R) {
    if (M <= 0 || R <= 0) return 0;
    return state_layout(M, R).bytes;
}

extern "C" int iwvi_gp_state_offsets(int M, int R, size_t out[8]) {
    if (M <= 0 || R <= 0 || !out) { set_error("iwvi_gp_state_offsets: bad argument"); return IWVI_ERR_ARG; }
    StateLayout s = state_layout(M, R);
    out[0] = s.off_Lm; out[1] = s.off_Linv; out[2] = s.off_LinvP; out[3] = s.off_LrTP;
    out[4] = s.off_QmuP; out[5] = s.off_Zs; out[6] = s.off_invls; out[7] = s.off_kl;
    return IWVI_OK;
}

extern "C" int iwvi_gp_precompute(const iwvi_gp_desc* layers, int n_layers, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!layers || n_layers <= 0) { set_error("iwvi_gp_precompute: no layers"); return IWVI_ERR_ARG; }
    for (int base = 0; base < n_layers; base += IWVI_MAX_LAYERS) {
        PreArgs a{};
        a.n = n_layers - base < IWVI_MAX_LAYERS ? n_layers - base : IWVI_MAX_LAYERS;
        size_t lds_a = 0, lds_b = 0;
        int max_roles = 0;
        for (int l = 0; l < a.n; ++l) {
            const iwvi_gp_desc& d = layers[base + l];
            if (!d.Z || !d.lengthscales || !d.q_mu || !d.q_sqrt || !d.state) {
                set_error("iwvi_gp_precompute: layer %d has a null pointer", base + l); return IWVI_ERR_ARG;
            }
            if (d.M <= 0 || d.M > IWVI_MAX_M || d.D <= 0 || d.D > IWVI_MAX_D || d.R <= 0 || d.R > IWVI_MAX_R) {
                set_error("iwvi_gp_precompute: layer %d size out of range (M=%d<=%d, D=%d<=%d, R=%d<=%d)",
                          base + l, d.M, IWVI_MAX_M, d.D, IWVI_MAX_D, d.R, IWVI_MAX_R);
                return IWVI_ERR_ARG;
            }
            if (d.kern_type != IWVI_KERN_RBF && d.kern_type != IWVI_KERN_MATERN52) {
                set_error("iwvi_gp_precompute: unknown kernel type %d", d.kern_type); return IWVI_ERR_UNSUPPORTED;
            }
            StateLayout s = state_layout(d.M, d.R);
            char* st = (char*)d.state;
            PreLayer& L = a.L[l];
            L.Z = d.Z; L.ls = d.lengthscales; L.q_mu = d.q_mu; L.q_sqrt = d.q_sqrt;
            L.Lm = (double*)(st + s.off_Lm); L.Linv = (double*)(st + s.off_Linv);
            L.LinvP = (float*)(st + s.off_LinvP); L.LrTP = (float*)(st + s.off_LrTP);
            L.QmuP = (float*)(st + s.off_QmuP); L.Zs = (float*)(st + s.off_Zs);
            L.invls = (float*)(st + s.off_invls);
            L.kl = (double*)(st + s.off_kl);
            L.jitter = d.jitter; L.variance = d.variance;
            L.M = d.M; L.D = d.D; L.R = d.R; L.Mp = s.Mp; L.nb = s.nb; L.kern_type = d.kern_type;
            size_t la = chol_lds_bytes(s.Mp);
            size_t lb = sizeof(double) * ((size_t)NB * XLD + (size_t)s.Mp * XLD);
            if (lb < 256 * sizeof(double)) lb = 256 * sizeof(double);
            if (la > lds_a) lds_a = la;
            if (lb > lds_b) lds_b = lb;
            if (s.nb + d.R + 1 > max_roles) max_roles = s.nb + d.R + 1;
        }
        int rc;
        if ((rc = ensure_lds_attr((const void*)k_kuu_chol, lds_a)) != IWVI_OK) return rc;
        if ((rc = ensure_lds_attr((const void*)k_inv_pack, lds_b)) != IWVI_OK) return rc;
        hipLaunchKernelGGL(k_kuu_chol, dim3(a.n), dim3(1024), lds_a, stream, a);
        if ((rc = check_launch("k_kuu_chol")) != IWVI_OK) return rc;
        hipLaunchKernelGGL(k_inv_pack, dim3(a.n, max_roles), dim3(256), lds_b, stream, a);
        if ((rc = check_launch("k_inv_pack")) != IWVI_OK) return rc;
    }
    return IWVI_OK;
}

extern "C" int iwvi_rbf_gram_sym(const float* Z, const float* ls, float variance, double jitter,
                                 int kern_type, int M, int D, double* Kuu, void* stream_) {
    if (!Z || !ls || !Kuu || M <= 0 || D <= 0) { set_error("iwvi_rbf_gram_sym: bad argument"); return IWVI_ERR_ARG; }
    if (kern_type != IWVI_KERN_RBF && kern_type != IWVI_KERN_MATERN52) {
        set_error("iwvi_rbf_gram_sym: unknown kernel type %d", kern_type); return IWVI_ERR_UNSUPPORTED;
    }
    size_t n = (size_t)M * M;
    int grid = (int)((n + 255) / 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_gram_sym, dim3(grid), dim3(256), 0, (hipStream_t)stream_, Z, ls, variance, jitter,
                       kern_type, M, D, Kuu);
    return check_launch("k_gram_sym");
}

extern "C" size_t iwvi_chol_ws_bytes(int M) {
    if (M <= 0) return 0;
    size_t Mp = (size_t)round_up(M, NB);
    return 2 * Mp * Mp * sizeof(double);
}

extern "C" int iwvi_chol_factor(const double* A, double* Lout, int M, void* ws, void* stream_) {
    if (!A || !Lout || !ws || M <= 0) { set_error("iwvi_chol_factor: bad argument"); return IWVI_ERR_ARG; }
    if (M > 2048) { set_error("iwvi_chol_factor: M=%d too large (max 2048)", M); return IWVI_ERR_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    const int Mp = round_up(M, NB);
    double* Ap = (double*)ws;
    double* Ip = Ap + (size_t)Mp * Mp;       // receives the 16x16 diagonal-block inverses (by-product)
    size_t lds = sizeof(double) * (2 * NB * PLD + (size_t)(Mp - NB > 0 ? Mp - NB : 1) * PLD);
    int rc;
    if ((rc = ensure_lds_attr((const void*)k_chol_only, lds)) != IWVI_OK) return rc;
    int grid = (int)(((size_t)Mp * Mp + 255) / 256); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(k_pad_copy, dim3(grid), dim3(256), 0, stream, A, M, Ap, Mp);
    hipLaunchKernelGGL(k_chol_only, dim3(1), dim3(1024), lds, stream, Ap, Mp, Ip);
    hipLaunchKernelGGL(k_unpad_copy, dim3(grid), dim3(256), 0, stream, (const double*)Ap, Mp, Lout, M);
    return check_launch("k_chol_only");
}

extern "C" int iwvi_gauss_kl(const float* q_mu, const float* q_sqrt, int M, int R, double* kl, void* stream_) {
    if (!q_mu || !q_sqrt || !kl || M <= 0 || R <= 0) { set_error("iwvi_gauss_kl: bad argument"); return IWVI_ERR_ARG; }
    hipLaunchKernelGGL(k_gauss_kl, dim3(1), dim3(256), 256 * sizeof(double), (hipStream_t)stream_, q_mu, q_sqrt, M, R, kl);
    return check_launch("k_gauss_kl");
}

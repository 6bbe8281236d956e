// Per-step inducing-set factorisation for every GP layer of the model, in float64:
//   Kuu = K(Z,Z) + jitter I  ->  Lm = chol(Kuu)  ->  Lm^-1  ->  MFMA-fragment packed float32 operands
// Replaces (reference file:line) Kuu + tf.cholesky (temp_workaround.py:39,48), the operand side of
// tf.matrix_triangular_solve (:51), tf.matrix_band_part(q_sqrt) (:78) and gauss_kl (:186-188).
//
// ONE launch for all layers of a model: grid = (layer, role) + encoder blocks, 1024-thread workgroups.
//   role 0        Gram + Cholesky + inverses of the diagonal 16x16 blocks + packing of the forward-substitution
//                 stream and of the K_uf operand Z~   (the serial critical path)
//   role 1..R     tril(q_sqrt[r])^T packing + latent GP r's share of KL[q(u) || p(u)] (role 1 also packs q_mu^T)
//   encoder       the LV layer's MLP on the minibatch's data rows (layers.py:137-152), 256 rows per block
// The factorisation works on 16x16 blocks of the lower triangle (row stride 17 doubles: conflict-free
// ds_read_b64), resident in LDS for Mp <= 128 (78 KB) and in an L2-resident workspace otherwise.  It is
// left-looking and generates the matrix as it goes (chol_blocks):
//   * diagonal pass: one wave, row-per-lane in registers over a 64-row window (diagonal block, two block rows
//     below it, identity rows that come out as L^-T); pivot-row broadcasts are DPP row_newbcast operands of
//     v_fmac_f64 -- no LDS round trips, no barriers, no SGPR traffic inside the 16 steps;
//   * beside it, the other waves: Gram block column p+2 (f64 MFMA dot products + exp), column p+1's catch-up
//     with the factored columns (f64 MFMA), column p-1's packed float32 stream;
//   * after it: one wave per block row finishes column p and applies it to column p+1.
// Dense Lm / Lm^-1 (debug / API parity only, IWVI_GP_WANT_DENSE): Lm^-1 by recursive doubling over the
// block-triangular structure ([[A,0],[C,B]]^-1 = [[A^-1,0],[-B^-1 C A^-1, B^-1]]).
#include "iwvi_common.h"
#include "precompute_dev.h"
#include <cstdlib>

namespace iwvi {

// an Encoder MLP (layers.py:137-152) evaluated for every row of the minibatch in the same launch: it does not
// depend on the factorisation, so it rides on otherwise idle CUs instead of the critical path of the layer kernel
struct PreEnc {
    const float* XY; float* out; const float* W[IWVI_MAX_ENC]; const float* b[IWVI_MAX_ENC];
    long long rows; int dims[IWVI_MAX_ENC + 1]; int n_enc, Lw, blk0, nblk, act;
    // sampling tail (sample_X != nullptr): see iwvi_enc_desc
    const float* X; int Dx, K, sampled_kl, layer_index;
    unsigned long long seed; const unsigned long long* rng_state;
    float* sample_X; float* sample_kl; float* sample_z;
};
constexpr int PRE_MAX_ENC = 2;
struct PreArgs { PreLayer L[IWVI_MAX_LAYERS]; int n; int stop_after; int stamp_p; unsigned long long* stamps; PreEnc E[PRE_MAX_ENC]; int n_enc; };

static unsigned long long* g_pre_stamps = nullptr;   // diagnostic; see iwvi_debug_set_pre_stamps

// standalone whitened KL (iwvi_gauss_kl): sum over all R
__device__ void role_kl_only(const float* q_mu, const float* q_sqrt, int M, int R, double* kl, double* red) {
    double acc = 0.0;
    for (size_t idx = threadIdx.x; idx < (size_t)M * R; idx += blockDim.x) { double v = q_mu[idx]; acc += v * v; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int row = wave; row < R * M; row += nw) {
        const int i = row % M;
        const float* q = q_sqrt + (size_t)row * M;
        for (int j = lane; j <= i; j += 64) {
            double v = q[j];
            acc += v * v;
            if (j == i) acc -= log(v * v);
        }
    }
    const double tot = block_sum(acc, red);
    if (threadIdx.x == 0) *kl = 0.5 * (tot - (double)M * R);
}

// rows [ENC_ROWS * blk, ENC_ROWS * (blk + 1)) of one encoder: weights and both activation buffers in LDS, one
// (row, output unit) item per thread-iteration, one barrier per MLP layer
constexpr int ENC_ROWS = 256;
__device__ void role_encoder(const PreEnc& E, int blk) {
    float* wts = reinterpret_cast<float*>(smem_raw);
    int wtotal = 0, mdim = 2 * E.Lw;
    for (int l = 0; l < E.n_enc; ++l) {
        const int nW = E.dims[l] * E.dims[l + 1], nb = E.dims[l + 1];
        for (int i = threadIdx.x; i < nW; i += blockDim.x) wts[wtotal + i] = E.W[l][i];
        for (int i = threadIdx.x; i < nb; i += blockDim.x) wts[wtotal + nW + i] = E.b[l] ? E.b[l][i] : 0.f;
        wtotal += nW + nb;
        if (E.dims[l] > mdim) mdim = E.dims[l];
        if (E.dims[l + 1] > mdim) mdim = E.dims[l + 1];
    }
    mdim |= 1;                                             // odd row stride: conflict-free column walks
    float* act0 = wts + ((wtotal + 3) & ~3);
    float* act1 = act0 + ENC_ROWS * mdim;
    const long long row0 = (long long)blk * ENC_ROWS;
    const int nrows = (int)((E.rows - row0) < ENC_ROWS ? (E.rows - row0) : ENC_ROWS);
    const int d0 = E.dims[0];
    for (int idx = threadIdx.x; idx < nrows * d0; idx += blockDim.x) {
        const int r = idx / d0, i = idx - r * d0;
        act0[r * mdim + i] = E.XY[(row0 + r) * d0 + i];
    }
    __syncthreads();
    float* in = act0; float* out = act1;
    int off = 0;
    for (int l = 0; l < E.n_enc; ++l) {
        const int din = E.dims[l], dout = E.dims[l + 1];
        const float* W = wts + off; const float* b = W + din * dout;
        for (int idx = threadIdx.x; idx < nrows * dout; idx += blockDim.x) {
            const int r = idx / dout, o = idx - r * dout;
            float acc = b[o];
            for (int i = 0; i < din; ++i) acc = fmaf(in[r * mdim + i], W[i * dout + o], acc);
            if (l < E.n_enc - 1) acc = enc_act(acc, E.act);                // layers.py:143-144
            if (din == dout) acc += in[r * mdim + o];                       // layers.py:146-147
            out[r * mdim + o] = acc;
        }
        off += din * dout + dout;
        __syncthreads();
        float* t = in; in = out; out = t;
    }
    const int no = 2 * E.Lw;                               // [means | raw]; q_sqrt = softplus(raw - 3)
    for (int idx = threadIdx.x; idx < nrows * no; idx += blockDim.x) {
        const int r = idx / no, o = idx - r * no;
        E.out[(row0 + r) * no + o] = in[r * mdim + o];
    }
    if (!E.sample_X) return;
    // the LatentVariableLayer itself for these rows' K samples each (layers.py:83-103), from the activations still in LDS
    const unsigned long long step = E.rng_state ? E.rng_state[0] : 0ULL;
    const int Lw = E.Lw, Dx = E.Dx, Do = Dx + Lw, K = E.K;
    for (int idx = threadIdx.x; idx < nrows * K; idx += blockDim.x) {
        const int r = idx / K;
        const long long t = row0 * K + idx;
        float* xo = E.sample_X + t * Do;
        for (int d = 0; d < Dx; ++d) xo[d] = E.X[(row0 + r) * Dx + d];
        float klsum = 0.f;
        for (int q = 0; 4 * q < Lw; ++q) {
            float z4[4];
            draw_normal4(E.seed, step, E.layer_index, t, q, z4);
            for (int e = 0; e < 4 && 4 * q + e < Lw; ++e) {
                const int l = 4 * q + e;
                const float mu = in[r * mdim + l], sg = softplus_f(in[r * mdim + Lw + l] - 3.f), z = z4[e];
                const float w = fmaf(z, sg, mu);                                   // layers.py:86-87
                float kl;
                if (E.sampled_kl) kl = -0.5f * z * z - __logf(sg) + 0.5f * w * w;  // log q(W) - log p(W), :98-100
                else kl = 0.5f * (sg * sg + mu * mu - 1.f) - __logf(sg);           // KL(N(mu,sg)||N(0,1)), :101-103
                klsum += kl;
                xo[Dx + l] = w;
                if (E.sample_z) E.sample_z[t * Lw + l] = z;
            }
        }
        E.sample_kl[t] = klsum;
    }
}

__global__ __launch_bounds__(1024) void k_precompute(PreArgs args) {
    if ((int)blockIdx.x >= args.n) {                       // encoder blocks follow the GP layers in x
        if (blockIdx.y != 0) return;
        const int b = blockIdx.x - args.n;
        for (int e = 0; e < args.n_enc; ++e)
            if (b >= args.E[e].blk0 && b < args.E[e].blk0 + args.E[e].nblk) role_encoder(args.E[e], b - args.E[e].blk0);
        return;
    }
    const PreLayer& L = args.L[blockIdx.x];
    const int role = blockIdx.y;
    if (role == 0) {
        if (L.flags & IWVI_GP_REUSE_FACTOR) return;         // (the state's factorisation is current: q(u) images only)
        if (L.Mp <= 128) role_factor<true>(L, args.stop_after, args.stamps, args.stamp_p); else role_factor<false>(L, args.stop_after, args.stamps, args.stamp_p);
    } else if (role <= L.R) {
        if (L.flags & IWVI_GP_FACTOR_ONLY) return;
        role_pack_r(L, role - 1, reinterpret_cast<double*>(smem_raw));
    }
}

// The split-f16 slabs of the super-block solve's dense part (iwvi_common.h: sb16_slabs), for every layer with M > 240, from the factor
// the launch before left in the layer's workspace (block storage; the blocks between different super-blocks still hold L there) or, for a
// state precomputed with IWVI_GP_WANT_DENSE / _LM, from its dense Lm.  A launch of its own: k_precompute is not touched by it
// (at 128 VGPRs with spills, a source change anywhere in that kernel moves the M <= 128 path by +-1 us: profiles/r04_precompute_notes.txt).
struct Ls16One { const double* blk; const double* Lm; unsigned short* dst; const float* LsP; const float* variance_dev; float variance; int nbk, Mp, M, first; };
struct Ls16All { Ls16One L[IWVI_MAX_LAYERS]; int n; };
__global__ __launch_bounds__(256) void k_pack_ls16(const Ls16All a) {
    int li = 0;
    while (li + 1 < a.n && (int)blockIdx.x >= a.L[li + 1].first) ++li;
    const Ls16One& L = a.L[li];
    const int nbk = L.nbk;
    int s = ((int)blockIdx.x - L.first) * 4 + (threadIdx.x >> 6);           // slab of this wave
    const int lane = threadIdx.x & 63;
    if (s >= sb16_slabs(nbk)) {
        // the triangular part's blocks left of the diagonal (iwvi_common.h: sb16_tri_blocks), from the fp32 blocks k_sb_inv / k_precompute
        // packed into LsP in the launches before: same lane, same four values -- scaled by 2^lg and split
        int tb = s - sb16_slabs(nbk);
        if (tb >= sb16_tri_blocks(nbk)) return;
        const int tb0 = tb;
        int I = 0, off = 0;
        for (;; ++I) {
            const int r0 = 8 * I, nr = nbk - r0 < 8 ? nbk - r0 : 8, cnt = nr * (nr - 1) / 2;
            if (tb < cnt) break;
            tb -= cnt; off += nr * r0 + nr * (nr + 1) / 2;
        }
        const int r0 = 8 * I, nr = nbk - r0 < 8 ? nbk - r0 : 8;
        int w = 1;
        while (w * (w + 1) / 2 <= tb) ++w;
        const int q = tb - w * (w - 1) / 2;
        const float var = L.variance_dev ? *L.variance_dev : L.variance;
        const float si = ldexpf(1.f, (int)ceilf(0.5f * log2f(fmaxf(var, 1e-30f))));
        const float4 v = reinterpret_cast<const float4*>(L.LsP)[(size_t)(off + nr * r0 + w * (w + 1) / 2 + q) * 64 + lane];
        const float x[4] = {v.x * si, v.y * si, v.z * si, v.w * si};
        pk_f16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xc = fminf(fmaxf(x[e], -65504.f), 65504.f);
            const _Float16 hh = (_Float16)xc;
            o[e] = hh; o[4 + e] = (_Float16)(xc - (float)hh);
        }
        unsigned short* dst = L.dst + (size_t)sb16_slabs(nbk) * 1024 + (size_t)tb0 * 512;
        *reinterpret_cast<float4*>(dst + lane * 8) = as_f4(o);
        return;
    }
    int I = 1;
    for (;; ++I) { const int nr = nbk - 8 * I < 8 ? nbk - 8 * I : 8, cnt = nr * 4 * I; if (s < cnt) break; s -= cnt; }
    const int row = s / (4 * I), kc = s - row * 4 * I;                      // block row 8 I + row, blocks 2 kc, 2 kc + 1
    const int bi = 8 * I + row;
    const float var = L.variance_dev ? *L.variance_dev : L.variance;
    const float sc = ldexpf(1.f, 10 - (int)ceilf(0.5f * log2f(fmaxf(var, 1e-30f))));   // 2^ea of a layer with M > 128 (role_factor: IWVI_CST_SA)
    const int i = lane & 15, g = lane >> 4;
    pk_f16x8 h1, h2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 32 * kc + 8 * g + e, bk = k >> 4;
        const int r = 16 * bi + i;
        double v = L.Lm ? L.Lm[(size_t)r * L.Mp + k] : L.blk[boff(bi, bk) + i * BLD + (k & 15)];
        if (r >= L.M || k >= L.M) v = 0.0;                                   // identity padding: nothing below the diagonal
        const float x = -(float)v * sc;
        const _Float16 hh = (_Float16)x;
        h1[e] = hh; h2[e] = (_Float16)(x - (float)hh);
    }
    unsigned short* dst = L.dst + ((size_t)blockIdx.x - L.first) * 4 * 1024 + (size_t)(threadIdx.x >> 6) * 1024;
    *reinterpret_cast<float4*>(dst + lane * 8) = as_f4(h1);
    *reinterpret_cast<float4*>(dst + 512 + lane * 8) = as_f4(h2);
}

// ---- M > 240: the inverses of the 128 x 128 diagonal super-blocks of Lm, packed as the triangular part of the layer kernel's solve stream
// (csrc/dgp_forward.hip: a_I = (L_II)^-1 r_I).  One workgroup (4 waves) per 16-column block J of a super-block, like k_linv:
//   X_J = D_J^-1,   X_t = -D_t^-1 sum_{J <= u < t} L(t, u) X_u   (float64 MFMA products, the sum dealt to the four waves)
// from the factor the launch before left in the layer's workspace (block storage; diagonal-block inverses transposed in `dinv`), written
// straight in A-fragment order behind the dense part of super-block I.  In k_precompute the same inverses were three doubling steps on the
// factorising workgroup's one CU: 53 us at M = 256 (two super-blocks), 16 workgroups here.
struct SbInvOne { const double* ws; float* LsP; int nbk, Mp, M, first; double* X64; };   // X64 (natural-gradient step): the inverse as float64 blocks (block storage), no LsP
struct SbInvAll { SbInvOne L[IWVI_MAX_LAYERS]; int n; };
__global__ __launch_bounds__(256) void k_sb_inv(const SbInvAll a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sbinv_smem[];
    int li = 0;
    while (li + 1 < a.n && (int)blockIdx.x >= a.L[li + 1].first) ++li;
    const SbInvOne& L = a.L[li];
    const int nbk = L.nbk, wg = (int)blockIdx.x - L.first;
    const int I = wg >> 3, J = wg & 7, r0 = 8 * I, nr = (nbk - r0 < 8) ? nbk - r0 : 8;
    if (J >= nr) return;
    const int nb = nr - J;                                       // blocks of this block column: rows J .. nr - 1 of the super-block
    const WsLayout w = ws_layout(L.Mp);
    const double* blk = L.ws + w.blk;
    const double* dinvT = L.ws + w.dinv;                         // [nbk][BLK]: L_pp^-T of the factorisation's diagonal passes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* X = reinterpret_cast<double*>(sbinv_smem);           // [nb][BLK]  the finished blocks of this block column
    double* Dv = X + (size_t)nb * BLK;                           // [nb][BLK]  D_t^-1
    double* Ls = Dv + (size_t)nb * BLK;                          // [nb (nb - 1) / 2][BLK]  L(t, u), u < t
    double* part = Ls + (size_t)(nb * (nb - 1) / 2) * BLK;       // [4][BLK]
    {   // everything this column needs, one round trip: all loads of a thread issued before its first store
        const int rr = tid >> 4, cc = tid & 15;
        constexpr int MAXB = 36;
        const int ntot = nb * (nb + 1) / 2;
        double v[MAXB];
#pragma unroll
        for (int q = 0; q < MAXB; ++q) {
            int t = 0;
            while ((t + 1) * (t + 2) / 2 <= q) ++t;
            const int u = q - t * (t + 1) / 2;
            const int bt = r0 + J + t, bu = r0 + J + u;
            v[q] = 0.0;
            if (q < ntot) v[q] = (u == t) ? dinvT[(size_t)bt * BLK + cc * BLD + rr] : blk[boff(bt, bu) + rr * BLD + cc];   // (transposed: D^-1[rr][cc])
        }
#pragma unroll
        for (int q = 0; q < MAXB; ++q) {
            int t = 0;
            while ((t + 1) * (t + 2) / 2 <= q) ++t;
            const int u = q - t * (t + 1) / 2;
            if (q < ntot) {
                if (u == t) Dv[(size_t)t * BLK + rr * BLD + cc] = (cc <= rr) ? v[q] : 0.0;
                else Ls[(size_t)(t * (t - 1) / 2 + u) * BLK + rr * BLD + cc] = v[q];
            }
        }
        if (tid < 256) X[rr * BLD + cc] = 0.0;
    }
    __syncthreads();
    for (int i = tid; i < BLK; i += 256) X[i] = Dv[i];            // X_0 = D_J^-1
    __syncthreads();
    // Right-looking: once X_u is known every block below it takes its term, S_t += L(t, u) X_u -- block t lives in the registers of wave
    // (t - 1) & 3 (two blocks per wave at most) -- and the owner of block u + 1 closes it: X_{u+1} = -D_{u+1}^-1 S_{u+1}.  One barrier per step
    // (round 5; the left-looking form dealt the sum of a step to the four waves and paid three barriers per step: 9.3 us for eight blocks).
    {
        f64x4 accA = {0.0, 0.0, 0.0, 0.0}, accB = {0.0, 0.0, 0.0, 0.0};
        const int tA = wave + 1, tB = wave + 5;
        for (int u = 0; u + 1 < nb; ++u) {
            if (tA > u && tA < nb) blk_mma<false>(accA, Ls + (size_t)(tA * (tA - 1) / 2 + u) * BLK, X + (size_t)u * BLK, lane, 1.0);
            if (tB > u && tB < nb) blk_mma<false>(accB, Ls + (size_t)(tB * (tB - 1) / 2 + u) * BLK, X + (size_t)u * BLK, lane, 1.0);
            if (wave == (u & 3)) {                               // owner of block u + 1
                blk_store(part + wave * BLK, (u + 1 == tA) ? accA : accB, lane);
                f64x4 x = {0.0, 0.0, 0.0, 0.0};
                blk_mma<false>(x, Dv + (size_t)(u + 1) * BLK, part + wave * BLK, lane, -1.0);
                blk_store(X + (size_t)(u + 1) * BLK, x, lane);
            }
            __syncthreads();
        }
    }
    if (L.X64) {                                                 // (one super-block: nbk <= 8)
        for (int it = tid; it < nb * BLK; it += 256) {
            const int t = it / BLK, e = it - t * BLK;
            L.X64[boff(J + t, J) + e] = X[(size_t)t * BLK + e];
        }
        return;
    }
    // packed fp32 blocks: row w = J + t of the super-block's triangle, block q = J: position nr r0 + w (w + 1) / 2 + J behind the super-block's start
    int off = 0;
    for (int i2 = 0; i2 < I; ++i2) { const int rr0 = 8 * i2, nn = (nbk - rr0 < 8) ? nbk - rr0 : 8; off += nn * rr0 + nn * (nn + 1) / 2; }
    float4* dst = reinterpret_cast<float4*>(L.LsP);
    const bool full = (L.M == L.Mp);
    for (int it = tid; it < nb * 64; it += 256) {
        const int t = it >> 6, ln = it & 63, ii = ln & 15, k0 = 4 * (ln >> 4);
        const int wrow = J + t;
        float v[4];
#pragma unroll
        for (int sgm = 0; sgm < 4; ++sgm) {
            const int i = 16 * (r0 + wrow) + ii, k = 16 * (r0 + J) + k0 + sgm;
            float f = (k <= i) ? (float)X[(size_t)t * BLK + ii * BLD + k0 + sgm] : 0.f;
            if (!full && (i >= L.M || k >= L.M)) f = (i == k) ? 1.f : 0.f;                 // padded rows solve to 0 against k = 0
            v[sgm] = f;
        }
        dst[(size_t)(off + nr * r0 + wrow * (wrow + 1) / 2 + J) * 64 + ln] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

__global__ __launch_bounds__(256) void k_gauss_kl(const float* q_mu, const float* q_sqrt, int M, int R,
                                                   double* kl) {
    role_kl_only(q_mu, q_sqrt, M, R, kl, reinterpret_cast<double*>(smem_raw));
}

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


// standalone Cholesky (K2): dense A -> block storage in ws -> factor -> dense lower L
__global__ __launch_bounds__(1024) void k_chol_only(const double* A, double* Lout, int M, int Mp, double* ws) {
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int nbk = Mp / NB;
    double* rinv = reinterpret_cast<double*>(smem_raw);
    double* blk = ws;
    for (int idx = tid; idx < nbk * nbk * 256; idx += nthreads) {
        const int b = idx >> 8, e = idx & 255;
        const int bi = b / nbk, bj = b - bi * nbk;
        if (bj > bi) continue;
        const int i = NB * bi + (e >> 4), j = NB * bj + (e & 15);
        blk[boff(bi, bj) + (e >> 4) * BLD + (e & 15)] = (i < M && j < M) ? A[(size_t)i * M + j] : ((i == j) ? 1.0 : 0.0);
    }
    __syncthreads();
    chol_blocks(blk, nbk, rinv, ws + (size_t)(nbk * (nbk + 1) / 2) * BLK, tid, nthreads, NoGen(), NoPost(), NoTail());
    for (int idx = tid; idx < M * M; idx += nthreads) {
        const int i = idx / M, k = idx - i * M;
        Lout[idx] = (k <= i) ? blk_get(blk, i, k) : 0.0;
    }
}

// ------------------------------------------------------------------------------------------------------------
// The natural-gradient step of one latent GP (iwvi_natgrad_step; algebra in csrc/backward.hip) in ONE workgroup, everything in
// LDS, for Mp <= 128 -- instead of 14 launches (conversions, three products, factorisation and triangular inverse in L2):
//   1  Qrev = J (I + gamma sym(Phi(L^T Lbar))) J, lower blocks, straight from the float32 inputs (f64 MFMA block products)
//   2  Qrev = C C^T              chol_blocks (the factorisation of the precompute launch)
//   3  C^-1                      invert_blocks
//   4  mu' = m - gamma L z,      z = J C^-T C^-1 J (L^T mbar)        (four matrix-vector products)
//   5  L' = L W,                 W[k][j] = C^-1[M-1-j][M-1-k]         (block products, held in registers until every wave has read L)
// blockIdx.x = latent GP r.  Lbar = -dq_sqrt, mbar = -dq_mu (the ELBO is maximised).
using f32x4g = __attribute__((ext_vector_type(4))) float;
// one 16 x 16 block (bi, bj) of Qrev by one wave, written to block storage at `dst` (LDS or global): 16 x 16 x M on the float64 matrix cores
__device__ __forceinline__ void ng_q_block(const float* __restrict__ Lf, const float* __restrict__ Gf, int M, double gamma, int bi, int bj, int lane, double* dst) {
    const int ri = lane & 15, g = lane >> 4;
    const int ig = NB * bi + ri, jg = NB * bj + ri;          // this lane's A row (i) and B column (j)
    const int q = M - 1 - ig, p = M - 1 - jg;
    int k0 = M - 1 - (NB * bj + NB - 1); if (k0 < 0) k0 = 0; k0 &= ~3;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int kb = k0; kb < M; kb += 64) {                    // 16 k-steps' operands per round trip (32 would not fit 128 VGPRs)
        double av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = kb + 4 * u + g;
            av[u] = (k < M && q >= 0 && k >= q) ? (double)Gf[(size_t)k * M + q] : 0.0;     // A[i][k] = dq[k][q]
            bv[u] = (k < M && p >= 0 && k >= p) ? (double)Lf[(size_t)k * M + p] : 0.0;     // B[k][j] = L[k][p]
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) if (kb + 4 * u < M) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {                            // acc[e] = C[g + 4e][ri]
        const int i = NB * bi + g + 4 * e, j = NB * bj + ri;
        double v = (i == j) ? 1.0 : 0.0;
        if (i < M && j < M) v -= gamma * acc[e];
        dst[(g + 4 * e) * BLD + ri] = v;
    }
}
// step 1 on many CUs (one wave per block, grid (blocks, R)): in the one-workgroup kernel it is 36 x 32 float64 MFMAs of 64 clocks on
// four SIMDs -- 20 of its 91 us.  The blocks go to the step's workspace in block storage; k_natgrad_small copies them into LDS.
__global__ __launch_bounds__(64) void k_ng_qbuild(const float* __restrict__ q_sqrt, const float* __restrict__ dq_sqrt, int M, double gamma, double* __restrict__ qws, int ntri) {
    const int o = blockIdx.x, r = blockIdx.y;
    int bi = 0; while ((bi + 1) * (bi + 2) / 2 <= o) ++bi;
    const int bj = o - bi * (bi + 1) / 2;
    double* dst = qws + (size_t)r * ntri * BLK + boff(bi, bj);
    ng_q_block(q_sqrt + (size_t)r * M * M, dq_sqrt + (size_t)r * M * M, M, gamma, bi, bj, threadIdx.x, dst);
    // the block's padding column (BLD = 17 doubles per row): k_natgrad_small copies whole blocks out of this caller-allocated, never
    // initialised workspace -- nothing reads the padding, but it should not carry whatever the allocation held
    if (threadIdx.x < NB) for (int c = NB; c < BLD; ++c) dst[threadIdx.x * BLD + c] = 0.0;
}
// the same block by FOUR waves, each a quarter of the contraction (k in chunks of 32, dealt round robin), the partial blocks added in a fixed
// order: the spread step's first launch (one wave per block walked up to 128 k behind two dependent load round trips: 12.5 us)
__global__ __launch_bounds__(256) void k_ng_qbuild4(const float* __restrict__ q_sqrt, const float* __restrict__ dq_sqrt, int M, double gamma, double* __restrict__ qws, int ntri) {
    __shared__ double part[3][BLK];
    const int o = blockIdx.x, r = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int bi = 0; while ((bi + 1) * (bi + 2) / 2 <= o) ++bi;
    const int bj = o - bi * (bi + 1) / 2;
    const float* Lf = q_sqrt + (size_t)r * M * M; const float* Gf = dq_sqrt + (size_t)r * M * M;
    double* dst = qws + (size_t)r * ntri * BLK + boff(bi, bj);
    const int ri = lane & 15, g = lane >> 4;
    const int ig = NB * bi + ri, jg = NB * bj + ri;
    const int q = M - 1 - ig, p = M - 1 - jg;
    int k0 = M - 1 - (NB * bj + NB - 1); if (k0 < 0) k0 = 0; k0 &= ~3;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int kb = k0 + 32 * wave; kb < M; kb += 128) {
        double av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = kb + 4 * u + g;
            av[u] = (k < M && q >= 0 && k >= q) ? (double)Gf[(size_t)k * M + q] : 0.0;     // A[i][k] = dq[k][q]
            bv[u] = (k < M && p >= 0 && k >= p) ? (double)Lf[(size_t)k * M + p] : 0.0;     // B[k][j] = L[k][p]
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (kb + 4 * u < M) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
    }
    if (wave > 0) blk_store(part[wave - 1], acc, lane);
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w_ = 0; w_ < 3; ++w_) { const f64x4 v = blk_load(part[w_], lane); acc += v; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {                            // acc[e] = C[g + 4e][ri]
        const int i = NB * bi + g + 4 * e, j = NB * bj + ri;
        double v = (i == j) ? 1.0 : 0.0;
        if (i < M && j < M) v -= gamma * acc[e];
        dst[(g + 4 * e) * BLD + ri] = v;
    }
    if (lane < NB) for (int c = NB; c < BLD; ++c) dst[lane * BLD + c] = 0.0;
}
__global__ __launch_bounds__(1024) void k_natgrad_small(float* q_mu, float* q_sqrt, const float* __restrict__ dq_mu, const float* __restrict__ dq_sqrt,
                                                        int M, int R, double gamma, int stop, const double* __restrict__ qws) {
    const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6;
    const int r = blockIdx.x, Mp = round_up(M, NB);
    const WsLayout w = ws_layout(Mp);
    const int nbk = w.nbk, ntri = nbk * (nbk + 1) / 2;
    double* sm = reinterpret_cast<double*>(smem_raw);
    double* rinv = sm;
    double* blk = sm + Mp + w.blk;
    double* dinv = sm + Mp + w.dinv;
    double* tbuf = sm + Mp + w.tbuf;
    double* va = sm + Mp + w.total;                          // four vectors of Mp doubles
    double* vb = va + Mp; double* vc = vb + Mp; double* vd = vc + Mp;
    double* ypart = vd + Mp;                                 // [16 waves][128] partial vectors of the first product
    float* Lf = q_sqrt + (size_t)r * M * M;
    const float* Gf = dq_sqrt + (size_t)r * M * M;
    const int ri = lane & 15, g = lane >> 4;
    auto tri_decode = [](int o, int& bi, int& bj) { bi = 0; while ((bi + 1) * (bi + 2) / 2 <= o) ++bi; bj = o - bi * (bi + 1) / 2; };

    // ---- 1: Qrev(bi, bj)[i][j] = delta - gamma * sum_k L[k][p] dq[k][q],  p = M-1-(16bj+j), q = M-1-(16bi+i)   (p >= q on and below the diagonal)
    if (qws) {                                                // formed by k_ng_qbuild on many CUs: copy the lower blocks (same storage order)
        const double* src = qws + (size_t)r * ntri * BLK;
        for (int i = tid; i < ntri * BLK; i += nthreads) blk[i] = src[i];
    } else
    for (int o = wave; o < ntri; o += nw) {
        int bi, bj; tri_decode(o, bi, bj);
        ng_q_block(Lf, Gf, M, gamma, bi, bj, lane, blk + boff(bi, bj));
    }
    // the vectors' inputs meanwhile
    if (tid < Mp) { va[tid] = tid < M ? (double)q_mu[(size_t)tid * R + r] : 0.0; vb[tid] = tid < M ? -(double)dq_mu[(size_t)tid * R + r] : 0.0; }
    __syncthreads();
    if (stop == 1) return;
    // ---- 2, 3
    chol_blocks(blk, nbk, rinv, dinv, tid, nthreads, NoGen(), NoPost(), NoTail());
    if (stop == 2) return;
    invert_blocks(blk, dinv, tbuf, rinv, nbk, tid, nthreads);
    if (stop == 3) return;
    // ---- 4: y1 = L^T mbar (reversed into vc), y2 = C^-1 vc (vd), y3 = C^-T y2 (reversed into vb), mu' = m - gamma L vb.
    //      A wave per output entry (8 each), lanes over the contraction index, every load of a wave's entries in flight together:
    //      a thread per entry walking its row was a chain of 128 dependent global / LDS round trips per product.
    auto wsum = [](double v) { for (int o_ = 32; o_ > 0; o_ >>= 1) v += __shfl_xor(v, o_, 64); return v; };
    constexpr int NPW = 128 / 16;                            // entries per wave (Mp <= 128, 16 waves)
    {   // y1[p] = sum_{k >= p} L[k][p] mbar[k]: lanes over p (rows of L read coalesced), wave w takes the rows k = w, w + 16, ..;
        double p0 = 0.0, p1 = 0.0;
        float l0[NPW], l1[NPW];
#pragma unroll
        for (int u = 0; u < NPW; ++u) {
            const int k = wave + nw * u;
            l0[u] = (k < M && lane <= k) ? Lf[(size_t)k * M + lane] : 0.f;
            l1[u] = (k < M && lane + 64 <= k) ? Lf[(size_t)k * M + lane + 64] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NPW; ++u) {
            const int k = wave + nw * u;
            const double mk = k < M ? vb[k] : 0.0;
            p0 = fma((double)l0[u], mk, p0); p1 = fma((double)l1[u], mk, p1);
        }
        ypart[wave * 128 + lane] = p0; ypart[wave * 128 + 64 + lane] = p1;
        __syncthreads();
        if (tid < Mp) {
            double v = 0.0;
#pragma unroll
            for (int w_ = 0; w_ < 16; ++w_) v += ypart[w_ * 128 + tid];
            if (tid < M) vc[M - 1 - tid] = v; else vc[tid] = 0.0;
        }
    }
    __syncthreads();
    {
        double part[NPW];
#pragma unroll
        for (int u = 0; u < NPW; ++u) {                      // y2[i] = sum_{k <= i} Cinv[i][k] vc[k]
            const int i_ = wave + nw * u;
            double a_ = 0.0;
            for (int k = lane; k < M; k += 64) if (i_ < M && k <= i_) a_ = fma(inv_get(blk, dinv, i_, k), vc[k], a_);
            part[u] = a_;
        }
#pragma unroll
        for (int u = 0; u < NPW; ++u) { const int i_ = wave + nw * u; const double v = wsum(part[u]); if (lane == 0 && i_ < Mp) vd[i_] = v; }
    }
    __syncthreads();
    {
        double part[NPW];
#pragma unroll
        for (int u = 0; u < NPW; ++u) {                      // y3[k] = sum_{i >= k} Cinv[i][k] y2[i]
            const int k_ = wave + nw * u;
            double a_ = 0.0;
            for (int i = lane; i < M; i += 64) if (k_ < M && i >= k_) a_ = fma(inv_get(blk, dinv, i, k_), vd[i], a_);
            part[u] = a_;
        }
#pragma unroll
        for (int u = 0; u < NPW; ++u) { const int k_ = wave + nw * u; const double v = wsum(part[u]); if (lane == 0 && k_ < M) vb[M - 1 - k_] = v; }   // (mbar is no longer needed)
    }
    __syncthreads();
    {
        double part[NPW];
#pragma unroll
        for (int u = 0; u < NPW; ++u) {                      // (L z)[i] = sum_{k <= i} L[i][k] z[k]
            const int i_ = wave + nw * u;
            double a_ = 0.0;
            for (int k = lane; k < M; k += 64) if (i_ < M && k <= i_) a_ = fma((double)Lf[(size_t)i_ * M + k], vb[k], a_);
            part[u] = a_;
        }
#pragma unroll
        for (int u = 0; u < NPW; ++u) { const int i_ = wave + nw * u; const double v = wsum(part[u]); if (lane == 0 && i_ < M) vd[i_] = va[i_] - gamma * v; }   // mu' (y2 is no longer needed)
    }
    if (stop == 4) return;
    // ---- 5: L'(bi, bj)[i][j] = sum_{k = 16bj .. 16bi+15} L[16bi+i][k] W[k][16bj+j]
    f64x4 outv[3];
    for (int t = 0; t < 3; ++t) {
        const int o = wave + t * nw;
        outv[t] = f64x4{0.0, 0.0, 0.0, 0.0};
        if (o >= ntri) continue;
        int bi, bj; tri_decode(o, bi, bj);
        const int ig = NB * bi + ri, jg = NB * bj + ri, k0 = NB * bj;
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        // k order of a 16-deep chunk c: MFMA step s contracts k = kb + 16c + 4g + s (g = lane >> 4), so that a lane's four A entries
        // are ONE 16-byte load of its row of L (16 rows x 64 B per wave-load instead of 16 rows x 4 B)
        const bool al16 = (M & 3) == 0;
        for (int kb = k0; kb < NB * bi + NB; kb += 64) {
            double av[16], bv[16];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int kq = kb + 16 * c + 4 * g;              // this lane's four consecutive k
                float a4[4] = {0.f, 0.f, 0.f, 0.f};
                if (ig < M && kq < NB * bi + NB && kq < M) {
                    if (al16) { const f32x4g v = *reinterpret_cast<const f32x4g*>(Lf + (size_t)ig * M + kq); a4[0] = v[0]; a4[1] = v[1]; a4[2] = v[2]; a4[3] = v[3]; }
                    else { for (int s_ = 0; s_ < 4; ++s_) if (kq + s_ < M) a4[s_] = Lf[(size_t)ig * M + kq + s_]; }
                }
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    const int k = kq + s_;
                    const bool in = k < NB * bi + NB && k < M;
                    av[4 * c + s_] = (in && k <= ig) ? (double)a4[s_] : 0.0;                                        // A[i][k] = L[i][k]
                    bv[4 * c + s_] = (in && jg < M && k >= jg) ? inv_get(blk, dinv, M - 1 - jg, M - 1 - k) : 0.0;     // B[k][j] = W[k][j]
                }
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) if (kb + 16 * (u >> 2) < NB * bi + NB) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
        }
        outv[t] = acc;
    }
    __syncthreads();                                         // every read of the old L is done
    for (int t = 0; t < 3; ++t) {
        const int o = wave + t * nw;
        if (o >= ntri) continue;
        int bi, bj; tri_decode(o, bi, bj);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = NB * bi + g + 4 * e, j = NB * bj + ri;
            if (i < M && j < M) Lf[(size_t)i * M + j] = (j <= i) ? (float)outv[t][e] : 0.f;
        }
    }
    for (int idx = tid; idx < M * M; idx += nthreads) {      // (blocks above the diagonal: zero, as tril() leaves them)
        const int i = idx / M, j = idx - i * M;
        if ((j >> 4) > (i >> 4)) Lf[idx] = 0.f;
    }
    __syncthreads();
    if (tid < M) q_mu[(size_t)tid * R + r] = (float)vd[tid];
}

// ------------------------------------------------------------------------------------------------------------
// The same step SPREAD over the chip (round 5, M <= 128): k_natgrad_small is 88 us on one CU -- factorisation 15, triangular inverse 17,
// four matrix-vector products 22, L' = L W 28 -- of which only the factorisation is a serial chain.  Four launches instead:
//   k_ng_qbuild   Qrev blocks, one wave per block (as before)
//   k_ng_chol     Qrev = C C^T in LDS (chol_blocks), factor and L_pp^-T blocks to the workspace -- what k_sb_inv reads
//   k_sb_inv      C^-1, one workgroup per 16-column block (the super-block inverse of the forward's operands, float64 output)
//   k_ng_rows     one workgroup per 16-row block of L' = L W (a wave per tile, C^-1 staged in LDS) and its share of t = L'^T mbar
//   k_ng_vec      t = the sum of the shares, mu' = m - gamma L' t  (csrc/backward.hip: the algebra; two products with L' instead of four)
// (back to back in a replayed graph the launches follow each other within 0.1 us; a last-arriving workgroup doing k_ng_vec's work inside
//  k_ng_rows read the other workgroups' output through agent-scope loads -- 26 us for the launch instead of 6 + 4)
__global__ __launch_bounds__(1024) void k_ng_chol(const double* __restrict__ qws, double* __restrict__ fac, int Mp) {
    const int tid = threadIdx.x, nthreads = blockDim.x, r = blockIdx.x;
    const WsLayout w = ws_layout(Mp);
    const int nbk = w.nbk, ntri = nbk * (nbk + 1) / 2;
    double* sm = reinterpret_cast<double*>(smem_raw);
    double* rinv = sm;
    double* blk = sm + Mp + w.blk;
    double* dinv = sm + Mp + w.dinv;
    const double* src = qws + (size_t)r * ntri * BLK;
    for (int i = tid; i < ntri * BLK; i += nthreads) blk[i] = src[i];
    __syncthreads();
    chol_blocks(blk, nbk, rinv, dinv, tid, nthreads, NoGen(), NoPost(), NoTail());
    __syncthreads();
    double* dst = fac + (size_t)r * w.total;
    for (int i = tid; i < ntri * BLK; i += nthreads) dst[w.blk + i] = blk[i];
    for (int i = tid; i < nbk * BLK; i += nthreads) dst[w.dinv + i] = dinv[i];
}

struct NgRowsArgs {
    float* q_mu; float* q_sqrt; const float* dq_mu; const double* X64; double* Lp64; double* tpart;
    int M, Mp, R; double gamma;
};
__global__ __launch_bounds__(512) void k_ng_rows(const NgRowsArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bi = blockIdx.x, r = blockIdx.y, M = a.M, Mp = a.Mp, nbk = Mp / NB, ntri = nbk * (nbk + 1) / 2;
    float* Lf = a.q_sqrt + (size_t)r * M * M;
    double* Lp = a.Lp64 + (size_t)r * ntri * BLK;
    double* tp = a.tpart + ((size_t)r * nbk + bi) * Mp;
    // C^-1 (lower blocks, diagonal blocks included) into LDS, whole blocks at 16 bytes per lane: W[k][j] = C^-1[M-1-j][M-1-k] is read transposed and
    // reversed -- from global memory that was sixteen 8-byte loads per lane and step, each across 16 rows
    double* Xs = reinterpret_cast<double*>(smem_raw);
    {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2* src = reinterpret_cast<const d2*>(a.X64 + (size_t)r * ntri * BLK);
        d2* dst = reinterpret_cast<d2*>(Xs);
        for (int i = tid; i < ntri * BLK / 2; i += 512) dst[i] = src[i];
    }
    __syncthreads();
    const int ri = lane & 15, g = lane >> 4, bj = wave;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    if (bj <= bi) {
        // L'(bi, bj)[i][j] = sum_{k = 16bj .. 16bi+15} L[16bi+i][k] W[k][16bj+j]
        const int ig = NB * bi + ri, jg = NB * bj + ri, k0 = NB * bj;
        const bool al16 = (M & 3) == 0;
        for (int kb = k0; kb < NB * bi + NB; kb += 64) {
            double av[16], bv[16];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int kq = kb + 16 * c + 4 * g;              // this lane's four consecutive k (MFMA step s of chunk c contracts k = kb + 16c + 4g + s)
                float a4[4] = {0.f, 0.f, 0.f, 0.f};
                if (ig < M && kq < NB * bi + NB && kq < M) {
                    if (al16) { const f32x4g v = *reinterpret_cast<const f32x4g*>(Lf + (size_t)ig * M + kq); a4[0] = v[0]; a4[1] = v[1]; a4[2] = v[2]; a4[3] = v[3]; }
                    else { for (int s_ = 0; s_ < 4; ++s_) if (kq + s_ < M) a4[s_] = Lf[(size_t)ig * M + kq + s_]; }
                }
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    const int k = kq + s_;
                    const bool in = k < NB * bi + NB && k < M;
                    av[4 * c + s_] = (in && k <= ig) ? (double)a4[s_] : 0.0;
                    double wv = 0.0;
                    if (in && jg < M && k >= jg) { const int i2 = M - 1 - jg, k2 = M - 1 - k; wv = Xs[boff(i2 >> 4, k2 >> 4) + (i2 & 15) * BLD + (k2 & 15)]; }
                    bv[4 * c + s_] = wv;
                }
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) if (kb + 16 * (u >> 2) < NB * bi + NB) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
        }
    }
    __syncthreads();                                         // every read of this block row of the old L is done (no other workgroup reads it)
    // acc[e] = L'[16bi + g + 4e][16bj + ri]: the tile to q_sqrt (float32) and to the workspace (float64), and its share of t = L'^T mbar
    double ts = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = NB * bi + g + 4 * e, j = NB * bj + ri;
        const bool in = bj <= bi && i < M && j < M && j <= i;
        const double v = in ? acc[e] : 0.0;
        if (i < M && j < M) Lf[(size_t)i * M + j] = (float)v;                            // (tiles above the diagonal: zero, as tril() leaves them)
        if (bj <= bi) Lp[boff(bi, bj) + (g + 4 * e) * BLD + ri] = v;
        if (in) ts = fma(v, -(double)a.dq_mu[(size_t)i * a.R + r], ts);                // mbar = -dq_mu
    }
    ts += __shfl_xor(ts, 16, 64);
    ts += __shfl_xor(ts, 32, 64);
    if (g == 0 && NB * bj + ri < Mp) tp[NB * bj + ri] = ts;                              // t_bi[j] = sum_{i in block row bi} L'[i][j] mbar[i]
}
// t = the block rows' shares summed in a fixed order, mu' = m - gamma L' t (the float64 L' of k_ng_rows); one workgroup per latent GP, a wave per
// block row: L'(bi, bj) T_bj on the float64 matrix cores with T_bj = [t_bj | 0 ..] -- fifteen sixteenths of the product are zeros, but a
// lane-per-column sum needs a six-step shuffle tree per ROW (sixteen dependent trees per wave: most of the 8.4 us this kernel took)
__global__ __launch_bounds__(512) void k_ng_vec(const NgRowsArgs a) {
    __shared__ double tm[8][BLK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = blockIdx.x, M = a.M, Mp = a.Mp, nbk = Mp / NB, ntri = nbk * (nbk + 1) / 2;
    const double* Lp = a.Lp64 + (size_t)r * ntri * BLK;
    const int ri = lane & 15, g = lane >> 4, bi = wave;
    // every global read of the kernel in flight before the first is used
    double av[8][4];
#pragma unroll
    for (int bj = 0; bj < 8; ++bj) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) av[bj][kk] = (bi < nbk && bj <= bi) ? Lp[boff(bi, bj) + ri * BLD + 4 * kk + g] : 0.0;
    }
    float mq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int i = NB * bi + g + 4 * e; mq[e] = (bi < nbk && ri == 0 && i < M) ? a.q_mu[(size_t)i * a.R + r] : 0.f; }
    for (int i = tid; i < 8 * BLK; i += 512) (&tm[0][0])[i] = 0.0;
    double v = 0.0;
    if (tid < Mp) {
        double sh[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) sh[b] = (b >= (tid >> 4) && b < nbk) ? a.tpart[((size_t)r * nbk + b) * Mp + tid] : 0.0;
#pragma unroll
        for (int b = 0; b < 8; ++b) v += sh[b];
    }
    __syncthreads();
    if (tid < Mp) tm[tid >> 4][(tid & 15) * BLD] = v;            // T_bj[k][0] = t[16 bj + k]
    __syncthreads();
    if (bi >= nbk) return;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int bj = 0; bj < 8; ++bj) {
        if (bj <= bi) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[bj][kk], tm[bj][(4 * kk + g) * BLD + ri], acc, 0, 0, 0);
        }
    }
    if (ri == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {                        // acc[e] = (L' t)[16 bi + g + 4e] in the lanes of column 0
            const int i = NB * bi + g + 4 * e;
            if (i < M) a.q_mu[(size_t)i * a.R + r] = (float)((double)mq[e] - a.gamma * acc[e]);
        }
    }
}

// host side: 0 if the shape is not covered (the caller then takes the multi-launch path)
int natgrad_small(float* q_mu, float* q_sqrt, const float* dq_mu, const float* dq_sqrt, int M, int R, double gamma, hipStream_t st, void* ws, size_t ws_bytes);

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


int natgrad_small(float* q_mu, float* q_sqrt, const float* dq_mu, const float* dq_sqrt, int M, int R, double gamma, hipStream_t st, void* ws, size_t ws_bytes) {
    const int Mp = round_up(M, NB);
    if (Mp > 128 || dbg_opt("IWVI_NATGRAD_UNFUSED")) return 0;
    const size_t lds = sizeof(double) * ((size_t)Mp + ws_layout(Mp).total + 4 * (size_t)Mp + 16 * 128);
    int rc;
    if ((rc = ensure_lds_attr((const void*)k_natgrad_small, lds)) != IWVI_OK) return rc;
    // step 1 (Qrev) on many CUs first, when the workspace holds the R block images
    const int nbk = Mp / NB, ntri = nbk * (nbk + 1) / 2;
    {   // the step spread over the chip (k_ng_chol, k_sb_inv, k_ng_rows): when the workspace holds its images
        const WsLayout w = ws_layout(Mp);
        const size_t n_q = (size_t)R * ntri * BLK, n_fac = (size_t)R * w.total, n_t = (size_t)R * nbk * Mp;
        const size_t need = sizeof(double) * (3 * n_q + n_fac + n_t) + 64;
        if (ws && need <= ws_bytes && R <= IWVI_MAX_LAYERS && !dbg_opt("IWVI_NG_STOP") && !dbg_opt("IWVI_NG_ONE_WG")) {
            double* qw = (double*)ws; double* fac = qw + n_q; double* X64 = fac + n_fac; double* Lp64 = X64 + n_q; double* tpart = Lp64 + n_q;
            hipLaunchKernelGGL(k_ng_qbuild4, dim3(ntri, R), dim3(256), 0, st, (const float*)q_sqrt, dq_sqrt, M, gamma, qw, ntri);
            if ((rc = check_launch("k_ng_qbuild")) != IWVI_OK) return rc;
            const size_t lds_c = sizeof(double) * ((size_t)Mp + w.total);
            if ((rc = ensure_lds_attr((const void*)k_ng_chol, lds_c)) != IWVI_OK) return rc;
            hipLaunchKernelGGL(k_ng_chol, dim3(R), dim3(1024), lds_c, st, (const double*)qw, fac, Mp);
            if ((rc = check_launch("k_ng_chol")) != IWVI_OK) return rc;
            SbInvAll q{};
            for (int r = 0; r < R; ++r) {
                SbInvOne& o = q.L[q.n++];
                o.ws = fac + (size_t)r * w.total; o.LsP = nullptr; o.nbk = nbk; o.Mp = Mp; o.M = Mp; o.first = 8 * r; o.X64 = X64 + (size_t)r * ntri * BLK;
            }
            const size_t lds_sb = sizeof(double) * (size_t)(8 + 8 + 28 + 4) * BLK;
            if ((rc = ensure_lds_attr((const void*)k_sb_inv, lds_sb)) != IWVI_OK) return rc;
            hipLaunchKernelGGL(k_sb_inv, dim3(8 * R), dim3(256), lds_sb, st, q);
            if ((rc = check_launch("k_sb_inv")) != IWVI_OK) return rc;
            NgRowsArgs a{q_mu, q_sqrt, dq_mu, X64, Lp64, tpart, M, Mp, R, gamma};
            const size_t lds_r = sizeof(double) * (size_t)ntri * BLK;
            if ((rc = ensure_lds_attr((const void*)k_ng_rows, lds_r)) != IWVI_OK) return rc;
            hipLaunchKernelGGL(k_ng_rows, dim3(nbk, R), dim3(512), lds_r, st, a);
            if ((rc = check_launch("k_ng_rows")) != IWVI_OK) return rc;
            hipLaunchKernelGGL(k_ng_vec, dim3(R), dim3(512), 0, st, a);
            rc = check_launch("k_ng_vec");
            return rc == IWVI_OK ? 2 : rc;                       // (2: the spread route; 1: one workgroup per latent GP)
        }
    }
    double* qws = nullptr;
    if (ws && (size_t)R * ntri * BLK * sizeof(double) <= ws_bytes && !dbg_opt("IWVI_NG_STOP")) {
        qws = (double*)ws;
        hipLaunchKernelGGL(k_ng_qbuild, dim3(ntri, R), dim3(64), 0, st, (const float*)q_sqrt, dq_sqrt, M, gamma, qws, ntri);
        if ((rc = check_launch("k_ng_qbuild")) != IWVI_OK) return rc;
    }
    hipLaunchKernelGGL(k_natgrad_small, dim3(R), dim3(1024), lds, st, q_mu, q_sqrt, dq_mu, dq_sqrt, M, R, gamma, dbg_opt("IWVI_NG_STOP"), (const double*)qws);
    rc = check_launch("k_natgrad_small");
    return rc == IWVI_OK ? 1 : rc;
}

}  // namespace iwvi

using namespace iwvi;

extern "C" size_t iwvi_natgrad_ws_bytes_ex(int M, int R) {
    size_t base = iwvi_natgrad_ws_bytes(M);
    if (base == 0 || R <= 0 || R > IWVI_MAX_R) return 0;
    const int Mp = round_up(M, NB);
    if (Mp <= 128 && R <= IWVI_MAX_LAYERS) {                     // what natgrad_small's spread route carves: three block images, the factor's workspace, the row shares
        const int nbk = Mp / NB, ntri = nbk * (nbk + 1) / 2;
        const WsLayout w = ws_layout(Mp);
        const size_t need = sizeof(double) * (3 * (size_t)R * ntri * BLK + (size_t)R * w.total + (size_t)R * nbk * Mp) + 64;
        if (need > base) base = (need + 255) & ~(size_t)255;
    }
    return base;
}

extern "C" size_t iwvi_gp_state_bytes(int M, int R) {
    if (M <= 0 || R <= 0) return 0;
    return state_layout(M, R).bytes;
}

extern "C" int iwvi_gp_state_offsets(int M, int R, size_t out[8]) {
    if (M <= 0 || R <= 0 || !out) { set_error("iwvi_gp_state_offsets: bad argument"); return IWVI_ERR_ARG; }
    StateLayout s = state_layout(M, R);
    out[0] = s.off_Lm; out[1] = s.off_Linv; out[2] = s.off_LsP; out[3] = s.off_LrTP;
    out[4] = s.off_QmuP; out[5] = s.off_ZtP; out[6] = s.off_cst; out[7] = s.off_kl;
    return IWVI_OK;
}

namespace iwvi {
// float64 stage-1 route: z~[m][d] = fl32(fl32(Z[m][d] / l[d]) - zc[d]) -- the centred, scaled inducing inputs exactly as k_precompute
// formed them for K_uu (precompute_dev.h: role 0; zc = the published centre, cst[32 + d]) -- as plain float32 [Mp][IWVI_MAX_D], padding 0.
// The forward's float64 Gram differences them against x~ directly (ZtP carries log2(e) z~, rounded once more: not the same numbers).
struct F64PrepOne { const float* Z; const float* ls; const float* cst; float* Zs; int M, Mp, D, pad_; };
struct F64PrepAll { F64PrepOne L[IWVI_MAX_LAYERS]; int n; };
__global__ __launch_bounds__(256) void k_f64_prep(const F64PrepAll a) {
    const F64PrepOne& L = a.L[blockIdx.x];
    for (int idx = threadIdx.x; idx < L.Mp * IWVI_MAX_D; idx += 256) {
        const int m = idx / IWVI_MAX_D, d = idx - m * IWVI_MAX_D;
        float v = 0.f;
        if (m < L.M && d < L.D) {
            const float zs = (float)((double)L.Z[(size_t)m * L.D + d] / (double)L.ls[d]);
            v = zs - L.cst[32 + d];
        }
        L.Zs[idx] = v;
    }
}
}  // namespace iwvi

extern "C" int iwvi_gp_precompute(const iwvi_gp_desc* layers, int n_layers, void* stream_) {
    return iwvi_model_precompute(layers, n_layers, nullptr, 0, stream_);
}

extern "C" int iwvi_model_precompute(const iwvi_gp_desc* layers, int n_layers, const iwvi_enc_desc* encs, int n_encs,
                                     void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!layers || n_layers <= 0) { set_error("iwvi_gp_precompute: no layers"); return IWVI_ERR_ARG; }
    if (n_encs < 0 || n_encs > PRE_MAX_ENC || (n_encs > 0 && !encs)) { set_error("iwvi_model_precompute: %d encoders (0..%d supported)", n_encs, PRE_MAX_ENC); return IWVI_ERR_ARG; }
    for (int base = 0; base < n_layers; base += IWVI_MAX_LAYERS) {
        PreArgs a{};
        a.n = n_layers - base < IWVI_MAX_LAYERS ? n_layers - base : IWVI_MAX_LAYERS;
        a.stop_after = dbg_opt("IWVI_DEBUG_STOP");
        a.stamps = g_pre_stamps;
        a.stamp_p = (g_pre_stamps && dbg_opt("IWVI_PRE_STAMP_P")) ? dbg_opt("IWVI_PRE_STAMP_P") : 1;
        size_t lds = 1024 * sizeof(double);
        int max_roles = 0;
        for (int l = 0; l < a.n; ++l) {
            const iwvi_gp_desc& d = layers[base + l];
            PreLayer& L = a.L[l];
            { const int rc = fill_pre_layer(d, base + l, L); if (rc != IWVI_OK) return rc; }
            if ((d.flags & IWVI_GP_REUSE_FACTOR) && (d.flags & IWVI_GP_FACTOR_ONLY)) { set_error("iwvi_gp_precompute: layer %d asks for IWVI_GP_REUSE_FACTOR and IWVI_GP_FACTOR_ONLY", base + l); return IWVI_ERR_ARG; }
            if (d.flags & IWVI_GP_F64_STAGE1) L.flags |= IWVI_GP_WANT_LM;      // the float64 route multiplies by the dense Lm^-1 (k_linv below)
            if (L.nbk >= 16 && !(d.flags & (IWVI_GP_WANT_DENSE | IWVI_GP_REUSE_FACTOR)) && !dbg_opt("IWVI_PRE_SB_INLINE")) L.flags |= IWVI_GP_SB_EXT_;   // super-block inverses by k_sb_inv
            size_t la = factor_lds_bytes(L.Mp);
            if (la > lds) lds = la;
            if (d.R + 1 > max_roles) max_roles = d.R + 1;
        }
        int enc_blocks = 0;
        if (base == 0) {                                   // encoders ride with the first batch of layers
            for (int e = 0; e < n_encs; ++e) {
                const iwvi_enc_desc& d = encs[e];
                if (!d.XY || !d.out || !d.enc_W || !d.dims || d.rows <= 0 || d.n_enc <= 0 || d.n_enc > IWVI_MAX_ENC || d.latent_dim <= 0) {
                    set_error("iwvi_model_precompute: bad encoder descriptor %d", e); return IWVI_ERR_ARG;
                }
                if (d.dims[d.n_enc] != 2 * d.latent_dim) { set_error("iwvi_model_precompute: encoder output %d != 2*latent_dim %d", d.dims[d.n_enc], 2 * d.latent_dim); return IWVI_ERR_ARG; }
                PreEnc& E = a.E[e];
                size_t w = 0;
                for (int k = 0; k <= d.n_enc; ++k) {
                    if (d.dims[k] <= 0 || d.dims[k] > 64) { set_error("iwvi_model_precompute: encoder width %d out of range (1..64)", d.dims[k]); return IWVI_ERR_ARG; }
                    E.dims[k] = d.dims[k];
                }
                for (int k = 0; k < d.n_enc; ++k) {
                    if (!d.enc_W[k]) { set_error("iwvi_model_precompute: null encoder weight %d", k); return IWVI_ERR_ARG; }
                    E.W[k] = d.enc_W[k]; E.b[k] = d.enc_b ? d.enc_b[k] : nullptr;
                    w += (size_t)d.dims[k] * d.dims[k + 1] + d.dims[k + 1];
                }
                E.XY = d.XY; E.out = d.out; E.rows = d.rows; E.n_enc = d.n_enc; E.Lw = d.latent_dim; E.act = d.act;
                if (d.act < IWVI_ACT_TANH || d.act > IWVI_ACT_IDENTITY) { set_error("iwvi_model_precompute: unknown activation %d", d.act); return IWVI_ERR_UNSUPPORTED; }
                E.sample_X = d.sample_X; E.sample_kl = d.sample_kl; E.sample_z = d.sample_z;
                if (d.sample_X) {
                    if (!d.X || !d.sample_kl || d.Dx <= 0 || d.Dx + d.latent_dim > IWVI_MAX_D || d.K <= 0) { set_error("iwvi_model_precompute: bad sampling tail of encoder %d", e); return IWVI_ERR_ARG; }
                    E.X = d.X; E.Dx = d.Dx; E.K = d.K; E.sampled_kl = d.sampled_kl; E.layer_index = d.layer_index;
                    E.seed = d.seed; E.rng_state = (const unsigned long long*)d.rng_state;
                }
                E.blk0 = enc_blocks; E.nblk = (int)((d.rows + ENC_ROWS - 1) / ENC_ROWS);
                enc_blocks += E.nblk;
                const size_t need = (w + 4 + 2 * (size_t)ENC_ROWS * 65) * sizeof(float);   // weights + two activation buffers
                if (need > lds) lds = need;
            }
            a.n_enc = n_encs;
        }
        int rc;
        if ((rc = ensure_lds_attr((const void*)k_precompute, lds)) != IWVI_OK) return rc;
        hipLaunchKernelGGL(k_precompute, dim3(a.n + enc_blocks, max_roles), dim3(1024), lds, stream, a);
        if ((rc = check_launch("k_precompute")) != IWVI_OK) return rc;
        {   // layers with M > 240: the inverses of the diagonal super-blocks, one workgroup per 16-column block of each
            SbInvAll q{};
            int grid = 0;
            size_t lds_sb = 0;
            for (int l = 0; l < a.n; ++l) {
                const PreLayer& L = a.L[l];
                if (!(L.flags & IWVI_GP_SB_EXT_)) continue;
                SbInvOne& o = q.L[q.n++];
                o.ws = L.ws; o.LsP = L.LsP; o.nbk = L.nbk; o.Mp = L.Mp; o.M = L.M; o.first = grid;
                grid += 8 * ((L.nbk + 7) / 8);
                lds_sb = sizeof(double) * (size_t)(8 + 8 + 28 + 4) * BLK;
            }
            if (q.n > 0) {
                if ((rc = ensure_lds_attr((const void*)k_sb_inv, lds_sb)) != IWVI_OK) return rc;
                hipLaunchKernelGGL(k_sb_inv, dim3(grid), dim3(256), lds_sb, stream, q);
                if ((rc = check_launch("k_sb_inv")) != IWVI_OK) return rc;
            }
        }
        {   // layers with M > 240: the split-f16 operands of the super-block solve's dense part
            Ls16All q{};
            int grid = 0;
            for (int l = 0; l < a.n; ++l) {
                const PreLayer& L = a.L[l];
                if (L.nbk < 16 || (L.flags & IWVI_GP_REUSE_FACTOR)) continue;
                const StateLayout sl = state_layout(L.M, L.R);
                Ls16One& o = q.L[q.n++];
                o.blk = L.ws + ws_layout(L.Mp).blk;
                o.Lm = (L.flags & (IWVI_GP_WANT_DENSE | IWVI_GP_WANT_LM)) ? L.Lm : nullptr;
                o.dst = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(L.Lm) - sl.off_Lm + sl.off_Ls16);
                o.LsP = L.LsP;
                o.variance = L.variance; o.variance_dev = L.variance_dev;
                o.nbk = L.nbk; o.Mp = L.Mp; o.M = L.M; o.first = grid;
                grid += (sb16_slabs(L.nbk) + sb16_tri_blocks(L.nbk) + 3) / 4;
            }
            if (q.n > 0) {
                hipLaunchKernelGGL(k_pack_ls16, dim3(grid), dim3(256), 0, stream, q);
                if ((rc = check_launch("k_pack_ls16")) != IWVI_OK) return rc;
            }
        }
        {   // layers prepared for the float64 stage-1 route (IWVI_GP_F64_STAGE1): Lm^-1 in float64 on many CUs, the plain z~ the factor saw
            iwvi_gp_desc f64l[IWVI_MAX_LAYERS];
            F64PrepAll fp{};
            int nf = 0;
            for (int l = 0; l < a.n; ++l) {
                const iwvi_gp_desc& d = layers[base + l];
                if (!(d.flags & IWVI_GP_F64_STAGE1) || (d.flags & IWVI_GP_REUSE_FACTOR)) continue;
                const PreLayer& L = a.L[l];
                const StateLayout sl = state_layout(L.M, L.R);
                f64l[nf] = d;
                F64PrepOne& o = fp.L[nf++];
                o.Z = L.Z; o.ls = L.ls; o.cst = L.cst; o.Zs = reinterpret_cast<float*>(reinterpret_cast<char*>(L.Lm) - sl.off_Lm + sl.off_Zs);
                o.M = L.M; o.Mp = L.Mp; o.D = L.D;
            }
            if (nf > 0) {
                fp.n = nf;
                if ((rc = iwvi_gp_dense_inverse(f64l, nf, stream_)) != IWVI_OK) return rc;
                hipLaunchKernelGGL(k_f64_prep, dim3(nf), dim3(256), 0, stream, fp);
                if ((rc = check_launch("k_f64_prep")) != IWVI_OK) return rc;
            }
        }
    }
    return IWVI_OK;
}

// ---- Lm^-1 from the dense Lm, one workgroup (4 waves) per 16-column block J of the inverse: X_J = D_J^-1, then block row by block row
// X_bi = -D_bi^-1 sum_{J <= bk < bi} L(bi, bk) X_bk (float64 MFMA products, the sum dealt to the four waves).  nbk workgroups per layer
// instead of the factorising workgroup's recursive doubling (17 of the 54 us of a dense factorisation at M = 128 on ONE CU, during which
// that CU is lost to the layer kernel): 5-6 us at M = 128.
namespace iwvi {
struct LinvOne { const double* Lm; double* Linv; int Mp, nbk, first; int pad_; };
struct LinvAll { LinvOne L[IWVI_MAX_STACK]; int n; };
__global__ __launch_bounds__(256) void k_linv(const LinvAll a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char linv_smem[];
    int li = 0;
    while (li + 1 < a.n && (int)blockIdx.x >= a.L[li + 1].first) ++li;
    const LinvOne& L = a.L[li];
    const int J = (int)blockIdx.x - L.first, nbk = L.nbk, Mp = L.Mp, nb = nbk - J;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* X = reinterpret_cast<double*>(linv_smem);            // [nb][BLK]: the finished blocks of this block column
    double* Dv = X + (size_t)nb * BLK;                          // [nb][BLK]: the diagonal blocks, then their inverses
    double* part = Dv + (size_t)nb * BLK;                       // [4][BLK]: the waves' partial sums
    double* rinv = part + 4 * BLK;                              // [nb][16]: reciprocal pivots
    // nbk <= 8: the lower blocks L(J + t, J + u), u <= t, go to LDS in ONE round trip (all loads of a thread issued before its first store:
    // a load-store loop pays a global round trip per block, 30 us for the 36 blocks of J = 0): the diagonal ones to Dv (inverted in place
    // below), the others to the packed triangle Ls[t (t - 1) / 2 + u] -- read from the dense factor inside the loop, every step would
    // wait for a round trip of its own
    double* Ls = rinv + (size_t)nb * 16;
    const bool staged = nbk <= 8;
    {
        const int rr = tid >> 4, cc = tid & 15;
        if (staged) {
            constexpr int MAXB = 36;
            const int ntot = nb * (nb + 1) / 2;
            double v[MAXB];
#pragma unroll
            for (int q = 0; q < MAXB; ++q) {
                int t = 0;
                while ((t + 1) * (t + 2) / 2 <= q) ++t;              // (compile-time: q is an unrolled constant)
                const int u = q - t * (t + 1) / 2;
                v[q] = (q < ntot) ? L.Lm[(size_t)(16 * (J + t) + rr) * Mp + 16 * (J + u) + cc] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < MAXB; ++q) {
                int t = 0;
                while ((t + 1) * (t + 2) / 2 <= q) ++t;
                const int u = q - t * (t + 1) / 2;
                if (q < ntot) {
                    if (u == t) { Dv[(size_t)t * BLK + rr * BLD + cc] = v[q]; if (rr == cc) rinv[t * 16 + rr] = 1.0 / v[q]; }
                    else Ls[(size_t)(t * (t - 1) / 2 + u) * BLK + rr * BLD + cc] = v[q];
                }
            }
        } else {
            for (int b0 = 0; b0 < nb; b0 += 8) {                     // diagonal blocks only, eight round trips in flight
                double v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { const int bi = J + (b0 + e < nb ? b0 + e : nb - 1); v[e] = L.Lm[(size_t)(16 * bi + rr) * Mp + 16 * bi + cc]; }
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (b0 + e < nb) { Dv[(size_t)(b0 + e) * BLK + rr * BLD + cc] = v[e]; if (rr == cc) rinv[(b0 + e) * 16 + rr] = 1.0 / v[e]; }
            }
        }
    }
    __syncthreads();
    for (int b = wave; b < nb; b += 4) diag_inverse(Dv + (size_t)b * BLK, rinv + b * 16, b == 0 ? X : Dv + (size_t)b * BLK, lane);
    __syncthreads();
    const int r = lane & 15, g = lane >> 4;
    for (int t = 1; t < nb; ++t) {
        const int bi = J + t;
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int bk = J + wave; bk < bi; bk += 4) {             // acc += L(bi, bk) X_bk
            const double* B = X + (size_t)(bk - J) * BLK;
            if (staged) blk_mma<false>(acc, Ls + (size_t)(t * (t - 1) / 2 + (bk - J)) * BLK, B, lane, 1.0);
            else {                                              // (the A operand straight from the dense factor)
                const double* Ag = L.Lm + (size_t)(16 * bi + r) * Mp + 16 * bk;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ag[4 * kk + g], B[(4 * kk + g) * BLD + r], acc, 0, 0, 0);
            }
        }
        blk_store(part + wave * BLK, acc, lane);
        __syncthreads();
        {   // the four partial sums, one entry per thread
            const int rr = tid >> 4, cc = tid & 15, o = rr * BLD + cc;
            part[o] = (part[o] + part[BLK + o]) + (part[2 * BLK + o] + part[3 * BLK + o]);
        }
        __syncthreads();
        if (wave == 0) {
            f64x4 x = {0.0, 0.0, 0.0, 0.0};
            blk_mma<false>(x, Dv + (size_t)t * BLK, part, lane, -1.0);
            blk_store(X + (size_t)t * BLK, x, lane);
        }
        __syncthreads();
    }
    // block column J of the dense inverse (zeros above the diagonal block)
    for (int idx = tid; idx < nbk * 256; idx += 256) {
        const int bi = idx >> 8, rr = (idx >> 4) & 15, cc = idx & 15;
        double v = 0.0;
        if (bi >= J) v = X[(size_t)(bi - J) * BLK + rr * BLD + cc];
        if (bi == J && cc > rr) v = 0.0;
        L.Linv[(size_t)(16 * bi + rr) * Mp + 16 * J + cc] = v;
    }
}
}  // namespace iwvi

extern "C" int iwvi_gp_dense_inverse(const iwvi_gp_desc* layers, int n_layers, void* stream_) {
    using namespace iwvi;
    if (!layers || n_layers <= 0 || n_layers > IWVI_MAX_STACK) { set_error("iwvi_gp_dense_inverse: bad argument"); return IWVI_ERR_ARG; }
    LinvAll a{};
    a.n = n_layers;
    int grid = 0;
    for (int i = 0; i < n_layers; ++i) {
        const iwvi_gp_desc& d = layers[i];
        if (!d.state || d.M <= 0 || d.M > IWVI_MAX_M || d.R <= 0 || d.R > IWVI_MAX_R) { set_error("iwvi_gp_dense_inverse: layer %d: null state or size out of range", i); return IWVI_ERR_ARG; }
        const StateLayout s = state_layout(d.M, d.R);
        a.L[i].Lm = (const double*)((const char*)d.state + s.off_Lm);
        a.L[i].Linv = (double*)((char*)d.state + s.off_Linv);
        a.L[i].Mp = s.Mp; a.L[i].nbk = s.nbk; a.L[i].first = grid;
        grid += s.nbk;
    }
    size_t lds = 0;
    for (int i = 0; i < n_layers; ++i) {
        const int nb = a.L[i].nbk;
        const size_t need = sizeof(double) * ((size_t)(2 * nb + 4 + (nb <= 8 ? nb * (nb - 1) / 2 : 0)) * BLK + (size_t)nb * 16);
        if (need > lds) lds = need;
    }
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_linv), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("iwvi_gp_dense_inverse: %zu B of LDS: %s", lds, hipGetErrorString(e)); return IWVI_ERR_LAUNCH; }
    }
    hipLaunchKernelGGL(k_linv, dim3(grid), dim3(256), lds, (hipStream_t)stream_, a);
    return check_launch("iwvi_gp_dense_inverse");
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
    return ws_layout(round_up(M, NB)).total * sizeof(double);
}

extern "C" int iwvi_chol_factor(const double* A, double* Lout, int M, void* ws, void* stream_) {
    if (!A || !Lout || !ws || M <= 0) { set_error("iwvi_chol_factor: bad argument"); return IWVI_ERR_ARG; }
    if (M > 1024) { set_error("iwvi_chol_factor: M=%d too large (max 1024)", M); return IWVI_ERR_ARG; }
    const int Mp = round_up(M, NB);
    size_t lds = sizeof(double) * (size_t)Mp;
    int rc;
    if ((rc = ensure_lds_attr((const void*)k_chol_only, lds)) != IWVI_OK) return rc;
    hipLaunchKernelGGL(k_chol_only, dim3(1), dim3(1024), lds, (hipStream_t)stream_, A, Lout, M, Mp, (double*)ws);
    return check_launch("k_chol_only");
}

extern "C" int iwvi_gauss_kl(const float* q_mu, const float* q_sqrt, int M, int R, double* kl, void* stream_) {
    if (!q_mu || !q_sqrt || !kl || M <= 0 || R <= 0) { set_error("iwvi_gauss_kl: bad argument"); return IWVI_ERR_ARG; }
    hipLaunchKernelGGL(k_gauss_kl, dim3(1), dim3(256), 256 * sizeof(double), (hipStream_t)stream_, q_mu, q_sqrt, M, R, kl);
    return check_launch("k_gauss_kl");
}

/* diagnostic: 16 words per layer of phase stamps (100 MHz wall clock) written by the factorisation role */
extern "C" void iwvi_debug_set_pre_stamps(void* buf) { iwvi::g_pre_stamps = (unsigned long long*)buf; }

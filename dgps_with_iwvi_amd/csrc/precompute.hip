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
#include <cstdlib>

namespace iwvi {

constexpr int NB = 16;            // block size
constexpr int BLD = NB + 1;       // padded row stride of a block (doubles)
constexpr int BLK = NB * BLD;     // doubles per block
constexpr int ZLD = 33;           // row stride of the LDS copy of Zs (floats)

struct PreLayer {
    const float* Z; const float* ls; const float* q_mu; const float* q_sqrt;
    double* Lm; double* Linv; float* LsP; float* LrTP; float* QmuP; float* ZtP; float* cst; double* kl;
    double* ws;
    unsigned short* LrT16; unsigned short* Qmu16;   // split-f16 images (iwvi_common.h: s16_*)
    double jitter; float variance; const float* variance_dev;
    int M, D, R, Mp, nbk, nrb, kern_type, flags;
};
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
#define PRE_STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)

// exp(-x) for x >= 0 in float64, ~2e-16 relative: n = rint(-x log2 e), t = -x - n ln2 (two-term), degree-12 Taylor
// on |t| <= ln2/2 (remainder < 3e-17), scaled by 2^n with v_ldexp_f64.  About 22 fp64 instructions
// against ~100 for the library exp; the Gram is 8k of these on one CU, on the critical path of every step.
__device__ __forceinline__ double fma_c(double p, double t, double c) {      // p * t + c as ONE v_fma_f64 (the compiler's choice for a
    double r;                                                                 // Horner step with the coefficient in a register is
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(t), "v"(c));     // v_mov_b64 + v_fmac_f64: two float64-rate instructions)
    return r;
}
__device__ __forceinline__ double exp_neg(double x) {
    const double y = -fmin(x, 1000.0);              // (exp(-1000) = 0 through v_ldexp_f64's underflow: no separate select)
    const double n = rint(y * 1.4426950408889634074);
    double t = fma(-n, 6.93147180369123816490e-01, y);
    t = fma(-n, 1.90821492927058770002e-10, t);
    double p = 2.08767569878680989792e-09;          // 1/12!
    p = fma_c(p, t, 2.50521083854417187751e-08);    // 1/11!
    p = fma_c(p, t, 2.75573192239858906526e-07);    // 1/10!
    p = fma_c(p, t, 2.75573192239858906526e-06);    // 1/9!
    p = fma_c(p, t, 2.48015873015873015873e-05);    // 1/8!
    p = fma_c(p, t, 1.98412698412698412698e-04);    // 1/7!
    p = fma_c(p, t, 1.38888888888888888889e-03);    // 1/6!
    p = fma_c(p, t, 8.33333333333333333333e-03);    // 1/5!
    p = fma_c(p, t, 4.16666666666666666667e-02);    // 1/4!
    p = fma_c(p, t, 1.66666666666666666667e-01);    // 1/3!
    p = fma(p, t, 0.5);
    p = fma(p, t, 1.0);
    p = fma(p, t, 1.0);
    return ldexp(p, (int)n);
}

__device__ __forceinline__ double kern_value(double r2, int type, double var) {
    if (type == IWVI_KERN_MATERN52) {
        const double s5 = 2.23606797749978969641;
        double r = sqrt(r2 + 1e-12);
        return var * (1.0 + s5 * r + (5.0 / 3.0) * r * r) * exp_neg(s5 * r);
    }
    return var * exp_neg(0.5 * r2);
}

__host__ __device__ __forceinline__ int boff(int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * BLK; }

// workspace carve (doubles): lower-triangle blocks | diagonal-block inverses | T scratch of the inversion
struct WsLayout { int nbk; size_t blk, dinv, tbuf, total; };
__host__ __device__ static inline WsLayout ws_layout(int Mp) {
    WsLayout w;
    w.nbk = Mp / NB;
    w.blk = 0;
    w.dinv = (size_t)(w.nbk * (w.nbk + 1) / 2) * BLK;
    w.tbuf = w.dinv + (size_t)w.nbk * BLK;
    w.total = w.tbuf + (size_t)((w.nbk * w.nbk + 3) / 4) * BLK;
    return w;
}

__device__ __forceinline__ double readlane_d(double v, int src) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// Cholesky of one 16x16 diagonal block, the triangular solve of up to two 16-row blocks below it AND the inverse
// of the factor, by ONE wave: lane l owns one row in registers -- lanes 0-15 the diagonal block's rows, lanes
// 16-47 the rows of blocks p+1, p+2 of the column, lanes 48-63 the rows of an identity block.  The pivot and the
// freshly scaled column are broadcast from the diagonal block's rows, and the very same
// per-column operations (scale by 1/l_jj, subtract l_ij l_kj) that factor the diagonal block perform
// x L_pp^T = a on every other row -- at no extra instruction.  For the identity rows the solution is L_pp^-T,
// which is all the inverse the rest of the pipeline needs.  16 steps, no LDS traffic, no barrier.
// win = number of blocks below carried along (<= 2).  Writes the factor back (the strict upper part of the diagonal
// block is left as computed: nothing reads it), X = L_pp^-T to xT (row i, column k at [i*BLD + k]) and 1/L[j][j] to rinv[0..15].
// The broadcasts are DPP operands, not instructions: every 16-lane group also carries the diagonal block's rows
// (dg), so "l_kj" for any lane is lane k of its own row of 16 -- v_fmac_f64 with row_newbcast:k reads it in
// place.  Per (column j, later column k): two v_fmac_f64_dpp (own row, diagonal-block copy) instead of two
// v_readlane + one fma, and no SGPR traffic; the pass is bound by the instruction count of this one wave.
template <int K>
__device__ __forceinline__ void fmac_neg_bcast(double& d, double bsrc, double s1) {
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(bsrc), "v"(s1), "n"(K));
}
template <int K>
__device__ __forceinline__ double bcast_row(double v) {       // lane K of each row of 16; s_nop: the source may have just been written
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
    return r;
}
template <int J, int K>
struct WindowUpd {
    static __device__ __forceinline__ void run(double (&a)[NB], double (&dg)[NB], double ld, double lij) {
        fmac_neg_bcast<K>(a[K], ld, lij);                        // a[k]  -= l_kj * l_ij
        fmac_neg_bcast<K>(dg[K], ld, ld);                        // diagonal-block copy of the same update
        WindowUpd<J, K + 1>::run(a, dg, ld, lij);
    }
};
template <int J>
struct WindowUpd<J, NB> { static __device__ __forceinline__ void run(double (&)[NB], double (&)[NB], double, double) {} };
template <int J>
struct WindowCol {
    static __device__ __forceinline__ void run(double (&a)[NB], double (&dg)[NB], double& rkeep, int i) {
        const double ajj = bcast_row<J>(dg[J]);
        // 1/sqrt(pivot): hardware seed (v_rsq_f64, ~2^-26) + one Newton step (-> ~1e-15); the pivot is positive (jitter)
        const double y0 = __builtin_amdgcn_rsq(ajj);
        const double e = fma(-(ajj * y0), y0, 1.0);
        const double r = fma(0.5 * y0, e, y0);
        const double lij = a[J] * r;
        double ld = dg[J] * r;
        a[J] = lij;
        rkeep = (i == J) ? r : rkeep;
        asm volatile("s_nop 1" : "+v"(ld));                       // ld: VALU write -> DPP read needs two wait states (tied to ld)
        dg[J] = ld;
        WindowUpd<J, J + 1>::run(a, dg, ld, lij);
        WindowCol<J + 1>::run(a, dg, rkeep, i);
    }
};
template <>
struct WindowCol<NB> { static __device__ __forceinline__ void run(double (&)[NB], double (&)[NB], double&, int) {} };

__device__ __forceinline__ void diag_factor_window(double* blk, int p, int win, double* xT, double* rinv, int lane, unsigned long long* st = nullptr) {
    const int i = lane & 15, lb = lane >> 4;
    const bool ident = lb == 3, live = lb <= win;
    double* rowp = ident ? xT + i * BLD : blk + boff(p + (live ? lb : 0), p) + i * BLD;
    const double* drow = blk + boff(p, p) + i * BLD;            // row i of the diagonal block: a copy in every group
    double a[NB], dg[NB];
    int io = i;                                      // opaque copy: keeps the 16 identity-row constants from being hoisted out
    asm volatile("" : "+v"(io));                     // of the caller's column loop, where they would live in scratch memory
#pragma clang loop unroll(full)
    for (int k = 0; k < NB; ++k) { a[k] = rowp[k]; dg[k] = drow[k]; }   // all loads first, unconditionally (xT is valid memory)
#pragma clang loop unroll(full)
    for (int k = 0; k < NB; ++k) a[k] = ident ? (k == io ? 1.0 : 0.0) : a[k];
    if (st && lane == 0) { asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory"); st[14] = wall_clock64(); }
    double rkeep = 0.0;                              // lane j keeps 1/l_jj: one store after the pass
    WindowCol<0>::run(a, dg, rkeep, i);
    if (st && lane == 0) st[15] = wall_clock64();
    if (lane < NB) rinv[lane] = rkeep;
    if (live || ident) {
#pragma clang loop unroll(full)
        for (int k = 0; k < NB; ++k) rowp[k] = a[k];              // (the diagonal block's strict upper part is never read)
    }
}

// inverse of a lower-triangular 16x16 block by ONE wave: lane c owns column c of X; L[r][k] is read
// from the row-per-lane register copy by v_readlane (wave-uniform scalar operand).
__device__ __forceinline__ void diag_inverse(const double* D, const double* rinv, double* X, int lane) {
    const int i = lane & 15;
    double a[NB];
#pragma clang loop unroll(full)
    for (int k = 0; k < NB; ++k) a[k] = D[i * BLD + k];
    double x[NB];
#pragma clang loop unroll(full)
    for (int r = 0; r < NB; ++r) {
        double s = (r == i) ? 1.0 : 0.0;
#pragma clang loop unroll(full)
        for (int k = 0; k < r; ++k) s = fma(-readlane_d(a[k], r), x[k], s);
        x[r] = s * rinv[r];
    }
    if (lane < NB) {
#pragma clang loop unroll(full)
        for (int r = 0; r < NB; ++r) X[r * BLD + i] = (r >= i) ? x[r] : 0.0;
    }
}

// one wave: acc(16x16) += sign * A * B^T (NT) or sign * A * B (NN) on v_mfma_f64_16x16x4_f64.
// Operands: lane l feeds A[l&15][4kk + (l>>4)] and B[4kk + (l>>4)][l&15]; the accumulator register e of
// lane l is C[(l>>4) + 4e][l&15] (f64 C/D map, cdna guide section 3).  Per 16-deep product a lane reads
// 8 doubles from LDS instead of 80 for a VALU formulation, which was LDS-bandwidth bound.
using f64x4 = __attribute__((ext_vector_type(4))) double;

template <bool NT>
__device__ __forceinline__ void blk_mma(f64x4& acc, const double* A, const double* B, int lane, double sign) {
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const double a = sign * A[r * BLD + 4 * kk + g];
        const double b = NT ? B[r * BLD + 4 * kk + g] : B[(4 * kk + g) * BLD + r];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
}
__device__ __forceinline__ f64x4 blk_load(const double* C, int lane) {
    const int c = lane & 15, g = lane >> 4;
    f64x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = C[(g + 4 * e) * BLD + c];
    return v;
}
__device__ __forceinline__ void blk_store(double* C, const f64x4& v, int lane) {
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) C[(g + 4 * e) * BLD + c] = v[e];
}

// Blocked LEFT-looking Cholesky on block storage, with the matrix generated column by column.  rinv: [16*nbk]
// reciprocal pivots.  The 16-column diagonal pass is a serial, instruction-bound chain of ONE wave (diag_factor_window:
// factor, the next two block rows and L_pp^-T in one pass), so everything else is arranged to run beside it:
//   up front   gen(0), gen(1): block columns 0 and 1 of the matrix (every wave)
//   step p, A  wave 0: the diagonal pass of column p            | the other waves, one block each:
//                                                               |   column p+1 catches up with columns k < p
//                                                               |   (left-looking: C(i,p+1) -= L(i,k) L(p+1,k)^T),
//                                                               |   gen(p+2), and post(p-1) (column p-1 is final);
//                                                               |   beside the last pass: tail()
//   step p, B  one wave per block row i > p: L(i,p) = A(i,p) L_pp^-T for the rows beyond the pass's window, then
//              C(i,p+1) -= L(i,p) L(p+1,p)^T  -- after which column p+1 is ready for its diagonal pass.
// Two barriers per step; no trailing update ever sits on the critical path.
template <class GEN, class POST, class TAIL>
__device__ __forceinline__ void chol_blocks(double* blk, int nbk, double* rinv, double* xT, int tid, int nthreads, GEN gen, POST post, TAIL tail,
                                            unsigned long long* stamps = nullptr, int stamp_p = 1) {
    const int lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6;
    gen(0, nbk < 2 ? 1 : 2, wave, nw);
    __syncthreads();
    PRE_STAMP(7);
    for (int p = 0; p < nbk; ++p) {
        const int m = nbk - 1 - p;                       // block rows below the diagonal block
        const int win = m < 2 ? m : 2;                   // of which the factoring wave carries this many
        if (p == stamp_p) PRE_STAMP(10);
        if (wave == 0) {
            __builtin_amdgcn_s_setprio(3);               // the serial pass is the critical path: its LDS traffic goes first
            diag_factor_window(blk, p, win, xT + (size_t)p * BLK, rinv + NB * p, lane, (stamps && p == stamp_p) ? stamps + (size_t)blockIdx.x * 16 : nullptr);
            __builtin_amdgcn_s_setprio(0);
            if (p == stamp_p) PRE_STAMP(11);
        } else if ((wave & 3) != 0 || nw < 8) {
            // the workers: every wave that does not share wave 0's SIMD (waves 4, 8, .. would slow the serial pass down)
            const int w = (nw < 8) ? wave - 1 : wave - 1 - (wave >> 2), nwo = (nw < 8) ? nw - 1 : nw - (nw >> 2);
            // column p+1 catches up with the factored columns k < p
            if (p > 0) {
                for (int b = w; b < m; b += nwo) {
                    const int bi = p + 1 + b;
                    double* C = blk + boff(bi, p + 1);
                    f64x4 acc = blk_load(C, lane);
                    for (int k = 0; k < p; ++k) blk_mma<true>(acc, blk + boff(bi, k), blk + boff(p + 1, k), lane, -1.0);
                    blk_store(C, acc, lane);
                }
            }
            int wg = w;                                  // generation starts with the workers the catch-up left idle
            if (p > 0 && m < nwo) { wg = w - m; if (wg < 0) wg += nwo; }
            if (p + 2 < nbk) gen(p + 2, p + 3, wg, nwo);
            // packing column p-1: dealt to the workers that do NOT generate in this step, the idle ones first (rotated index wg: [0, ng)
            // generate, then the idle waves, the catch-up waves last) -- generating a block costs ten times a catch-up product, and the
            // barrier of an early pass waits for the generating waves
            if (p > 0) {
                const int ng = (p + 2 < nbk) ? nbk - p - 2 : 0;
                const int rel = ng < nwo ? wg - ng : w, nrel = ng < nwo ? nwo - ng : nwo;
                if (rel >= 0) for (int it = rel * 64 + lane; it < (nbk - p + 1) * 64; it += nrel * 64) post(p - 1, it >> 6, it & 63);
            }
            if (p == nbk - 1) tail(w * 64 + lane, nwo * 64);     // work nobody waits for, beside the last (otherwise idle) pass
        }
        __syncthreads();
        if (p == stamp_p) PRE_STAMP(12);
        // block rows below: finish column p (rows beyond the window) and bring column p+1 up to date with it
        for (int b = wave; b < m; b += nw) {
            const int bi = p + 1 + b;
            double* A = blk + boff(bi, p);
            if (b >= win) {
                f64x4 acc = {0.0, 0.0, 0.0, 0.0};
                blk_mma<false>(acc, A, xT + (size_t)p * BLK, lane, 1.0);
                blk_store(A, acc, lane);
            }
            double* C = blk + boff(bi, p + 1);
            f64x4 acc = blk_load(C, lane);
            blk_mma<true>(acc, A, blk + boff(p + 1, p), lane, -1.0);
            blk_store(C, acc, lane);
        }
        if (m > 0) __syncthreads();
        if (p == stamp_p) PRE_STAMP(13);
    }
    if (tid < 64) post(nbk - 1, 0, tid);                 // the last column: its diagonal block
    __syncthreads();
}
struct NoGen { __device__ void operator()(int, int, int, int) const {} };
struct NoTail { __device__ void operator()(int, int) const {} };
struct NoPost { __device__ void operator()(int, int, int) const {} };   // (column, block of it, lane)

// X = L^-1 in place: off-diagonal blocks of blk become blocks of X, diagonal blocks of X live in dinv.
// s_lo .. s_hi: the doubling steps to run (group sizes 2 s_lo .. s_hi); s_lo == 1 also inverts the diagonal blocks.  Stopping at
// s_hi = 8 leaves the inverses of the 128 x 128 diagonal SUPER-blocks (the blocks outside them still hold L).
__device__ __forceinline__ void invert_blocks(double* blk, double* dinv, double* tbuf, const double* rinv, int nbk,
                              int tid, int nthreads, int s_lo = 1, int s_hi = 1 << 30) {
    const int lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6;
    if (s_lo == 1) {
        for (int b = wave; b < nbk; b += nw) diag_inverse(blk + boff(b, b), rinv + NB * b, dinv + (size_t)b * BLK, lane);
        __syncthreads();
    }
    for (int s = s_lo; s < nbk && s < s_hi; s *= 2) {
        // stage 1: T_ij = sum_{k=j..aend-1} L_ik X_kj   (i in the B half, j in the A half of a 2s group)
        for (int o = wave; o < nbk * nbk; o += nw) {
            const int i = o / nbk, j = o - i * nbk;
            if (i / (2 * s) != j / (2 * s) || (i % (2 * s)) < s || (j % (2 * s)) >= s) continue;
            const int g = i / (2 * s), a0 = g * 2 * s, aend = a0 + s, b0 = aend;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            for (int k = j; k < aend; ++k)
                blk_mma<false>(acc, blk + boff(i, k), (k == j) ? dinv + (size_t)j * BLK : blk + boff(k, j), lane, 1.0);
            blk_store(tbuf + (size_t)(g * s * s + (i - b0) * s + (j - a0)) * BLK, acc, lane);
        }
        __syncthreads();
        // stage 2: X_ij = - sum_{k=b0..i} X_ik T_kj
        for (int o = wave; o < nbk * nbk; o += nw) {
            const int i = o / nbk, j = o - i * nbk;
            if (i / (2 * s) != j / (2 * s) || (i % (2 * s)) < s || (j % (2 * s)) >= s) continue;
            const int g = i / (2 * s), a0 = g * 2 * s, b0 = a0 + s;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            for (int k = b0; k <= i; ++k)
                blk_mma<false>(acc, (k == i) ? dinv + (size_t)i * BLK : blk + boff(i, k),
                               tbuf + (size_t)(g * s * s + (k - b0) * s + (j - a0)) * BLK, lane, -1.0);
            blk_store(blk + boff(i, j), acc, lane);
        }
        __syncthreads();
    }
}

extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

// element (i, k), k <= i, of the factor / of its inverse from block storage
__device__ __forceinline__ double blk_get(const double* blk, int i, int k) {
    return blk[boff(i >> 4, k >> 4) + (i & 15) * BLD + (k & 15)];
}
__device__ __forceinline__ double inv_get(const double* blk, const double* dinv, int i, int k) {
    const int bi = i >> 4, bk = k >> 4;
    return (bi == bk) ? dinv[(size_t)bi * BLK + (i & 15) * BLD + (k & 15)] : blk[boff(bi, bk) + (i & 15) * BLD + (k & 15)];
}

template <bool IN_LDS>
__device__ void role_factor(const PreLayer& Lin, int stop_after, unsigned long long* stamps, int stamp_p) {
    PreLayer L = Lin;
    if (L.variance_dev) L.variance = *L.variance_dev;        // a device-resident (trained) kernel variance
    PRE_STAMP(0);
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int M = L.M, D = L.D, Mp = L.Mp;
    const WsLayout w = ws_layout(Mp);
    const int nbk = w.nbk;
    // LDS carve: rinv [Mp] doubles | (IN_LDS: blocks, dinv, tbuf) | zs [Mp][ZLD] floats | zn [Mp] | zcs [32]
    double* sm = reinterpret_cast<double*>(smem_raw);
    double* rinv = sm;
    double* base = IN_LDS ? sm + Mp : L.ws;
    double* blk = base + w.blk;
    double* dinv = base + w.dinv;
    double* tbuf = base + w.tbuf;
    float* zs = reinterpret_cast<float*>(sm + Mp + (IN_LDS ? w.total : 0));
    float* zn = zs + (size_t)Mp * ZLD;          // |zs_m - zc|^2
    float* zcs = zn + Mp;                       // centre of the scaled inducing inputs
    double* znd = reinterpret_cast<double*>(zcs + 32);   // the same squared norms in float64 (Gram)

    // scaled inducing inputs, float32-rounded (the values the K_uf Gram also sees)
    // (1024 threads: thread (pr = tid >> 5, d = tid & 31) loads exactly the rows m = pr, pr + 32, .. of column d -- the 32 partial sums of
    // the column means below are formed right here, in the same order, without waiting for the tile)
    const bool fused_sum = nthreads == 1024;
    double colpart = 0.0;
    for (int idx = tid; idx < Mp * 32; idx += nthreads) {
        const int m = idx >> 5, d = idx & 31;
        float v = 0.f;
        if (m < M && d < D) v = (float)((double)L.Z[(size_t)m * D + d] / (double)L.ls[d]);
        zs[m * ZLD + d] = v;
        if (m < M) colpart += (double)v;
    }
    if (fused_sum) (znd + Mp)[(tid >> 5) * 32 + (tid & 31)] = colpart;
    if (tid < 32) L.cst[tid] = (tid < D) ? (float)(1.0 / (double)L.ls[tid]) : 0.f;
    const int lg_sigma = (int)ceilf(0.5f * log2f(fmaxf(L.variance, 1e-30f)));
    const bool st1_16 = (L.nbk <= 8) && ((L.nbk & 1) == 0);  // this layer's solve takes split-f16 off-diagonal updates (iwvi_common.h: IWVI_CST_U)
    const int est = st1_16 ? 7 - lg_sigma : 0;
    const float st1_iu = ldexpf(1.f, -2 * est), st1_sc = ldexpf(1.f, est);
    if (tid == 32) L.cst[IWVI_CST_SA] = ldexpf(1.f, (st1_16 ? 7 : 10) - lg_sigma);   // 2^ea: the split-f16 scale of a = Lm^-1 k (|a| <= sigma); = 2^est when stage 1 writes the planes itself
    if (tid == 33) L.cst[IWVI_CST_U] = ldexpf(1.f, 2 * est);
    if (tid == 34) L.cst[IWVI_CST_SB] = ldexpf(1.f, est);
    __syncthreads();
    // centre: K_uf is formed as exp2(x~ . z~) with |x|^2 + |z|^2 - 2 x.z expanded (like gpflow's
    // square_dist); subtracting a common centre leaves r^2 unchanged and keeps the expansion well scaled
    {   // column means of zs: 32 partial sums per column, then one thread per column adds them in a fixed order
        double* partd = znd + Mp;                                    // [32][32] partial sums
        const int d = tid & 31, pr = tid >> 5;                       // 32 parts (1024 threads)
        if (!fused_sum) {
            if (pr < 32) {
                double acc = 0.0;
                for (int m = pr; m < M; m += 32) acc += (double)zs[m * ZLD + d];
                partd[pr * 32 + d] = acc;
            }
            __syncthreads();
        }
        if (tid < 32) {
            double acc = 0.0;
            for (int q = 0; q < 32; ++q) acc += partd[q * 32 + tid];
            const float c = (tid < D) ? (float)(acc / (double)M) : 0.f;
            zcs[tid] = c;
            L.cst[32 + tid] = c;
        }
    }
    __syncthreads();
    for (int m = tid; m < Mp; m += nthreads) {                       // centred (and re-rounded) from here on: the
        double n2 = 0.0;                                             // values K_uu and K_uf both see
        for (int d = 0; d < D; ++d) {
            const float c = zs[m * ZLD + d] - zcs[d];
            zs[m * ZLD + d] = c;
            n2 = fma((double)c, (double)c, n2);
        }
        znd[m] = n2;
        zn[m] = (float)n2;
    }
    __syncthreads();
    PRE_STAMP(1);
    if (stop_after == 1) return;
    // Gram in float64, lower blocks only, one wave per 16x16 block: z_i . z_j by v_mfma_f64_16x16x4_f64 (the inner
    // dimension is D <= 32), then r^2 = |z_i|^2 + |z_j|^2 - 2 z_i.z_j and the kernel value: ~35 fp64 instructions
    // per element instead of ~100.  Generated block column by block column, two columns ahead of the
    // factorisation, by the waves that are not busy with the diagonal pass (chol_blocks).
    auto gen = [&](int bj0, int bj1, int w, int nwv) {
        const int lane = tid & 63;
        const int r = lane & 15, g = lane >> 4, nk4 = (D + 3) >> 2;
        int o = w;
        for (int bj = bj0; bj < bj1; ++bj) {
            const int nb_col = nbk - bj;
            for (; o < nb_col; o += nwv) {
                const int bi = bj + o;
                f64x4 acc = {0.0, 0.0, 0.0, 0.0};
                for (int kk = 0; kk < nk4; ++kk) {
                    const double a = (double)zs[(NB * bi + r) * ZLD + 4 * kk + g];
                    const double b = (double)zs[(NB * bj + r) * ZLD + 4 * kk + g];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
                const int j = NB * bj + r;
                const double nj = znd[j];
                const bool pad_blk = (NB * bi + NB > M) || (NB * bj + NB > M);   // only the last block row / column can hold padding
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = NB * bi + g + 4 * e;                   // f64 C/D map: row = (lane >> 4) + 4 * reg
                    const double r2 = fmax(znd[i] + nj - 2.0 * acc[e], 0.0);
                    double v = kern_value(r2, L.kern_type, (double)L.variance);
                    if (bi == bj) v += (i == j) ? L.jitter : 0.0;
                    if (pad_blk) v = (i >= M || j >= M) ? ((i == j) ? 1.0 : 0.0) : v;   // identity padding
                    blk[boff(bi, bj) + (g + 4 * e) * BLD + r] = v;
                }
            }
            o -= nb_col;                                                 // continue the round-robin in the next column
        }
    };
    // what only the layer kernel needs (nothing here waits for it): done by the worker waves beside the last diagonal pass
    auto tail = [&](int t, int nt) {
        if (t < 64) {                                                // extent of the inducing cloud in lengthscale units:
            float mx = 0.f;                                          // the layer kernel picks its Gram form by it
            for (int m = t; m < M; m += 64) mx = fmaxf(mx, zn[m]);
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            if (t == 0) L.cst[64] = mx;
        }
        // Gram operand of K_uf in A-fragment order
        const int nsteps = round_up(D + 2, 4) / 4;
        const bool rbf = L.kern_type == IWVI_KERN_RBF;
        const double c = 1.4426950408889634;
        const double l2v = log2((double)L.variance);
        const float rns = 1.0f / (float)nsteps;
        for (int idx = t; idx < nbk * nsteps * 64; idx += nt) {
            const int lane = idx & 63, wb = idx >> 6;
            const int bi = (int)(((float)wb + 0.5f) * rns), s = wb - bi * nsteps;   // exact for these small integers
            const int m = 16 * bi + (lane & 15), f = 4 * s + (lane >> 4);
            float v = 0.f;
            if (m < M) {
                if (f < D) v = rbf ? (float)(c * (double)zs[m * ZLD + f]) : -2.f * zs[m * ZLD + f];
                else if (f == D) v = rbf ? (float)c : 1.f;
                else if (f == D + 1) v = rbf ? (float)(-0.5 * c * znd[m] + l2v) : zn[m];
            } else if (f == D + 1 && rbf) v = -1.0e30f;              // padding rows: k = exp2(-huge) = 0
            L.ZtP[idx] = v;
        }
    };
    PRE_STAMP(2);
    if (stop_after == 2) return;
    // post-processing of a finished block column bj, run by the waves that do not factor: the packed float32 solve
    // stream of the column (its first block is the inverse of the diagonal block, from the factoring wave), column-block major: [L(bj,bj)^-1, -L(bj+1,bj), .., -L(nbk-1,bj)]; identity padding -> 0
    auto post = [&](int bj, int b, int ln) {
        // one item = one lane's four consecutive floats of a packed block (one 16-byte store): lane (g, ii) holds
        // G[ii][4g .. 4g+3]; block 0 of the column is the diagonal block's inverse, then the blocks below it
        float4* dst = reinterpret_cast<float4*>(L.LsP + (size_t)tri_upper_off(nbk, bj) * BLK16);
        const bool full = (M == Mp);
        {
            const int it = b * 64 + ln;
            const int ii = ln & 15, k0 = 4 * (ln >> 4);
            float v[4];
            if (b == 0) {                                        // L(bj,bj)^-1 = transpose of the factoring wave's L^-T
#pragma unroll
                for (int sgm = 0; sgm < 4; ++sgm) {
                    const int kk = k0 + sgm;
                    const double x = dinv[(size_t)bj * BLK + kk * BLD + ii];
                    const int i = 16 * bj + ii, k = 16 * bj + kk;
                    float f = (kk <= ii) ? (float)x : 0.f;
                    if (!full) f = (i < M && k < M) ? f : ((i == k) ? 1.f : 0.f);   // padded rows solve to 0 against k = 0 anyway
                    v[sgm] = f;
                }
            } else {
                const int bi = bj + b;
                const double* src = blk + boff(bi, bj) + ii * BLD + k0;
#pragma unroll
                for (int sgm = 0; sgm < 4; ++sgm) {
                    float f = -(float)src[sgm];
                    if (!full) f = (16 * bi + ii < M && 16 * bj + k0 + sgm < M) ? f : 0.f;
                    v[sgm] = f;
                }
            }
            float4 o = make_float4(v[0], v[1], v[2], v[3]);
            if (st1_16) {                                        // split-f16 solve (iwvi_common.h: IWVI_CST_U): Dinv times 1/U, the other
                if (b == 0) { o.x *= st1_iu; o.y *= st1_iu; o.z *= st1_iu; o.w *= st1_iu; }   // blocks as [h1 x 4 | h2 x 4] of 2^est (-L(bi, bj))
                else {
                    _Float16 h[8];
#pragma unroll
                    for (int sgm = 0; sgm < 4; ++sgm) { const float x = v[sgm] * st1_sc; h[sgm] = (_Float16)x; h[4 + sgm] = (_Float16)(x - (float)h[sgm]); }
                    o = *reinterpret_cast<const float4*>(h);
                }
            }
            dst[it] = o;
        }
    };
    chol_blocks(blk, nbk, rinv, dinv, tid, nthreads, gen, post, tail, stamps, stamp_p);
    PRE_STAMP(3);
    if (stop_after == 3 || stop_after > 30) return;
    PRE_STAMP(4);
    PRE_STAMP(5);
    const bool dense = (L.flags & IWVI_GP_WANT_DENSE) != 0;
    if (dense || (L.flags & IWVI_GP_WANT_LM)) {                  // (WANT_LM: the factor only; iwvi_gp_dense_inverse forms Lm^-1 on many CUs)
        for (int idx = tid; idx < Mp * Mp; idx += nthreads) {
            const int i = idx / Mp, k = idx - i * Mp;
            L.Lm[idx] = (k <= i) ? blk_get(blk, i, k) : 0.0;
        }
        __syncthreads();
    }
    if (nbk >= 16) {                                              // (== FW_SB_MIN_NBK of csrc/dgp_forward.hip)
        // M > 240: the layer kernel's solve a = Lm^-1 k runs super-block by super-block (8 block rows = 128 rows at a time):
        //   r_I = k_I - L(I, <I) a_<I   (a dense product, every wave busy)      a_I = (L_II)^-1 r_I   (a triangular product)
        // so that nothing in it is a dependent chain of one wave.  Its operand stream REPLACES the column-major substitution
        // stream in LsP (same number of blocks): per super-block I, row by row, [-L(bi, 0 .. 8I-1)], then row by row
        // [(L_II)^-1 (bi, 8I .. bi)].  The super-block inverses are the first three doubling steps of the dense inversion.
        invert_blocks(blk, dinv, tbuf, rinv, nbk, tid, nthreads, 1, 8);
        float4* dst = reinterpret_cast<float4*>(L.LsP);
        const bool full = (M == Mp);
        const int nsb = (nbk + 7) / 8;
        int off = 0;                                              // blocks written so far
        for (int I = 0; I < nsb; ++I) {
            const int r0 = 8 * I, nr = (nbk - r0 < 8) ? nbk - r0 : 8;
            const int nx = nr * r0, ny = nr * (nr + 1) / 2;
            for (int it = tid; it < (nx + ny) * 64; it += nthreads) {
                const int b = it >> 6, ln = it & 63, ii = ln & 15, k0 = 4 * (ln >> 4);
                int bi, bk; bool inv;
                if (b < nx) { bi = r0 + b / r0; bk = b - (b / r0) * r0; inv = false; }
                else { int q = b - nx, w = 0; while ((w + 1) * (w + 2) / 2 <= q) ++w; bi = r0 + w; bk = r0 + q - w * (w + 1) / 2; inv = true; }
                float v[4];
#pragma unroll
                for (int sgm = 0; sgm < 4; ++sgm) {
                    const int i = 16 * bi + ii, k = 16 * bk + k0 + sgm;
                    double x;
                    if (!inv) x = -blk[boff(bi, bk) + ii * BLD + k0 + sgm];
                    else x = (k <= i) ? inv_get(blk, dinv, i, k) : 0.0;
                    float f = (float)x;
                    if (!full && (i >= M || k >= M)) f = (inv && i == k) ? 1.f : 0.f;      // padded rows solve to 0 against k = 0
                    v[sgm] = f;
                }
                dst[(size_t)off * 64 + it] = make_float4(v[0], v[1], v[2], v[3]);
            }
            off += nx + ny;
        }
        __syncthreads();
    }
    if (dense) {
        if (nbk >= 16) invert_blocks(blk, dinv, tbuf, rinv, nbk, tid, nthreads, 8);     // the remaining doubling steps
        else invert_blocks(blk, dinv, tbuf, rinv, nbk, tid, nthreads);
        for (int idx = tid; idx < Mp * Mp; idx += nthreads) {
            const int i = idx / Mp, k = idx - i * Mp;
            L.Linv[idx] = (k <= i) ? inv_get(blk, dinv, i, k) : 0.0;
        }
    }
    PRE_STAMP(6);
}

__device__ __forceinline__ double block_sum(double v, double* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    return red[0];
}

// Role r+1: pack tril(q_sqrt[r])^T into MFMA fragment order (upper-triangular 16x16 blocks, row-block
// major) and, from the same values, this latent GP's share of the whitened KL:
//   kl[r] = 1/2 (|q_mu[:,r]|^2 - M - sum log L_ii^2 + |tril L|^2).
// One float4 of the packed image per thread-iteration, one 16-byte store.
__device__ void role_pack_r(const PreLayer& L, int r, double* red) {
    const int nbk = L.nbk, M = L.M, R = L.R;
    const float* q = L.q_sqrt + (size_t)r * M * M;
    float4* dstm = reinterpret_cast<float4*>(L.LrTP + (size_t)r * tri_blocks(nbk) * BLK16);
    double acc = 0.0;
    const int nvec = nbk * nbk * 64;
    for (int v4 = threadIdx.x; v4 < nvec; v4 += blockDim.x) {
        const int b = v4 >> 6, bi = b / nbk, bk = b - bi * nbk;
        if (bi > bk) continue;
        const int lane = v4 & 63;
        const int i = 16 * bi + (lane & 15);
        const int k0 = 16 * bk + 4 * (lane >> 4);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + e;                      // (L_r^T)[i][k] = L_r[k][i], non-zero for k >= i
            float x = 0.f;
            if (i < M && k < M && k >= i) x = q[(size_t)k * M + i];
            o[e] = x;
            acc += (double)x * (double)x;
            if (k == i && i < M) acc -= log((double)x * (double)x);
        }
        dstm[(size_t)(tri_upper_off(nbk, bi) + (bk - bi)) * 64 + lane] = make_float4(o[0], o[1], o[2], o[3]);
    }
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
        const double v = L.q_mu[(size_t)m * R + r];
        acc += v * v;
    }
    if (r == 0) {
        // q_mu^T as MFMA A blocks [nrb][nbk]: row = latent GP (padded to 16), k = inducing point
        float4* dq = reinterpret_cast<float4*>(L.QmuP);
        for (int v4 = threadIdx.x; v4 < L.nrb * nbk * 64; v4 += blockDim.x) {
            const int b = v4 >> 6, rb = b / nbk, bk = b - rb * nbk, lane = v4 & 63;
            const int rr = 16 * rb + (lane & 15), k0 = 16 * bk + 4 * (lane >> 4);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (rr < R && k0 + e < M) ? L.q_mu[(size_t)(k0 + e) * R + rr] : 0.f;
            dq[v4] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    const double tot = block_sum(acc, red);
    if (threadIdx.x == 0) L.kl[r] = 0.5 * (tot - (double)M);
    // ---- the split-f16 image of L_r^T (and, role 1, of q_mu^T) with its power-of-two scale ----------------------------------
    if (nbk & 1) return;
    const float var = L.variance_dev ? *L.variance_dev : L.variance;
    const int ea = ((L.nbk <= 8 && (L.nbk & 1) == 0) ? 7 : 10) - (int)ceilf(0.5f * log2f(fmaxf(var, 1e-30f)));   // |a| <= sigma  ->  |a| 2^ea <= 2^10 (2^7 = 2^est where stage 1 writes the planes: role_factor)
    double mx = 0.0;
    for (int idx = threadIdx.x; idx < M * M; idx += blockDim.x) { const int k = idx / M, i = idx - k * M; if (k >= i) mx = fmax(mx, fabs((double)q[idx])); }
    __syncthreads();
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s_ = blockDim.x / 2; s_ > 0; s_ >>= 1) { if ((int)threadIdx.x < s_) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s_]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    const int er = mx > 0.0 ? 13 - (int)floor(log2(mx)) : 0;                  // max |L_r| 2^er in [2^13, 2^14)
    const float sr = ldexpf(1.f, er);
    if (threadIdx.x == 0) L.cst[IWVI_CST_FR + r] = ldexpf(1.f, -(ea + er));
    {
        const int nst = s16_slabs_total(nbk);
        unsigned short* dst = L.LrT16 + (size_t)r * nst * 1024;                 // 1024 halves per slab (2 planes x 512)
        for (int v = threadIdx.x; v < nst * 64; v += blockDim.x) {            // one lane-vector (8 k) of a slab per thread-iteration
            const int sl = v >> 6, lane = v & 63;
            int bi = 0, o = 0;
            while (o + s16_slabs(nbk, bi) <= sl) { o += s16_slabs(nbk, bi); ++bi; }
            const int kc = ((bi & ~1) >> 1) + (sl - o);                       // 32-chunk of k
            const int i = 16 * bi + (lane & 15), k0 = 32 * kc + 8 * (lane >> 4);
            _Float16 h1[8], h2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e;
                float x = 0.f;
                if (i < M && k < M && k >= i) x = q[(size_t)k * M + i] * sr;  // (L_r^T)[i][k] = L_r[k][i]
                h1[e] = (_Float16)x; h2[e] = (_Float16)(x - (float)h1[e]);
            }
            // the slabs of row-blocks 2p and 2p+1 are interleaved chunk by chunk (they are multiplied as one step: same B vectors)
            const int slp = ((bi & 1) ? o - s16_slabs(nbk, bi) : o) + 2 * (sl - o) + (bi & 1);
            *reinterpret_cast<float4*>(dst + (size_t)slp * 1024 + lane * 8) = *reinterpret_cast<const float4*>(h1);
            *reinterpret_cast<float4*>(dst + (size_t)slp * 1024 + 512 + lane * 8) = *reinterpret_cast<const float4*>(h2);
        }
    }
    if (r == 0) {
        double mq = 0.0;
        for (int idx = threadIdx.x; idx < M * R; idx += blockDim.x) mq = fmax(mq, fabs((double)L.q_mu[idx]));
        red[threadIdx.x] = mq;
        __syncthreads();
        for (int s_ = blockDim.x / 2; s_ > 0; s_ >>= 1) { if ((int)threadIdx.x < s_) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s_]); __syncthreads(); }
        mq = red[0];
        const int eq = mq > 0.0 ? 13 - (int)floor(log2(mq)) : 0;
        const float sq = ldexpf(1.f, eq);
        if (threadIdx.x == 0) L.cst[IWVI_CST_FMEAN] = ldexpf(1.f, -(ea + eq));
        const int nkc = nbk / 2;
        for (int v = threadIdx.x; v < L.nrb * nkc * 64; v += blockDim.x) {
            const int sl = v >> 6, lane = v & 63, rb = sl / nkc, kc = sl - rb * nkc;
            const int rr = 16 * rb + (lane & 15), k0 = 32 * kc + 8 * (lane >> 4);
            _Float16 h1[8], h2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = (rr < R && k0 + e < M) ? L.q_mu[(size_t)(k0 + e) * R + rr] * sq : 0.f;
                h1[e] = (_Float16)x; h2[e] = (_Float16)(x - (float)h1[e]);
            }
            *reinterpret_cast<float4*>(L.Qmu16 + (size_t)sl * 1024 + lane * 8) = *reinterpret_cast<const float4*>(h1);
            *reinterpret_cast<float4*>(L.Qmu16 + (size_t)sl * 1024 + 512 + lane * 8) = *reinterpret_cast<const float4*>(h2);
        }
    }
}

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
        if (L.Mp <= 128) role_factor<true>(L, args.stop_after, args.stamps, args.stamp_p); else role_factor<false>(L, args.stop_after, args.stamps, args.stamp_p);
    } else if (role <= L.R) {
        role_pack_r(L, role - 1, reinterpret_cast<double*>(smem_raw));
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
__global__ __launch_bounds__(1024) void k_natgrad_small(float* q_mu, float* q_sqrt, const float* __restrict__ dq_mu, const float* __restrict__ dq_sqrt,
                                                        int M, int R, double gamma, int stop) {
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
    for (int o = wave; o < ntri; o += nw) {
        int bi, bj; tri_decode(o, bi, bj);
        const int ig = NB * bi + ri, jg = NB * bj + ri;      // this lane's A row (i) and B column (j)
        const int q = M - 1 - ig, p = M - 1 - jg;
        int k0 = M - 1 - (NB * bj + NB - 1); if (k0 < 0) k0 = 0; k0 &= ~3;
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int kb = k0; kb < M; kb += 64) {                // 16 k-steps' operands per round trip (32 would not fit 128 VGPRs)
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
        for (int e = 0; e < 4; ++e) {                        // acc[e] = C[g + 4e][ri]
            const int i = NB * bi + g + 4 * e, j = NB * bj + ri;
            double v = (i == j) ? 1.0 : 0.0;
            if (i < M && j < M) v -= gamma * acc[e];
            blk[boff(bi, bj) + (g + 4 * e) * BLD + ri] = v;
        }
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

// host side: 0 if the shape is not covered (the caller then takes the multi-launch path)
int natgrad_small(float* q_mu, float* q_sqrt, const float* dq_mu, const float* dq_sqrt, int M, int R, double gamma, hipStream_t st);

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


static size_t factor_lds_bytes(int Mp) {
    size_t d = (size_t)Mp;                                   // rinv
    if (Mp <= 128) d += ws_layout(Mp).total;                 // blocks + dinv + tbuf resident in LDS
    return d * sizeof(double) + ((size_t)Mp * ZLD + Mp + 32) * sizeof(float) + ((size_t)Mp + 1024) * sizeof(double) + 8;
}

int natgrad_small(float* q_mu, float* q_sqrt, const float* dq_mu, const float* dq_sqrt, int M, int R, double gamma, hipStream_t st) {
    const int Mp = round_up(M, NB);
    if (Mp > 128 || getenv("IWVI_NATGRAD_UNFUSED")) return 0;
    const size_t lds = sizeof(double) * ((size_t)Mp + ws_layout(Mp).total + 4 * (size_t)Mp + 16 * 128);
    int rc;
    if ((rc = ensure_lds_attr((const void*)k_natgrad_small, lds)) != IWVI_OK) return rc;
    hipLaunchKernelGGL(k_natgrad_small, dim3(R), dim3(1024), lds, st, q_mu, q_sqrt, dq_mu, dq_sqrt, M, R, gamma, getenv("IWVI_NG_STOP") ? atoi(getenv("IWVI_NG_STOP")) : 0);
    rc = check_launch("k_natgrad_small");
    return rc == IWVI_OK ? 1 : rc;
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
    out[0] = s.off_Lm; out[1] = s.off_Linv; out[2] = s.off_LsP; out[3] = s.off_LrTP;
    out[4] = s.off_QmuP; out[5] = s.off_ZtP; out[6] = s.off_cst; out[7] = s.off_kl;
    return IWVI_OK;
}

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
        { const char* e = getenv("IWVI_DEBUG_STOP"); a.stop_after = e ? atoi(e) : 0; }
        a.stamps = g_pre_stamps;
        { const char* e = g_pre_stamps ? getenv("IWVI_PRE_STAMP_P") : nullptr; a.stamp_p = e ? atoi(e) : 1; }
        size_t lds = 1024 * sizeof(double);
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
            L.LsP = (float*)(st + s.off_LsP); L.LrTP = (float*)(st + s.off_LrTP);
            L.QmuP = (float*)(st + s.off_QmuP); L.ZtP = (float*)(st + s.off_ZtP);
            L.LrT16 = (unsigned short*)(st + s.off_LrT16); L.Qmu16 = (unsigned short*)(st + s.off_Qmu16);
            L.cst = (float*)(st + s.off_cst);
            L.kl = (double*)(st + s.off_kl);
            L.ws = (double*)(st + s.off_ws);
            L.jitter = d.jitter; L.variance = d.variance; L.variance_dev = d.variance_dev;
            L.M = d.M; L.D = d.D; L.R = d.R; L.Mp = s.Mp; L.nbk = s.nbk; L.nrb = s.nrb; L.kern_type = d.kern_type; L.flags = d.flags;
            size_t la = factor_lds_bytes(s.Mp);
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

// Device code of the inducing-set factorisation shared by the precompute launch (csrc/precompute.hip) and the merged
// precompute + forward launch (csrc/dgp_forward.hip): block Cholesky with on-the-fly Gram generation (role_factor) and the
// tril(q_sqrt)^T packing + KL share (role_pack_r).  Reference lines replaced: temp_workaround.py:39,48,51,78,186-188.
#pragma once
#include "iwvi_common.h"

namespace iwvi {

constexpr int NB = 16;            // block size
constexpr int BLD = NB + 1;       // padded row stride of a block (doubles)
constexpr int BLK = NB * BLD;     // doubles per block
constexpr int ZLD = 33;           // row stride of the LDS copy of Zs (floats)

constexpr int IWVI_GP_SB_EXT_ = 1 << 8;   // PreLayer.flags, internal: the super-block inverses of a layer with M > 240 come from k_sb_inv (csrc/precompute.hip)
struct PreLayer {
    const float* Z; const float* ls; const float* q_mu; const float* q_sqrt;
    double* Lm; double* Linv; float* LsP; float* LrTP; float* QmuP; float* ZtP; float* cst; double* kl;
    double* ws;
    unsigned short* LrT16; unsigned short* Qmu16;   // split-f16 images (iwvi_common.h: s16_*)
    double jitter; float variance; const float* variance_dev;
    int M, D, R, Mp, nbk, nrb, kern_type, flags;
};
#define PRE_STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)

// eight f16 values as one 16-byte vector IN REGISTERS (an _Float16 array read back through reinterpret_cast lives in scratch memory: the
// pack role's split-f16 image loop was two scratch round trips per vector -- 8.5 of its 24 us at 512 threads)
typedef _Float16 pk_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float4 as_f4(const pk_f16x8 h) { return __builtin_bit_cast(float4, h); }
__device__ __forceinline__ void st_pub(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_pub4(float4* p, const float4 v) { *p = v; }

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

// the same for four values in lockstep: each Horner step is issued for all four before the next one, so that one wave carries four
// independent float64 chains (a dependent v_fma_f64 issues every ~16 clocks; entry after entry, the Gram generation was bound by
// that latency whenever fewer than four waves shared a SIMD -- the 512-thread workgroups of the merged launch).  Same operations per
// value, in the same order: bit-identical to exp_neg.
__device__ __forceinline__ void exp_neg4(const double (&x)[4], double (&out)[4]) {
    double n[4], t[4], p[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const double y = -fmin(x[e], 1000.0); n[e] = rint(y * 1.4426950408889634074); t[e] = fma(-n[e], 6.93147180369123816490e-01, y); }
#pragma unroll
    for (int e = 0; e < 4; ++e) { t[e] = fma(-n[e], 1.90821492927058770002e-10, t[e]); p[e] = 2.08767569878680989792e-09; }
    constexpr double C[9] = {2.50521083854417187751e-08, 2.75573192239858906526e-07, 2.75573192239858906526e-06, 2.48015873015873015873e-05,
                             1.98412698412698412698e-04, 1.38888888888888888889e-03, 8.33333333333333333333e-03, 4.16666666666666666667e-02,
                             1.66666666666666666667e-01};
#pragma unroll
    for (int c = 0; c < 9; ++c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = fma_c(p[e], t[e], C[c]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) p[e] = fma(p[e], t[e], 0.5);
#pragma unroll
    for (int e = 0; e < 4; ++e) p[e] = fma(p[e], t[e], 1.0);
#pragma unroll
    for (int e = 0; e < 4; ++e) p[e] = fma(p[e], t[e], 1.0);
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = ldexp(p[e], (int)n[e]);
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
// acc(16x16) += sign * A^T * B (A read transposed): X(r, j) = -(L_rr^-T)^T S_rj of the progressive inverse (chol_blocks)
__device__ __forceinline__ void blk_mma_tn(f64x4& acc, const double* A, const double* B, int lane, double sign) {
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const double a = sign * A[(4 * kk + g) * BLD + r];
        const double b = B[(4 * kk + g) * BLD + r];
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
// PINV (round 6, layers of eight blocks whose factor itself is not asked for; iwvi_common.h: INV8): X = L^-1 is formed ROW BY ROW beside the
// factorisation and written over the rows of L the factorisation no longer reads (row r's last use is the catch-up of column r, in
// step r - 1):   step r, A   the workers also form S_rj = sum_{k = j .. r-1} L(r, k) X(k, j), j < r  (tbuf[j]; X(k, k) = L_kk^-1 = xT_k^T)
//                step r, B   waves m .. m + r - 1: X(r, j) = -L_rr^-1 S_rj -> blk(r, j)        (beside the updates of column r + 1)
//                step r + 1, A   the workers pack row r of X (pinv_pack: the layer kernel's stage-1 operand stream)
// Only the last row is left behind the last pass: seven products, one barrier, eight blocks to pack (~0.7 us) -- the three doubling
// steps of invert_blocks on the finished factor were 4 us of this one CU.
template <bool PINV = false, class GEN, class POST, class TAIL, class PPACK = int>
__device__ __forceinline__ void chol_blocks(double* blk, int nbk, double* rinv, double* xT, int tid, int nthreads, GEN gen, POST post, TAIL tail,
                                            unsigned long long* stamps = nullptr, int stamp_p = 1, double* tbuf = nullptr, PPACK pinv_pack = PPACK()) {
    const int lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6;
    gen(0, nbk < 2 ? 1 : 2, wave, nw);
    __syncthreads();
    PRE_STAMP(7);
    // step B of a pass, by every wave: block rows below -- finish column p (rows beyond the window) and bring column p+1 up to date with it
    auto step_b = [&](int p, int m, int win) {
        for (int b = wave; b < m; b += nw) {
            const int bi = p + 1 + b;
            double* A = blk + boff(bi, p);
            double* C = blk + boff(bi, p + 1);
            if (b >= win) {
                // Round 6: L(bi, p) = A L_pp^-T is formed TRANSPOSED -- T = L_pp^-1 A^T, the same products and sums -- because the
                // accumulator of lane (g, c) then holds T[g + 4 e][c] = L(bi, p)[c][4 e + g], e = 0 .. 3: exactly what the lane feeds as
                // the A operand of the update product behind it (blk_mma: A[r][4 kk + g], r = c, kk = e).  The block goes from one matrix
                // instruction into the next in registers; before, it was stored to LDS and read back (same wave, in order: a store and a
                // load latency on the step's chain, ~0.15 us of every step that has rows beyond the window).  Its store (later catch-ups
                // read it) follows the update's instructions.
                const int r = lane & 15, g = lane >> 4;
                const double* xt = xT + (size_t)p * BLK;
                f64x4 t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    t = __builtin_amdgcn_mfma_f64_16x16x4f64(xt[(4 * kk + g) * BLD + r], A[r * BLD + 4 * kk + g], t, 0, 0, 0);
                f64x4 acc = blk_load(C, lane);
                const double* Bp = blk + boff(p + 1, p);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-t[kk], Bp[r * BLD + 4 * kk + g], acc, 0, 0, 0);
                blk_store(C, acc, lane);
#pragma unroll
                for (int e = 0; e < 4; ++e) A[r * BLD + 4 * e + g] = t[e];
                continue;
            }
            f64x4 acc = blk_load(C, lane);
            blk_mma<true>(acc, A, blk + boff(p + 1, p), lane, -1.0);
            blk_store(C, acc, lane);
        }
        if constexpr (PINV) {                                // row p of X, by the waves the updates leave idle (m + p <= 7 < nw)
            const int j = wave - m;
            if (j >= 0 && j < p) {
                f64x4 acc = {0.0, 0.0, 0.0, 0.0};
                blk_mma_tn(acc, xT + (size_t)p * BLK, tbuf + (size_t)j * BLK, lane, -1.0);
                blk_store(blk + boff(p, j), acc, lane);
            }
        }
    };
    // Round 6: the column loop exists TWICE, once for the factoring wave and once for everybody else (same barriers, same arithmetic, same
    // order: bit-identical).  In one loop, everything the worker side keeps live across a step -- the Gram generation's operands, the packing's
    // scales and addresses, the catch-up's block offsets: ~50 registers at the 128 this 1024-thread kernel may use -- was live across the
    // diagonal pass too, and the pass (32 doubles of window + its temporaries) spilled around itself: the ISA had 17 scratch loads per step,
    // two of them on the factoring wave's own path, in a chain that is 8 x 2.3 us of one wave's latency.
    if (wave == 0) {
        for (int p = 0; p < nbk; ++p) {
            const int m = nbk - 1 - p;                       // block rows below the diagonal block
            const int win = m < 2 ? m : 2;                   // of which the factoring wave carries this many
            if (p == stamp_p) PRE_STAMP(10);
            __builtin_amdgcn_s_setprio(3);                   // the serial pass is the critical path: its LDS traffic goes first
            diag_factor_window(blk, p, win, xT + (size_t)p * BLK, rinv + NB * p, lane, (stamps && p == stamp_p) ? stamps + (size_t)blockIdx.x * 16 : nullptr);
            __builtin_amdgcn_s_setprio(0);
            if (p == stamp_p) PRE_STAMP(11);
            __syncthreads();
            if (p == stamp_p) PRE_STAMP(12);
            step_b(p, m, win);
            if (m > 0 || PINV) __syncthreads();
            if (p == stamp_p) PRE_STAMP(13);
        }
    } else {
        for (int p = 0; p < nbk; ++p) {
            const int m = nbk - 1 - p;
            const int win = m < 2 ? m : 2;
            if ((wave & 3) != 0 || nw < 8) {
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
                    if constexpr (PINV) {
                        // row p - 1 of X is complete: pack it (p blocks), and form this row's sums S_pj, the longest (j = 0: p products) first,
                        // on the workers counted from the END of the rotation (the generating waves are the busy ones)
                        if (rel >= 0) for (int it = rel * 64 + lane; it < p * 64; it += nrel * 64) pinv_pack(p - 1, it >> 6, it & 63);
                        for (int j = nwo - 1 - wg; j < p; j += nwo) {
                            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
                            blk_mma<true>(acc, blk + boff(p, j), xT + (size_t)j * BLK, lane, 1.0);                  // L(p, j) X(j, j),  X(j, j) = xT_j^T
                            for (int k = j + 1; k < p; ++k) blk_mma<false>(acc, blk + boff(p, k), blk + boff(k, j), lane, 1.0);   // + L(p, k) X(k, j)
                            blk_store(tbuf + (size_t)j * BLK, acc, lane);
                        }
                    }
                }
                if (p == nbk - 1) tail(w * 64 + lane, nwo * 64);     // work nobody waits for, beside the last (otherwise idle) pass
            }
            __syncthreads();
            step_b(p, m, win);
            if (m > 0 || PINV) __syncthreads();
        }
    }
    if (tid < 64) post(nbk - 1, 0, tid);                 // the last column: its diagonal block
    if constexpr (PINV) {                                // the last row of X: one block per wave
        for (int it = tid; it < nbk * 64; it += nthreads) pinv_pack(nbk - 1, it >> 6, it & 63);
    }
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
__device__ __forceinline__ void role_factor(const PreLayer& Lin, int stop_after, unsigned long long* stamps, int stamp_p) {
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
    // (512 threads: thread (pr = tid >> 5 < 16, d) loads the rows m = pr, pr + 16, ..: two of the 32 partial sums, alternately)
    const bool fused_sum = nthreads == 1024 || nthreads == 512;
    const bool two_parts = nthreads == 512;
    double colpart = 0.0, colpart2 = 0.0;
    int odd = 0;
    for (int idx = tid; idx < Mp * 32; idx += nthreads) {
        const int m = idx >> 5, d = idx & 31;
        float v = 0.f;
        if (m < M && d < D) v = (float)((double)L.Z[(size_t)m * D + d] / (double)L.ls[d]);
        zs[m * ZLD + d] = v;
        if (m < M) { if (two_parts && odd) colpart2 += (double)v; else colpart += (double)v; }
        odd ^= 1;
    }
    if (fused_sum) {
        (znd + Mp)[(tid >> 5) * 32 + (tid & 31)] = colpart;
        if (two_parts) (znd + Mp)[((tid >> 5) + 16) * 32 + (tid & 31)] = colpart2;
    }
    if (tid < 32) st_pub(L.cst + tid, (tid < D) ? (float)(1.0 / (double)L.ls[tid]) : 0.f);
    const int lg_sigma = (int)ceilf(0.5f * log2f(fmaxf(L.variance, 1e-30f)));
    const bool inv8 = inv8_layer(L.nbk);                     // stage 1 as a product with the explicit inverse (iwvi_common.h: INV8)
    const bool st1_16 = (L.nbk <= 8) && ((L.nbk & 1) == 0) && !inv8;  // this layer's solve takes split-f16 off-diagonal updates (iwvi_common.h: IWVI_CST_U)
    const int est = st1_16 ? 7 - lg_sigma : 0;
    const float st1_iu = ldexpf(1.f, -2 * est), st1_sc = ldexpf(1.f, est);
    if (tid == 32) st_pub(L.cst + IWVI_CST_SA, ldexpf(1.f, (st1_16 ? 7 : 10) - lg_sigma));   // 2^ea: the split-f16 scale of a = Lm^-1 k (|a| <= sigma); = 2^est when stage 1 writes the planes itself
    if (tid == 33) st_pub(L.cst + IWVI_CST_U, ldexpf(1.f, 2 * est));
    if (tid == 34) st_pub(L.cst + IWVI_CST_SB, ldexpf(1.f, est));
    __syncthreads();
    // centre: K_uf is formed as exp2(x~ . z~) with |x|^2 + |z|^2 - 2 x.z expanded (like gpflow's
    // square_dist); subtracting a common centre leaves r^2 unchanged and keeps the expansion well scaled
    {   // column means of zs: 32 partial sums per column, then one thread per column adds them in a fixed order
        double* partd = znd + Mp;                                    // [32][32] partial sums
        const int d = tid & 31, pr = tid >> 5;                       // 32 parts (1024 threads)
        if (!fused_sum) {
            for (int q = pr; q < 32; q += (nthreads >> 5)) {           // the same 32 partial sums per column, whatever the thread count
                double acc = 0.0;
                for (int m = q; m < M; m += 32) acc += (double)zs[m * ZLD + d];
                partd[q * 32 + d] = acc;
            }
            __syncthreads();
        }
        if (tid < 32) {
            double acc = 0.0;
            for (int q = 0; q < 32; ++q) acc += partd[q * 32 + tid];
            const float c = (tid < D) ? (float)(acc / (double)M) : 0.f;
            zcs[tid] = c;
            st_pub(L.cst + 32 + tid, c);
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
                double r2[4], kv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) r2[e] = fmax(znd[NB * bi + g + 4 * e] + nj - 2.0 * acc[e], 0.0);   // f64 C/D map: row = (lane >> 4) + 4 * reg
                if (L.kern_type == IWVI_KERN_RBF) {                      // the four entries of a lane in lockstep (exp_neg4)
                    double hx[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) hx[e] = 0.5 * r2[e];
                    exp_neg4(hx, kv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) kv[e] = (double)L.variance * kv[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) kv[e] = kern_value(r2[e], L.kern_type, (double)L.variance);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = NB * bi + g + 4 * e;
                    double v = kv[e];
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
            if (t == 0) st_pub(L.cst + 64, mx);
        }
        // Gram operand of K_uf in A-fragment order
        const int nsteps = round_up(D + 2, 4) / 4;
        const bool rbf = L.kern_type == IWVI_KERN_RBF;
        const double c = 1.4426950408889634;
        const double l2v = log2((double)L.variance);
        const float rns = 1.0f / (float)nsteps;
        for (int i4 = t; i4 < nbk * nsteps * 16; i4 += nt) {        // four consecutive lanes' entries per thread: one 16-byte store
            const int idx0 = 4 * i4, lane0 = idx0 & 63, wb = idx0 >> 6;
            const int bi = (int)(((float)wb + 0.5f) * rns), s = wb - bi * nsteps;   // exact for these small integers
            const int f = 4 * s + (lane0 >> 4);
            float v4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 16 * bi + ((lane0 + e) & 15);
                float v = 0.f;
                if (m < M) {
                    if (f < D) v = rbf ? (float)(c * (double)zs[m * ZLD + f]) : -2.f * zs[m * ZLD + f];
                    else if (f == D) v = rbf ? (float)c : 1.f;
                    else if (f == D + 1) v = rbf ? (float)(-0.5 * c * znd[m] + l2v) : zn[m];
                } else if (f == D + 1 && rbf) v = -1.0e30f;          // padding rows: k = exp2(-huge) = 0
                v4[e] = v;
            }
            st_pub4(reinterpret_cast<float4*>(L.ZtP + idx0), make_float4(v4[0], v4[1], v4[2], v4[3]));
        }
    };
    PRE_STAMP(2);
    if (stop_after == 2) return;
    // post-processing of a finished block column bj, run by the waves that do not factor: the packed float32 solve
    // stream of the column (its first block is the inverse of the diagonal block, from the factoring wave), column-block major: [L(bj,bj)^-1, -L(bj+1,bj), .., -L(nbk-1,bj)]; identity padding -> 0
    auto post = [&](int bj, int b, int ln) {
        if (inv8) return;                                    // (the inverse route packs X = Lm^-1 behind the factorisation instead)
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
                    pk_f16x8 h;
#pragma unroll
                    for (int sgm = 0; sgm < 4; ++sgm) { const float x = v[sgm] * st1_sc; const _Float16 hh = (_Float16)x; h[sgm] = hh; h[4 + sgm] = (_Float16)(x - (float)hh); }
                    o = as_f4(h);
                }
            }
            st_pub4(dst + it, o);
        }
    };
    // the layer kernel's stage-1 stream of the inverse route (iwvi_common.h: INV8), row r of X = Lm^-1: item b < r = block (r, b) as split-f16
    // pairs of 2^lg X from the row's place in the block storage, item b == r = the diagonal block X(r, r) = xT_r^T in fp32
    const bool want_L = (L.flags & (IWVI_GP_WANT_DENSE | IWVI_GP_WANT_LM)) != 0;
    const bool pinv = inv8 && IN_LDS && !want_L;             // (the factor is asked for too: invert_blocks on the finished factor, below)
    const float inv_si = ldexpf(1.f, lg_sigma);
    auto pinv_pack = [&](int r, int b, int ln) {
        const int ii = ln & 15, k0 = 4 * (ln >> 4);
        float4* dst = reinterpret_cast<float4*>(L.LsP);
        const bool full = (M == Mp);
        float v[4];
#pragma unroll
        for (int sgm = 0; sgm < 4; ++sgm) {
            const int kk = k0 + sgm;
            const int i = 16 * r + ii, k = 16 * b + kk;
            float f = (b == r) ? ((kk <= ii) ? (float)dinv[(size_t)r * BLK + kk * BLD + ii] : 0.f) : (float)blk[boff(r, b) + ii * BLD + kk];
            if (!full && (i >= M || k >= M)) f = (i == k) ? 1.f : 0.f;            // padded rows solve to 0 against k = 0
            v[sgm] = f;
        }
        if (b == r) { st_pub4(dst + (size_t)r * 64 + ln, make_float4(v[0], v[1], v[2], v[3])); return; }
        pk_f16x8 h;
#pragma unroll
        for (int sgm = 0; sgm < 4; ++sgm) {
            const float x = fminf(fmaxf(v[sgm] * inv_si, -65504.f), 65504.f);
            const _Float16 hh = (_Float16)x; h[sgm] = hh; h[4 + sgm] = (_Float16)(x - (float)hh);
        }
        st_pub4(dst + (size_t)(8 + r * (r - 1) / 2 + b) * 64 + ln, as_f4(h));
    };
    if (pinv) chol_blocks<true>(blk, nbk, rinv, dinv, tid, nthreads, gen, post, tail, stamps, stamp_p, tbuf, pinv_pack);
    else chol_blocks(blk, nbk, rinv, dinv, tid, nthreads, gen, post, tail, stamps, stamp_p);
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
    if (inv8 && !pinv) {
        // X = Lm^-1 (float64, three doubling steps on the LDS-resident factor), packed as the layer kernel's stage-1 operand stream
        // (iwvi_common.h: INV8): 8 diagonal blocks in fp32, the 28 blocks left of the diagonal as split-f16 pairs scaled by 2^lg
        invert_blocks(blk, dinv, tbuf, rinv, nbk, tid, nthreads, 1, 8);
        float4* dst = reinterpret_cast<float4*>(L.LsP);
        const bool full = (M == Mp);
        const float si = ldexpf(1.f, lg_sigma);
        for (int it = tid; it < 36 * 64; it += nthreads) {
            const int b = it >> 6, ln = it & 63, ii = ln & 15, k0 = 4 * (ln >> 4);
            int bi, bk;
            if (b < 8) { bi = bk = b; }
            else { const int q = b - 8; int w = 1; while (w * (w + 1) / 2 <= q) ++w; bi = w; bk = q - w * (w - 1) / 2; }
            float v[4];
#pragma unroll
            for (int sgm = 0; sgm < 4; ++sgm) {
                const int i = 16 * bi + ii, k = 16 * bk + k0 + sgm;
                float f = (k <= i) ? (float)inv_get(blk, dinv, i, k) : 0.f;
                if (!full && (i >= M || k >= M)) f = (i == k) ? 1.f : 0.f;        // padded rows solve to 0 against k = 0
                v[sgm] = f;
            }
            float4 o = make_float4(v[0], v[1], v[2], v[3]);
            if (b >= 8) {
                pk_f16x8 h;
#pragma unroll
                for (int sgm = 0; sgm < 4; ++sgm) {
                    const float x = fminf(fmaxf(v[sgm] * si, -65504.f), 65504.f);
                    const _Float16 hh = (_Float16)x; h[sgm] = hh; h[4 + sgm] = (_Float16)(x - (float)hh);
                }
                o = as_f4(h);
            }
            st_pub4(dst + it, o);
        }
        __syncthreads();
    }
    if (nbk >= 16) {                                              // (== FW_SB_MIN_NBK of csrc/dgp_forward.hip)
        // M > 240: the layer kernel's solve a = Lm^-1 k runs super-block by super-block (8 block rows = 128 rows at a time):
        //   r_I = k_I - L(I, <I) a_<I   (a dense product, every wave busy)      a_I = (L_II)^-1 r_I   (a triangular product)
        // so that nothing in it is a dependent chain of one wave.  Its operand stream REPLACES the column-major substitution
        // stream in LsP (same number of blocks): per super-block I, row by row, [-L(bi, 0 .. 8I-1)], then row by row
        // [(L_II)^-1 (bi, 8I .. bi)].  The super-block inverses are the first three doubling steps of the dense inversion.
        // (round 5: unless the dense inverse is asked for, the super-block inverses are formed and packed by k_sb_inv behind this launch, one
        //  workgroup per 16-column block of each super-block -- here they were 53 us on this one CU at M = 256, 16.5 % of configs[3])
        const bool sb_ext = (L.flags & IWVI_GP_SB_EXT_) != 0 && !dense;
        if (!sb_ext) invert_blocks(blk, dinv, tbuf, rinv, nbk, tid, nthreads, 1, 8);
        float4* dst = reinterpret_cast<float4*>(L.LsP);
        const bool full = (M == Mp);
        const int nsb = (nbk + 7) / 8;
        int off = 0;                                              // blocks written so far
        for (int I = 0; I < nsb; ++I) {
            const int r0 = 8 * I, nr = (nbk - r0 < 8) ? nbk - r0 : 8;
            const int nx = nr * r0, ny = nr * (nr + 1) / 2;
            for (int it = tid; it < (sb_ext ? nx : nx + ny) * 64; it += nthreads) {
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
        else if (!inv8) invert_blocks(blk, dinv, tbuf, rinv, nbk, tid, nthreads);   // (inv8: the blocks already hold the whole inverse)
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
// sum of PACK_VT partial sums already in red[0 .. PACK_VT): the same tree for every workgroup size (bit-identical results from the
// 1024-thread precompute launch and from the 512-thread workgroups of the merged launch)
constexpr int PACK_VT = 1024;
__device__ __forceinline__ double block_sum_vt(double* red) {
    __syncthreads();
    for (int s = PACK_VT / 2; s > 0; s >>= 1) {
        for (int i = threadIdx.x; i < s; i += blockDim.x) red[i] += red[i + s];
        __syncthreads();
    }
    return red[0];
}

// Role r+1: pack tril(q_sqrt[r])^T into MFMA fragment order (upper-triangular 16x16 blocks, row-block
// major) and, from the same values, this latent GP's share of the whitened KL:
//   kl[r] = 1/2 (|q_mu[:,r]|^2 - M - sum log L_ii^2 + |tril L|^2).
// One float4 of the packed image per thread-iteration, one 16-byte store.  The KL share is accumulated by PACK_VT virtual threads
// (a workgroup of fewer threads plays several of them in turn), so its rounding does not depend on the launch geometry.
#define PACK_STAMP(k) do { if (st && threadIdx.x == 0) st[k] = wall_clock64(); } while (0)
// the role's work on q_sqrt[r] read through `q`: an LDS pointer when the matrix was staged (Mp <= 128), a global one otherwise -- a pointer
// that could be either makes every access a FLAT load with a full wait behind it (the role's loops were chains of those)
template <class QP>
__device__ __forceinline__ void role_pack_body(const PreLayer& L, int r, double* red, const QP q, const bool lg_in_lds, unsigned long long* st) {
    const int nbk = L.nbk, M = L.M, R = L.R;
    PACK_STAMP(1);
    float4* dstm = reinterpret_cast<float4*>(L.LrTP + (size_t)r * tri_blocks(nbk) * BLK16);
    const int nvec = nbk * nbk * 64;
    // log L_ii^2 of the M diagonal entries, one per thread, up front: inside the accumulation loop below four lanes of a wave reach a
    // diagonal entry at a time and the whole wave then runs the float64 log four times over (5 us of the role).  Same values, subtracted
    // at the same point of the same accumulator: the KL share is unchanged bit for bit.
    double* lg = red + PACK_VT + (lg_in_lds ? (M * M + 1) / 2 : 0);        // behind the staged q_sqrt (doubles)
    const bool lg_ok = lg_in_lds;
    if (lg_ok) {
        for (int i = threadIdx.x; i < M; i += blockDim.x) { const double x = (double)q[(size_t)i * M + i]; lg[i] = log(x * x); }
        __syncthreads();
    }
    for (int vt = threadIdx.x; vt < PACK_VT; vt += blockDim.x) {
    double acc = 0.0;
    for (int v4 = vt; v4 < nvec; v4 += PACK_VT) {
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
            if (k == i && i < M) acc -= lg_ok ? lg[i] : log((double)x * (double)x);
        }
        st_pub4(dstm + (size_t)(tri_upper_off(nbk, bi) + (bk - bi)) * 64 + lane, make_float4(o[0], o[1], o[2], o[3]));
    }
    for (int m = vt; m < M; m += PACK_VT) {
        const double v = L.q_mu[(size_t)m * R + r];
        acc += v * v;
    }
    red[vt] = acc;
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
            st_pub4(dq + v4, make_float4(o[0], o[1], o[2], o[3]));
        }
    }
    PACK_STAMP(2);
    const double tot = block_sum_vt(red);
    PACK_STAMP(3);
    if (threadIdx.x == 0) {
        L.kl[r] = 0.5 * (tot - (double)M);
    }
    // ---- the split-f16 image of L_r^T (and, role 1, of q_mu^T) with its power-of-two scale ----------------------------------
    if (nbk & 1) return;
    const float var = L.variance_dev ? *L.variance_dev : L.variance;
    const int ea = ((L.nbk <= 8 && (L.nbk & 1) == 0 && !inv8_layer(L.nbk)) ? 7 : 10) - (int)ceilf(0.5f * log2f(fmaxf(var, 1e-30f)));   // |a| <= sigma  ->  |a| 2^ea <= 2^10 (2^7 = 2^est where stage 1 writes the planes: role_factor)
    double mx = 0.0;                                             // (a wave per row, lanes along it: no division per element; max is order-free)
    for (int k = threadIdx.x >> 6; k < M; k += (int)(blockDim.x >> 6))
        for (int i = threadIdx.x & 63; i <= k; i += 64) mx = fmax(mx, fabs((double)q[(size_t)k * M + i]));
    __syncthreads();
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s_ = blockDim.x / 2; s_ > 0; s_ >>= 1) { if ((int)threadIdx.x < s_) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s_]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    PACK_STAMP(4);
    const int er = mx > 0.0 ? 13 - ilogb(mx) : 0;                             // max |L_r| 2^er in [2^13, 2^14)  (ilogb == floor(log2) exactly, without a float64 log in every thread)
    const float sr = ldexpf(1.f, er);
    if (threadIdx.x == 0) st_pub(L.cst + IWVI_CST_FR + r, ldexpf(1.f, -(ea + er)));
    {
        const int nst = s16_slabs_total(nbk);
        unsigned short* dst = L.LrT16 + (size_t)r * nst * 1024;                 // 1024 halves per slab (2 planes x 512)
        for (int v = threadIdx.x; v < nst * 64; v += blockDim.x) {            // one lane-vector (8 k) of a slab per thread-iteration
            const int sl = v >> 6, lane = v & 63;
            int bi = 0, o = 0;
            while (o + s16_slabs(nbk, bi) <= sl) { o += s16_slabs(nbk, bi); ++bi; }
            const int kc = ((bi & ~1) >> 1) + (sl - o);                       // 32-chunk of k
            const int i = 16 * bi + (lane & 15), k0 = 32 * kc + 8 * (lane >> 4);
            pk_f16x8 h1, h2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e;
                float x = 0.f;
                if (i < M && k < M && k >= i) x = q[(size_t)k * M + i] * sr;  // (L_r^T)[i][k] = L_r[k][i]
                const _Float16 hh = (_Float16)x;
                h1[e] = hh; h2[e] = (_Float16)(x - (float)hh);
            }
            // the slabs of row-blocks 2p and 2p+1 are interleaved chunk by chunk (they are multiplied as one step: same B vectors)
            const int slp = ((bi & 1) ? o - s16_slabs(nbk, bi) : o) + 2 * (sl - o) + (bi & 1);
            st_pub4(reinterpret_cast<float4*>(dst + (size_t)slp * 1024 + lane * 8), as_f4(h1));
            st_pub4(reinterpret_cast<float4*>(dst + (size_t)slp * 1024 + 512 + lane * 8), as_f4(h2));
        }
    }
    PACK_STAMP(5);
    if (r == 0) {
        double mq = 0.0;
        for (int idx = threadIdx.x; idx < M * R; idx += blockDim.x) mq = fmax(mq, fabs((double)L.q_mu[idx]));
        red[threadIdx.x] = mq;
        __syncthreads();
        for (int s_ = blockDim.x / 2; s_ > 0; s_ >>= 1) { if ((int)threadIdx.x < s_) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s_]); __syncthreads(); }
        mq = red[0];
        const int eq = mq > 0.0 ? 13 - ilogb(mq) : 0;
        const float sq = ldexpf(1.f, eq);
        if (threadIdx.x == 0) st_pub(L.cst + IWVI_CST_FMEAN, ldexpf(1.f, -(ea + eq)));
        const int nkc = nbk / 2;
        for (int v = threadIdx.x; v < L.nrb * nkc * 64; v += blockDim.x) {
            const int sl = v >> 6, lane = v & 63, rb = sl / nkc, kc = sl - rb * nkc;
            const int rr = 16 * rb + (lane & 15), k0 = 32 * kc + 8 * (lane >> 4);
            pk_f16x8 h1, h2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = (rr < R && k0 + e < M) ? L.q_mu[(size_t)(k0 + e) * R + rr] * sq : 0.f;
                const _Float16 hh = (_Float16)x;
                h1[e] = hh; h2[e] = (_Float16)(x - (float)hh);
            }
            st_pub4(reinterpret_cast<float4*>(L.Qmu16 + (size_t)sl * 1024 + lane * 8), as_f4(h1));
            st_pub4(reinterpret_cast<float4*>(L.Qmu16 + (size_t)sl * 1024 + 512 + lane * 8), as_f4(h2));
        }
    }
}


__device__ __forceinline__ void role_pack_r(const PreLayer& L, int r, double* red, unsigned long long* st = nullptr) {
    PACK_STAMP(0);
    const int nbk = L.nbk, M = L.M, R = L.R;
    const float* q = L.q_sqrt + (size_t)r * M * M;
    if (L.Mp <= 128) {
        // q_sqrt[r] -> LDS first, every thread's loads in flight at once (behind the reduction slots; both launches give a layer
        // with Mp <= 128 the factorisation's ~158 KB of LDS): the loops below read each value once or twice through dependent
        // addresses -- from global memory that was a chain of round trips (8 us with 1024 threads, 24 us with 512)
        float* qs = reinterpret_cast<float*>(red + PACK_VT);
        const int n = M * M;
        if ((n & 3) == 0) {
            const float4* q4 = reinterpret_cast<const float4*>(q);
            float4* s4 = reinterpret_cast<float4*>(qs);
            const int n4 = n >> 2, nt = (int)blockDim.x;
            for (int i0 = threadIdx.x; i0 < n4; i0 += 8 * nt) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + u * nt; v[u] = q4[i < n4 ? i : n4 - 1]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + u * nt; if (i < n4) s4[i] = v[u]; }
            }
        } else {
            for (int i = threadIdx.x; i < n; i += blockDim.x) qs[i] = q[i];
        }
        __syncthreads();
        typedef const __attribute__((address_space(3))) float* lds_cf;
        role_pack_body(L, r, red, (lds_cf)qs, true, st);
        return;
    }
    typedef const __attribute__((address_space(1))) float* glb_cf;
    role_pack_body(L, r, red, (glb_cf)q, false, st);
}

// ---- host side, shared by the two launch functions ----------------------------------------------------------
static inline size_t factor_lds_bytes(int Mp) {
    size_t d = (size_t)Mp;                                   // rinv
    if (Mp <= 128) d += ws_layout(Mp).total;                 // blocks + dinv + tbuf resident in LDS
    return d * sizeof(double) + ((size_t)Mp * ZLD + Mp + 32) * sizeof(float) + ((size_t)Mp + 1024) * sizeof(double) + 8;
}
// validated descriptor -> the kernel's view of a layer's state buffer; 0 or an IWVI_ERR_* (text set)
static inline int fill_pre_layer(const iwvi_gp_desc& d, int index, PreLayer& L) {
    if (!d.Z || !d.lengthscales || !d.q_mu || !d.q_sqrt || !d.state) {
        set_error("iwvi_gp_precompute: layer %d has a null pointer", index); return IWVI_ERR_ARG;
    }
    if (d.M <= 0 || d.M > IWVI_MAX_M || d.D <= 0 || d.D > IWVI_MAX_D || d.R <= 0 || d.R > IWVI_MAX_R) {
        set_error("iwvi_gp_precompute: layer %d size out of range (M=%d<=%d, D=%d<=%d, R=%d<=%d)",
                  index, d.M, IWVI_MAX_M, d.D, IWVI_MAX_D, d.R, IWVI_MAX_R);
        return IWVI_ERR_ARG;
    }
    if (d.kern_type != IWVI_KERN_RBF && d.kern_type != IWVI_KERN_MATERN52) {
        set_error("iwvi_gp_precompute: unknown kernel type %d", d.kern_type); return IWVI_ERR_UNSUPPORTED;
    }
    const StateLayout s = state_layout(d.M, d.R);
    char* st = (char*)d.state;
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
    return IWVI_OK;
}
}  // namespace iwvi

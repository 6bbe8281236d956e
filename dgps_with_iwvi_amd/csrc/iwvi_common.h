// Shared declarations for the gfx950 IW-ELBO kernels (internal; the public ABI is include/iwvi_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/iwvi_hip.h"

namespace iwvi {

// thread-local error text behind iwvi_last_error()
void set_error(const char* fmt, ...);
int check_launch(const char* what);
// development route switch (csrc/abi.hip; set by iwvi_debug_set_option, 0 by default): the library never reads the environment
int dbg_opt(const char* name);
// the natural-gradient step in one workgroup per latent GP (csrc/precompute.hip); 1 = launched, 0 = shape not covered, < 0 = error
int natgrad_small(float* q_mu, float* q_sqrt, const float* dq_mu, const float* dq_sqrt, int M, int R, double gamma, hipStream_t st, void* ws, size_t ws_bytes);

__host__ __device__ static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ---- MFMA operand packing (v_mfma_f32_16x16x4_f32) --------------------------------------------
// A 16x16 block of a matrix G (rows = output rows, columns = contraction index k) is stored as 64
// float4, one per lane: lane l = 16*g + i holds G[16*bi + i][16*bk + 4*g + s], s = 0..3.  MFMA step s
// of a 16-deep chunk therefore contracts the k's {4g + s : g = 0..3}; the B operand uses the same
// slot order: float4 (chunk c, g, column j) = { B[16c + 4g + s][j] }.  That is exactly what the four
// accumulator registers of lane (g, j) hold for rows 16c + 4g .. +3 (C/D map: col = lane & 15,
// row = 4*(lane >> 4) + reg), so a result tile is the next product's B operand without any shuffle.
constexpr int BLK16 = 256;                       // floats per packed 16x16 block
constexpr int IWVI_CST_FLOATS = 104;             // invls[32] | zc[32] | zmax2 = max_m |Z_m/l - zc|^2 | 2^ea | 2^-(ea+eq) | pad | 2^-(ea+e_r) [32]
constexpr int IWVI_CST_SA = 65, IWVI_CST_FMEAN = 66, IWVI_CST_FR = 72;   // split-f16 scales (see s16_* below)
// stage 1 with split-f16 off-diagonal updates (M <= 128, even block count; csrc/dgp_forward.hip: split_b16): the right-hand sides run in
// units of U = 2^(2 est), sigma 2^est -> 2^7.  [U] = U (1 when the layer's solve is fp32): the Gram tile is written times U and the LsP
// stream's Dinv blocks are packed times 1/U;  [SB] = 2^est: a_j -> the B operand of the updates, whose A blocks are 2^est (-L) in two halves
constexpr int IWVI_CST_U = 67, IWVI_CST_SB = 68;
// Round 6, layers of EXACTLY eight 16-row blocks (112 < M <= 128: the headline's M = 128): stage 1 is a triangular PRODUCT with the explicit
// inverse X = Lm^-1 instead of a blocked substitution -- a = X k, row-block i on wave (i < 4 ? i : 11 - i), no dependent chain of eight
// column solves on one wave (the substitution kept five waves busy, two of them on one SIMD, and three idle: 2.6-3.2 us per layer at the
// headline shape against ~1.3).  The layer kernel is bound by such latency chains, not by its MFMA rate (scripts/ns_scaling_headline.py:
// a workgroup of 16 samples takes 27.4 us, one of 80 takes 29.9).  LsP then holds, 1 KiB each: blocks 0-7 the diagonal blocks X(i, i) in
// fp32 A-fragment order; blocks 8 + i (i - 1) / 2 + q the blocks X(i, q), q < i, as [h1 x 4 | h2 x 4] of 2^lg X per lane (lg = ceil(log2
// sigma): exactly the form of the super-block solve's inverse part at M > 240, csrc/precompute.hip: k_pack_ls16).  [U] = [SB] = 1,
// [SA] = 2^(10 - lg).  The inverse is formed in float64 beside the factorisation (csrc/precompute_dev.h: role_factor).
// MEASURED AND NOT THE DEFAULT (round 6; profiles/r06j_inv8_stamp_phases.txt, r06k_inv8_precompute.txt, LABNOTES.md): parity-green (260 GPU
// tests), but stage 1 takes 6100-6700 clocks on every wave against 3900 / 6300 (single / doubled SIMD) of the substitution -- the splits of
// k and a, the published tile's LDS round trips and a 7-block row on one wave cost what the chain did -- and the factorisation with the
// inverse formed beside it 26.9 us against 20.0.  Build with -DIWVI_INV8=1 to reproduce.
#ifndef IWVI_INV8
#define IWVI_INV8 0
#endif
constexpr bool INV8 = IWVI_INV8 != 0;
__host__ __device__ static inline bool inv8_layer(int nbk) { return INV8 && nbk == 8; }

// triangular block storage, row-block major:
//   solve stream LsP (column-block major): column bj = [Lm(bj,bj)^-1, -Lm(bj+1,bj), .., -Lm(nbk-1,bj)],
//                                starting at block tri_upper_off(nbk, bj)
//   upper (tril(q_sqrt[r])^T):   row-block bi holds blocks bk = bi..nbk-1
__host__ __device__ static inline int tri_lower_off(int bi) { return bi * (bi + 1) / 2; }
__host__ __device__ static inline int tri_upper_off(int nbk, int bi) { return bi * nbk - bi * (bi - 1) / 2; }
__host__ __device__ static inline int tri_blocks(int nbk) { return nbk * (nbk + 1) / 2; }

// ---- per-layer state layout (see include/iwvi_hip.h) ----------------------------------------
// ---- split-f16 operands (x = h1 + h2, h1 = f16(x), h2 = f16(x - h1): 22 mantissa bits; h1 h1' + h1 h2' + h2 h1' on
// v_mfma_f32_16x16x32_f16 reproduces the fp32 product to ~2^-22) ------------------------------------------------
// A 16 x 32 SLAB of a matrix G (rows = output rows, 32 consecutive k starting at an EVEN 16-block) is 2 planes x 64 lanes x 16 B:
// lane l = 16 g + i holds G[i][8 g + j], j = 0..7, as eight f16 (plane 0: h1, plane 1: h2).  Row-block bi of the upper-triangular
// L_r^T starts at block column bi & ~1 (the block below the diagonal is stored as zeros), so it has s16_slabs(nbk, bi) slabs -- those of
// row-blocks 2p and 2p+1 (same count) are INTERLEAVED chunk by chunk from s16_slab_off(nbk, 2p) on: the forward multiplies them as one step --; the
// B operand (the a tile in LDS as 16-B vectors of eight f16: rows 8 kc + 2 g (h1) and 8 kc + 2 g + 1 (h2) hold a[32 kc + 8 g .. + 7]) needs no realignment.
// Values are pre-multiplied by a power of two per matrix (L_r: max -> [2^13, 2^14); a: 2^ea with sigma -> <= 2^10) and the
// accumulators scaled back (IWVI_CST_FR + r); only even nbk takes this path.
__host__ __device__ static inline int s16_slabs(int nbk, int bi) { return (nbk - (bi & ~1) + 1) / 2; }
__host__ __device__ static inline int s16_slab_off(int nbk, int bi) { int o = 0; for (int b = 0; b < bi; ++b) o += s16_slabs(nbk, b); return o; }
__host__ __device__ static inline int s16_slabs_total(int nbk) { return s16_slab_off(nbk, nbk); }

// M > 240 (nbk >= 16): the solve runs super-block by super-block (8 block rows each; csrc/dgp_forward.hip), and the dense part of it --
// r_I = k_I - L(I, <I) a_<I -- takes split-f16 operands when the launch does (S16): the blocks -L(bi, 0 .. 8I-1) of a block row as
// 2-KiB slabs (16 rows x 32 k, planes h1 | h2 of 2^ea (-L), lane 16 g + i holds G[i][8 g .. 8 g + 7]), 4 I slabs per row, rows in order,
// super-blocks I = 1 .. in order (k_pack_ls16, csrc/precompute.hip)
__host__ __device__ static inline int sb16_slabs(int nbk) {
    int n = 0;
    for (int I = 1; 8 * I < nbk; ++I) { const int nr = nbk - 8 * I < 8 ? nbk - 8 * I : 8; n += nr * 4 * I; }
    return nbk >= 16 ? n : 0;
}
// ... and the triangular part -- a_I = (L_II)^-1 r_I -- takes split-f16 operands for every block LEFT of the diagonal (round 5): behind the
// slabs, one 1-KiB block per (super-block I, row w = 1 .. nr - 1, q < w), rows in order, lane 16 g + i holding [h1 x 4 | h2 x 4] of
// 2^lg (L_II)^-1 [16 w + i][16 q + 4 g .. + 3] (lg = ceil(log2 sigma): the diagonal entries, >= 1 / sigma, come out >= 1) -- k_pack_ls16 again.
// |L_II^-1| <= 1 / sqrt(lambda_min(K_uu)) <= 1 / sqrt(jitter): nothing reaches the largest f16 while variance / jitter < 2^30 (variance 1000 at the default
// jitter 1e-6); the host side of the layer API asks for the fp32 variant beyond that (settings.split16_variance_ok), the packer saturates as the last resort
// (measured, round 5: 2^(lg - 3) -- room for variance / jitter < 2^36 -- costs a factor 1.2-1.8 in the error at configs[3] / [4], and carrying the
// scale in the state's constants moved the five-sub-tile variant 16 B further into scratch: configs[3] +1.7 %)
__host__ __device__ static inline int sb16_tri_blocks(int nbk) {
    int n = 0;
    for (int I = 0; 8 * I < nbk; ++I) { const int nr = nbk - 8 * I < 8 ? nbk - 8 * I : 8; n += nr * (nr - 1) / 2; }
    return nbk >= 16 ? n : 0;
}
struct StateLayout {
    int Mp, nbk, nrb, nsteps;
    size_t off_Lm, off_Linv, off_LsP, off_LrTP, off_QmuP, off_ZtP, off_cst, off_kl, off_ws, off_LrT16, off_Qmu16, off_Ls16, off_Zs, bytes;
};
__host__ __device__ static inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }
// ZtP is sized for the largest input dimension (IWVI_MAX_D) so that the layout depends on (M, R) only
__host__ __device__ static inline StateLayout state_layout(int M, int R) {
    StateLayout s;
    s.Mp = round_up(M, 16);
    s.nbk = s.Mp / 16;
    s.nrb = (R + 15) / 16;
    s.nsteps = round_up(IWVI_MAX_D + 2, 4) / 4;   // allocation; the packing uses the layer's own step count
    const size_t ntri = (size_t)tri_blocks(s.nbk);
    size_t o = 0;
    s.off_Lm = o;    o = align256(o + sizeof(double) * s.Mp * s.Mp);
    s.off_Linv = o;  o = align256(o + sizeof(double) * s.Mp * s.Mp);
    s.off_LsP = o;   o = align256(o + sizeof(float) * ntri * BLK16);
    s.off_LrTP = o;  o = align256(o + sizeof(float) * (size_t)R * ntri * BLK16);
    s.off_QmuP = o;  o = align256(o + sizeof(float) * (size_t)s.nrb * s.nbk * BLK16);
    s.off_ZtP = o;   o = align256(o + sizeof(float) * (size_t)s.nbk * s.nsteps * 64);
    s.off_cst = o;   o = align256(o + sizeof(float) * IWVI_CST_FLOATS);   // invls[32] | zc[32] | zmax2, pad
    s.off_kl = o;    o = align256(o + sizeof(double) * IWVI_MAX_R);
    {   // factorisation workspace: 16x16 blocks (17-double rows) of the lower triangle + inverses + scratch
        const size_t nbk = s.nbk;
        const size_t blocks = nbk * (nbk + 1) / 2 + nbk + (nbk * nbk + 3) / 4;
        s.off_ws = o; o = align256(o + sizeof(double) * blocks * 16 * 17);
    }
    // split-f16 images of L_r^T and q_mu^T for v_mfma_f32_16x16x32_f16 (s16_*): 2-KiB slabs (16 rows x 32 k, two f16 planes)
    s.off_LrT16 = o; o = align256(o + (size_t)R * s16_slabs_total(s.nbk) * 2048);
    s.off_Qmu16 = o; o = align256(o + (size_t)s.nrb * ((s.nbk + 1) / 2) * 2048);
    s.off_Ls16 = o;  o = align256(o + (size_t)sb16_slabs(s.nbk) * 2048 + (size_t)sb16_tri_blocks(s.nbk) * 1024);
    // float64 stage-1 route (IWVI_GP_F64_STAGE1): the centred, scaled inducing inputs z~ as PLAIN float32 [Mp][IWVI_MAX_D] -- the very values
    // the factorisation saw (ZtP holds them times log2 e, rounded again) -- written by k_f64_prep behind the precompute launch
    s.off_Zs = o;    o = align256(o + sizeof(float) * (size_t)s.Mp * IWVI_MAX_D);
    s.bytes = o;
    return s;
}

// ---- Philox4x32-10 (shared by the standalone generator and the in-kernel draws) ---------------
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
// words (0,1) -> (r cos, r sin), words (2,3) likewise.  Hardware transcendentals (v_log_f32 = log2,
// v_sin/v_cos take revolutions): |error| ~ 1e-6 on an N(0,1) draw, irrelevant for a noise source; parity tests
// read the draws back (noise_out) and feed the very same values to the oracle.
__device__ __forceinline__ void box_muller4(const uint32_t c[4], float v[4]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float u1 = ((float)c[2 * p] + 0.5f) * 2.3283064365386963e-10f;       // (0,1)
        float u2 = ((float)c[2 * p + 1] + 0.5f) * 2.3283064365386963e-10f;
        u1 = fminf(fmaxf(u1, 1.1754944e-38f), 0.99999994f);
        const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1)
        v[2 * p] = rad * __builtin_amdgcn_cosf(u2); v[2 * p + 1] = rad * __builtin_amdgcn_sinf(u2);
    }
}

// the layer stack's own noise stream: 4 normals for (layer li, sample t, component group q) of evaluation `step`
__device__ __forceinline__ void draw_normal4(unsigned long long seed, unsigned long long step, int li,
                                             long long t, int q, float v[4]) {
    uint32_t c[4] = {(uint32_t)t, (uint32_t)((unsigned long long)t >> 32), (uint32_t)(li * 256 + q), (uint32_t)step};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(step >> 32));
    box_muller4(c, v);
}
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(__expf(x)); }
// Encoder activation (layers.py:109,143-144: activation_func, default tf.nn.tanh) and its derivative from the OUTPUT a = act(x)
__device__ __forceinline__ float enc_act(float x, int act) {
    switch (act) {
        case IWVI_ACT_RELU: return fmaxf(x, 0.f);
        case IWVI_ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
        case IWVI_ACT_SOFTPLUS: return softplus_f(x);
        case IWVI_ACT_IDENTITY: return x;
        default: return tanhf(x);
    }
}
__device__ __forceinline__ float enc_act_grad(float a, int act) {
    switch (act) {
        case IWVI_ACT_RELU: return a > 0.f ? 1.f : 0.f;
        case IWVI_ACT_SIGMOID: return a * (1.f - a);
        case IWVI_ACT_SOFTPLUS: return 1.f - __expf(-a);
        case IWVI_ACT_IDENTITY: return 1.f;
        default: return 1.f - a * a;
    }
}

}  // namespace iwvi

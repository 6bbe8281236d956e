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
// One 512-thread workgroup (8 waves, two per SIMD) owns a chunk of 16*NS samples, all layers.
//   prologue  every small operand of every layer (1/lengthscales, mixing W, mean-function A, the Gram operand
//             Z~, encoder weights), the chunk's data rows and the injected noise are copied global -> LDS with
//             asynchronous LDS-DMA loads, all in flight at once; in-kernel Philox draws overlap their latency.
//   per GP layer:
//   x~      augmented, scaled, centred inputs (so that the Gram is one small MFMA product): produced by the
//           PREVIOUS layer's last phase straight from its results; only a first GP layer has a phase of its own
//   Gram    k = exp2(Z~ x~)   16x16x4 f32 MFMA, result already in B-operand order               -> LDS kuf
//   stage 1 a = Lm^-1 k by blocked right-looking forward substitution (matrix_triangular_solve, :51): one wave
//           per 16-sample sub-tile streams the packed factor (staged in LDS one layer ahead); a_j = Dinv_j r_j, then
//           r_i -= L_ij a_j for all i > j (independent MFMA chains); the B operand is the result tile just computed,
//           still in registers (accumulator layout == B layout).  Even block counts <= 8: the updates
//           run on split-f16 operands (split_b16) and a is written as the f16 planes stage 2 reads        -> LDS at, |a|^2
//   stage 2 u_r = tril(q_sqrt_r)^T a (upper blocks) -> |u_r|^2 only (never stored); mean = q_mu^T a
//           a wave owns one 16-row block of the output for ALL NS sub-tiles: each 1-KiB packed A block is
//           loaded once (coalesced, L2 -> registers, prefetched three blocks ahead, across job boundaries) and
//           feeds 4*NS MFMAs; the row-block jobs are dealt to the waves by the host (plan_stage2) in contiguous
//           runs levelled per SIMD pair; a block's operand fetches are scheduled between its MFMAs
//   epilogue var, sample; mixing + mean function as one small MFMA product that also emits the next layer's x~
//           (+ optional HBM outputs)
//   tail    per-sample log-weight, chunk-local logsumexp over K, one partial per workgroup; the last workgroup to
//           arrive finishes the ELBO (models.py:138-150) and advances the device-resident noise counter
// Every partial sum goes to its own LDS slot / workspace word and is added in a fixed order: results are
// bit-reproducible.
#include "iwvi_common.h"
#include "precompute_dev.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace iwvi {

using f32x4 = __attribute__((ext_vector_type(4))) float;
// streamed operands are read through explicitly global pointers: a generic (flat) load must be waited for with
// vmcnt(0) before the next one issues, which would serialise the prefetch ring
typedef const __attribute__((address_space(1))) f32x4* gptr4;
typedef const __attribute__((address_space(1))) float* gptr1;
typedef __attribute__((address_space(1))) float* gout1;          // outputs: a flat store poisons the vmcnt bookkeeping
typedef __attribute__((address_space(1))) f32x4* gout4;

static unsigned long long* g_stamp_buf = nullptr;   // diagnostic; see iwvi_debug_set_stamps
static long long g_stamp_wgs = 0;
static int g_dbg_exit = 0;                           // diagnostic; see iwvi_debug_set_exit
static int g_last_variant = 0;                       // diagnostic; see iwvi_debug_last_forward_variant

constexpr int XSTR_MAX = 37;          // largest row stride (floats) of the activation tiles: D, P <= 32, D + 2 <= 36
constexpr int FW_MAXNS = 5;
// From this many 16-row blocks (M > 240) stage 1 runs super-block by super-block with the packed inverses of the 128 x 128
// diagonal super-blocks (csrc/precompute.hip writes that operand stream under the same condition).  Below it the column-at-a-time
// substitution stays: it is the more accurate form when K_uu is ill-conditioned (an explicit 128-row inverse in float32 amplifies
// rounding by cond(L_II); the 1-D, M = 160 case of tests/test_gpu_random_sweep.py sits at the edge of its 5e-3 tolerance), and
// with at most 15 block columns its idle-wave loss is small.
constexpr int FW_SB_MIN_NBK = 16;

#ifndef IWVI_SBD
#define IWVI_SBD 0
#endif
#ifndef IWVI_SBT
#define IWVI_SBT 0
#endif
// The machine scheduler SINKS an operand prefetch to just in front of its first use when registers are short (round 5, found in the
// ISA of the super-block solve: its "four slabs in flight" were loaded one step ahead of nothing -- every step waited for a full L2 round
// trip with vmcnt(0)): a scheduling barrier behind the refill of a ring slot keeps the load where the source puts it.
#ifndef IWVI_NOPIN
#define FW_PIN_LOADS() __builtin_amdgcn_sched_barrier(0)
#else
#define FW_PIN_LOADS() do { } while (0)
#endif
constexpr int dbgSBD = IWVI_SBD, dbgSBT = IWVI_SBT;   // development builds: -DIWVI_SBD=n -DIWVI_SBT=n override the look-ahead depths of the super-block solve
#ifndef IWVI_REBASE_ALL
#define IWVI_REBASE_ALL 0
#endif
constexpr bool REBASE_ALL = IWVI_REBASE_ALL != 0;
constexpr int FW_THREADS = 512;
constexpr int FW_WAVES = FW_THREADS / 64;

struct alignas(16) FwGp {             // copied LDS -> registers in 16-byte pieces
    const f32x4* LsP; const f32x4* LrTP; const f32x4* QmuP; const float* ZtP; const float* cst;
    const float* W; const float* mfA; const float* mfb;
    float* a_out; float* u_out;
    int M, Mp, nbk, nrb, nsteps, R, P, kern_type, mf_type;
    int zt_off;                       // LDS offset of the staged Z~ (floats), or -1: read it from L2
    int ls_off;                       // LDS offset of the staged solve stream LsP, or -1: stream it from L2
    float variance;
    // static stage-2 schedule: wave w streams a run of (r-major, bi ascending) row-block jobs of the
    // contiguous LrTP image (nblk[w] packed blocks), after the q_mu^T row-blocks assigned to it (mean_wave)
    unsigned char jr[FW_WAVES], jbi[FW_WAVES];   // first job of wave w: latent GP jr[w], row-block jbi[w]
    unsigned short nblk[FW_WAVES];
    signed char mean_wave[2];
    signed char s16;                  // stage 2 runs on the split-f16 images: nblk counts 2-KiB slabs, LrTP / QmuP point to them
    signed char f64;                  // IWVI_LAYER_F64_STAGE1 (host-side planning only: the kernel reads FWF_F64 of the hot descriptor)
    // the same schedule as ONE 8-byte record per wave (one LDS read at the top of the layer instead of five and a loop over row-blocks):
    // low word = offset of the run in LrTP, 16-byte units; high word = nblk | jr << 16 | jbi << 24 | (wave takes q_mu^T row-block 0 / 1) << 30 / 31
    unsigned long long s2w[FW_WAVES];
};
struct FwLv {
    const float* W[IWVI_MAX_ENC]; const float* b[IWVI_MAX_ENC];
    const float* enc_out;             // precomputed encoder output [data rows, 2*Lw] (iwvi_model_precompute), or NULL
    float* kl_local;
    int dims[IWVI_MAX_ENC + 1];
    int n_enc, Lw, sampled_kl, wtotal, maxdim, act;
};
// leading fields of a layer descriptor; nx_*: the following layer when that is a GP layer (its Gram operand is
// produced by this layer's last phase)
#define FW_LAYER_HEAD \
    int type, D, zero_noise; \
    int c_off, z_off;                 /* LDS offsets (floats): this layer's constants / its noise [dims][NSAMP] */ \
    int nx_gp, nx_c_off, nx_nsteps, nx_rbf; \
    int nx_ls_off, nx_ls_n;           /* the next staged solve stream: LDS offset (or -1) and length in floats */ \
    const float* nx_ls; \
    const float* noise; float* noise_out; float* sample; float* mean; float* var; \
    float* gmv_out; float* spare_;    /* GP: [T, 3R] = (sample | mean | variance) of the latent GPs before mixing */
struct alignas(16) FwLayerHead { FW_LAYER_HEAD };
struct FwLayer {
    FW_LAYER_HEAD
    union { FwGp gp; FwLv lv; };
};
struct FwHeadGp { FwLayerHead h; FwGp gp; };      // how a GP layer's descriptor starts: one batched LDS read
static_assert(offsetof(FwLayer, gp) == sizeof(FwLayerHead) && offsetof(FwHeadGp, gp) == sizeof(FwLayerHead), "layer head layout");
// LDS carve, float offsets (all multiples of 4)
struct FwLds { int ltab, xa, xb, xt, lw, rowi, pidx, asq, meanp, gbuf, obuf, znoise, xyrows, yrows, cnt, scratch, total; };

// Kernel arguments.  The header is read through the scalar cache; the layer table is copied to LDS with
// vector loads first thing (one round trip for all of it): read line by line through the scalar cache, a
// cold 3-KiB kernarg segment costs a memory round trip per 64 bytes, serialised by the control flow.
constexpr int FW_MAX_GLOB = 16;
// optional tail: the last workgroup to finish performs models.py:138-150 on the log-weights
struct FwElboHot {                   // what the end of the kernel reads: 24 words, fetched as ONE block of scalar loads (k_dgp_forward: tail)
    int enabled, K, K_total, mode_vi, n_glob;
    int kl_total;                    // entries of all klg arrays together
    int fast;                        // one packed atomic per workgroup carries its partial sum AND its ticket (fw_arrive); decided by the host
    int pad0;
    long long B, stride_b, stride_k;
    double scale;
    float* ms; float* logp; double* elbo; double* ws;
};
static_assert(sizeof(FwElboHot) == 96, "one s_load_dwordx16 + one s_load_dwordx8");
struct FwElbo : FwElboHot {
    const double* klg[FW_MAX_GLOB]; int klg_n[FW_MAX_GLOB];
    // optional heads of the bound's adjoint (iwvi_elbo_desc.adj_*): per-sample weights and d / d final moments from the tail, the sums by the last workgroup
    float* adj_w; float* adj_dmean; float* adj_dvar; double* adj_sums;
};
// a block of kernel-argument words as registers the compiler cannot rematerialise: it treats kernel-argument loads as free to repeat and
// re-loads a field next to each use -- one dependent scalar-cache round trip per field on whatever path reads them
template <class T>
__device__ __forceinline__ T opaque_block(const T& src) {
    static_assert(sizeof(T) % 4 == 0, "whole words");
    uint32_t w[sizeof(T) / 4];
    __builtin_memcpy(w, &src, sizeof(T));
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); ++i) asm volatile("" : "+s"(w[i]));
    T v;
    __builtin_memcpy(&v, w, sizeof(T));
    return v;
}
struct FwHead {
    int n_layers;
    const float* X; const float* XY; const float* Y;
    int Dx, XYdim, Dy;
    long long T;
    unsigned row_div, row_mod;       // T < 2^31 is enforced by the host
    float lik_variance;
    const float* lik_var_dev;        // optional device scalars, read instead of lik_variance / a layer's variance
    const float* var_dev[IWVI_MAX_STACK];
    unsigned long long seed; unsigned long long* rng_state;
    float* out_logw;
    unsigned long long* stamps;      // diagnostic only (iwvi_debug_set_stamps): 128 words per workgroup
    int ncopy;
    int nchunks;                     // chunks of 16 * NS samples (== workgroups)
    int ls_first;                    // first GP layer whose solve stream is staged in LDS (fetched in the prologue), or -1
    int n_early;                     // the last n_early waves issue the prologue's big copies (first solve stream, every Z~ image) before anything else
    unsigned zt_mask;                // GP layers whose Z~ image those waves copy (bit = layer); 0 when n_early == 0: the copy list carries them
    int nz_cnt[IWVI_MAX_STACK], nz_zoff[IWVI_MAX_STACK], nz_dims[IWVI_MAX_STACK];   // the noise plan of a stack that draws all of its noise: items, slot,
    unsigned nz_zero_mask;           // components per layer + which layers' noise is zero -- in the header: scalar loads, no table walk in LDS
    int noise_drawn, noise_any_src;  // prologue: (layer, 4-component group, sample) items drawn in the kernel; any layer with injected noise
    unsigned var_dev_mask;           // layers whose kernel variance is a device scalar (var_dev[layer]); the prologue fetches them into LDS
    unsigned pre_enc_mask;           // LV layers whose encoder output was evaluated before this launch (bit = layer): the prologue gathers their rows
    const float* lw_init;            // optional [T]: local regularisers of layers evaluated before this launch (summed per sample)
    int layer_base;                  // index of this stack's first layer in the model (keys the noise streams)
    int x_per_sample;                // X has one row per sample (the output of a layer evaluated before this launch)
    int xstr;                        // row stride of the activation tiles: odd, >= max(D + 2 padded to 4, P) of the stack
    int dbg_exit;                    // diagnostic only (iwvi_debug_set_exit): leave after phase N (1: at once, 2: after the prologue)
    FwLds lds;
    FwElbo e;
};
// prologue work lists, built by the host: plain contiguous copies global -> LDS (every small operand of every layer)
// and the per-layer noise slots; they ride in the LDS copy of the table, one entry per wave-iteration
struct alignas(16) FwCopy { const float* src; int n; int dst; };           // n floats to LDS float offset dst; n < 0: 16-byte pieces
struct alignas(16) FwNoise { const float* src; int dims, z_off, zero, layer; int pad[2]; };   // 32 bytes: one batched LDS read
constexpr int FW_MAX_COPY = 6 * IWVI_MAX_STACK;
// What the layer loop reads of a layer, 32 words, fetched through the SCALAR cache straight from the kernel-argument
// segment (two s_load_dwordx16, no VALU): making the same values wave-uniform from the LDS copy of the table costs an
// LDS read + v_readfirstlane per word in each of the eight waves -- ~0.5 us per layer.  The lines are requested first
// thing in the kernel (warm_hot) so that they have arrived when the first layer starts.  Optional per-layer outputs
// (sample / mean / var / noise / the adjoint's a, u, gmv; kl_local) stay in the LDS table and are read only when
// FWF_ANY_OUT says there is one (never on the ELBO path).
enum { FWF_NX_GP = 1, FWF_NX_RBF = 2, FWF_HASW = 4, FWF_HAS_MFB = 8, FWF_ANY_OUT = 16, FWF_PRE_ENC = 32, FWF_SAMPLED_KL = 64,
       FWF_S16 = 128,        // stage 2 on split-f16 operands (LrTP / QmuP then point to the state's slab images; iwvi_common.h: s16_*)
       FWF_F64 = 256 };      // this layer's K_uf, a = Lm^-1 k and sigma^2 - |a|^2 in float64 (IWVI_LAYER_F64_STAGE1; only the F64 kernel variants)
struct alignas(64) FwHot {
    int type, D, c_off, z_off, flags;
    int nx_c_off, nx_nsteps, nx_ls_off, nx_ls_n;
    int M, Mp, nbk, nrb, nsteps, R, P, kern_type, mf_type, zt_off, ls_off;   // LV layer: R = Lw, nbk = n_enc, Mp = maxdim
    float variance;
    int ls16_off;                        // M > 240: the split-f16 slabs of the super-block solve, in 16-byte units behind LsP (iwvi_common.h: sb16_slabs)
    const float* nx_ls; const f32x4* LrTP; const f32x4* QmuP; const f32x4* LsP; const float* ZtP;   // LV: LrTP = enc_out
};
static_assert(sizeof(FwHot) == 128, "two scalar-cache lines per layer");
struct FwArgs {
    FwHead h;
    FwHot H[IWVI_MAX_STACK];
    FwLayer L[IWVI_MAX_STACK];
    FwNoise N[IWVI_MAX_STACK];
    FwCopy C[FW_MAX_COPY];
};

__host__ __device__ static inline int up4(int x) { return (x + 3) & ~3; }

// constant-block layout of a GP layer in LDS: invls[32] | zc[32] | zmax2, pad | W[P*R] | mfA[D*P] | mfb[P] | (Z~)
__host__ __device__ static inline int gpc_W() { return IWVI_CST_FLOATS; }
__host__ __device__ static inline int gpc_A(int P, int R) { return IWVI_CST_FLOATS + up4(P * R); }
__host__ __device__ static inline int gpc_b(int D, int P, int R) { return gpc_A(P, R) + up4(D * P); }
__host__ __device__ static inline int gpc_size(int D, int P, int R) { return gpc_b(D, P, R) + up4(P); }

// scratch needs (floats) of a layer for a chunk of nsamp samples
static inline int gp_scratch_floats(int Mp, int nbk, int R, int nsamp, bool f64 = false) {
    // one tile: the Gram tile is solved in place (a wave reads and overwrites only its own sample columns), plus
    // the |u|^2 slots [wave][r]; float64 route: + the float64 Gram tile and the waves' float64 shares of |a|^2
    return Mp * nsamp + FW_WAVES * R * nsamp + (f64 ? 2 * Mp * nsamp + 2 * FW_WAVES * nsamp : 0);
}
static inline int lv_scratch_floats(int maxdim, int nsamp) { return 2 * nsamp * up4(maxdim); }

__device__ __forceinline__ float kern_from_acc(float acc, int type, float var) {
    if (type == IWVI_KERN_MATERN52) {
        const float s5 = 2.2360679774997896f;
        const float r2 = fmaxf(acc, 0.f);
        const float r = sqrtf(r2 + 1e-12f);
        return var * (1.0f + s5 * r + (5.0f / 3.0f) * r2) * __expf(-s5 * r);
    }
    return __builtin_amdgcn_exp2f(acc);            // log2(var) is folded into Z~
}

// asynchronous global -> LDS copy of n floats (LDS-DMA, no VGPR round trip, nothing waits until the caller's
// vmcnt(0) + barrier).  One wave-instruction moves 64 consecutive floats; dst + i0 is wave-uniform.
__device__ __forceinline__ void async_copy_f32(const float* __restrict__ src, float* lds_dst, int n, int tid) {
    const int lane = tid & 63;
    for (int i0 = (tid & ~63); i0 < n; i0 += FW_THREADS) {
        const int i = i0 + lane;
        if (i < n)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i),
                                             (__attribute__((address_space(3))) void*)(lds_dst + i0), 4, 0, 0);
    }
}

// same with 16 bytes per lane (1 KiB per wave-instruction); src and dst 16-byte aligned, n a multiple of 4
__device__ __forceinline__ void async_copy_f32x4(const float* __restrict__ src, float* lds_dst, int n, int tid) {
    const int lane = tid & 63, n4 = n >> 2;
    for (int i0 = (tid & ~63); i0 < n4; i0 += FW_THREADS) {
        const int i = i0 + lane;
        if (i < n4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * i),
                                             (__attribute__((address_space(3))) void*)(lds_dst + 4 * i0), 16, 0, 0);
    }
}

// sum over the 16 rows of a result block, per sample column: 4 registers, then across the 4 lane groups
__device__ __forceinline__ float colsumsq4(const f32x4& v) {
    float s = v[0] * v[0];
    s = fmaf(v[1], v[1], s); s = fmaf(v[2], v[2], s); s = fmaf(v[3], v[3], s);
    return s;
}
// the same over two blocks at once, as packed fp32 pairs (v_pk_mul_f32 / v_pk_fma_f32 on the accumulators' own register pairs)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float colsumsq8(const f32x4& a, const f32x4& b) {
    const f32x2 a0 = __builtin_shufflevector(a, a, 0, 1), a1 = __builtin_shufflevector(a, a, 2, 3);
    const f32x2 b0 = __builtin_shufflevector(b, b, 0, 1), b1 = __builtin_shufflevector(b, b, 2, 3);
    f32x2 p = a0 * a0;
    p = __builtin_elementwise_fma(a1, a1, p); p = __builtin_elementwise_fma(b0, b0, p); p = __builtin_elementwise_fma(b1, b1, p);
    return p[0] + p[1];
}
// max / sum over the 32 lanes of a half-wave, every lane gets the result: four DPP steps inside each row of 16 (quad swaps, half-row mirror,
// row mirror), then one swizzle that exchanges the two rows (lane ^ 16 within 32)
__device__ __forceinline__ float dpp_f(float v, int ctrl_sel) {
    int x = __float_as_int(v), r;
    switch (ctrl_sel) {
        case 0: r = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false); break;    // quad_perm [1,0,3,2]
        case 1: r = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false); break;    // quad_perm [2,3,0,1]
        case 2: r = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false); break;   // row_half_mirror
        default: r = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false); break;  // row_mirror
    }
    return __int_as_float(r);
}
__device__ __forceinline__ float halfwave_all_max(float v) {
    v = fmaxf(v, dpp_f(v, 0)); v = fmaxf(v, dpp_f(v, 1)); v = fmaxf(v, dpp_f(v, 2)); v = fmaxf(v, dpp_f(v, 3));
    return fmaxf(v, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)));
}
__device__ __forceinline__ float halfwave_all_sum(float v) {
    v += dpp_f(v, 0); v += dpp_f(v, 1); v += dpp_f(v, 2); v += dpp_f(v, 3);
    return v + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));
}
__device__ __forceinline__ float xgroup_sum(float s) {
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    return s;
}

// N(0,1) draws for (layer li, sample t), components 4q .. 4q+3: Philox4x32-10 with
//   counter = (t_lo, t_hi, li * 256 + q, step_lo), key = (seed_lo, seed_hi ^ step_hi)
extern __shared__ __attribute__((aligned(16))) unsigned char fw_smem[];

// diagnostic phase stamps: [k] = 100 MHz wall clock, [64 + k] = shader clock; written only when registered
#define FW_STAMP(k) do { if (g.stamps && tid == 0 && (k) < 64) { \
        g.stamps[(size_t)blockIdx.x * 128 + (k)] = wall_clock64(); \
        g.stamps[(size_t)blockIdx.x * 128 + 64 + (k)] = clock64(); } } while (0)

#ifdef IWVI_S2_STEP_STAMPS   /* development: per-wave clock stamps of layer 1 into the unused rows of the stamp buffer (scripts/s2_steps.py) */
/* the rows behind the workgroups' own: row nchunks + 8 * workgroup + wave, for the first 1900 workgroups (the scripts register 32768 rows) */
#define DBG_WSTAMP(slot) do { if (g.stamps && lane == 0 && li == 1 && blockIdx.x < 1900) g.stamps[(size_t)(g.nchunks + blockIdx.x * 8 + wave) * 128 + (slot)] = clock64(); } while (0)
#else
#define DBG_WSTAMP(slot) do { } while (0)
#endif
// wave-uniform copies of values that were read from the LDS copy of the layer table
__device__ __forceinline__ int ufirst(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ T* ufirst(T* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}
template <class T>
__device__ __forceinline__ T uniform_words(const T& s) {
    // all words of the descriptor are loaded first (wide LDS reads, one wait), then made wave-uniform: reading
    // field by field costs an LDS round trip per field
    constexpr int NW = (int)(sizeof(T) / 4);
    static_assert(sizeof(T) % 16 == 0, "copied in 16-byte pieces");
    uint32_t w[NW];
    const uint4* src = reinterpret_cast<const uint4*>(&s);
#pragma unroll
    for (int i = 0; i < NW / 4; ++i) { const uint4 v = src[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)w[i]);
    T G;
    __builtin_memcpy(&G, w, sizeof(T));
    // (jb / nblk / mean_wave are read from the LDS table where they are indexed by the wave id: a register copy
    // indexed at run time would be placed in scratch)
    return G;
}

// x / n for 0 <= x < 2^20 and a small positive divisor, through the reciprocal rcp = 1.0f / n: exact there
// ((x + 0.5) / n is at least 0.5 / n away from an integer, the product's rounding error is far smaller);
// a run-time integer division costs ~40 instructions per lane
__device__ __forceinline__ int div_small(int x, float rcp) { return (int)(((float)x + 0.5f) * rcp); }

// sum_k a[k * sa] * b[k * sb], k < n, accumulated in index order.  The loads of eight terms are issued together
// (clamped indices, masked products): a plain run-time-trip loop pays one LDS round trip per term.
__device__ __forceinline__ float dot_lds(const float* a, int sa, const float* b, int sb, int n, float acc = 0.f) {
    for (int k0 = 0; k0 < n; k0 += 8) {
        float av[8], bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int k = (k0 + e < n) ? k0 + e : n - 1; av[e] = a[k * sa]; bv[e] = b[k * sb]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf((k0 + e < n) ? av[e] : 0.f, bv[e], acc);
    }
    return acc;
}

// one MFMA adds a value across the four 16-lane groups: D[i][j] = sum_k 1 * B[k][j], B[k][j] = lane (k, j)
__device__ __forceinline__ float xgroup_sum_mfma(float s) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, s, z, 0, 0, 0);
    return d[0];
}

// Gram operand of a GP layer for one 16-sample sub-tile, by one whole wave: lane (gq, jq) takes coordinates
// gq, gq + 4, .. of sample jq's input row xr and writes x~ = [x/l - zc, -|.|^2/2 (RBF) or |.|^2 (Matern52), 1, 0..]
// (4 * nsteps entries) to xo; the norm is summed over the four coordinate groups with one MFMA.
__device__ __forceinline__ void xt_subtile(const float* xr, float* xo, const float* invls, const float* zc, int D, int nsteps,
                                           bool rbf, int gq) {
    float n2 = 0.f;
    for (int d0 = 0; d0 < D; d0 += 16) {                           // loads first (clamped), then arithmetic, masked stores
        float xv[4], il[4], zz[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int d = d0 + 4 * u + gq, dc = d < D ? d : D - 1; xv[u] = xr[dc]; il[u] = invls[dc]; zz[u] = zc[dc]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int d = d0 + 4 * u + gq;
            const float v = fmaf(xv[u], il[u], -zz[u]);
            if (d < D) { xo[d] = v; n2 = fmaf(v, v, n2); }
        }
    }
    n2 = xgroup_sum_mfma(n2);
    if (gq == 0) {
        xo[D] = rbf ? -0.5f * n2 : n2;
        xo[D + 1] = 1.f;
        for (int d = D + 2; d < 4 * nsteps; ++d) xo[d] = 0.f;
    }
}

// ---- stage 1, fully unrolled for NBK <= 8 (M <= 128): right-looking blocked forward substitution with every
// right-hand-side block r_i in registers.  Column bj: a_bj = Dinv_bj r_bj (4 dependent MFMAs), then the NBK-1-bj
// updates r_i += (-L(i,bj)) a_bj are independent accumulator chains interleaved at the MFMA issue rate; the next
// column's packed blocks are loaded while this one computes.
// Split-f16 form of the off-diagonal updates (H16: every layer with an even nbk <= 8): the packed block holds, per lane,
// [h1 x 4 | h2 x 4] of 2^est (-L(bi,bj)) (csrc/precompute.hip: post()), the freshly solved a_bj is split the same way after scaling
// by sb = 2^est, and r_bi += (h1 + h2)(h1' + h2')' on v_mfma_f32_16x16x32_f16 (see the loop) -- two MFMAs of 16 clocks for four of 32.
// r therefore lives in units of U = 2^(2 est): the Gram tile is written times U and the stream's Dinv blocks are packed
// times 1/U (both exact), so a itself, |a|^2 and everything behind stage 1 are as in the fp32 form.  The diagonal solves stay fp32
// (measured, round 4: Dinv as f16 pairs against r / 2 -- two MFMAs of 16 clocks for four of 32 -- left the evaluation where it was,
// 57.8 us: the two solve waves that share a SIMD then wait on each other's vector instructions instead).
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_b16(const f32x4 a, float sb, f16x4& b1, f16x4& b2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float x = a[e] * sb; const _Float16 h = (_Float16)x; b1[e] = h; b2[e] = (_Float16)(x - (float)h); }
}
struct A16 { f16x4 h1, h2; };
__device__ __forceinline__ A16 as_a16(const f32x4 A) { A16 o; __builtin_memcpy(&o, &A, 16); return o; }

// P16 (an S16 launch): a goes to the tile as the two f16 planes stage 2 reads (its scale 2^ea = sb for these layers), i.e. the halves
// b1 / b2 just formed: lane (gq, jq) owns 8 bytes of the h1 vector and 8 of the h2 vector of (chunk bj / 2, group 2 (bj & 1) + gq / 2).
template <int NS, int NBK, bool H16, bool P16, class AP>
__device__ __forceinline__ float stage1_unrolled(AP Ap, const f32x4* kuf, f32x4* at, int tcol, int gq,
                                                 gout1 a_out_row /* or nullptr */, float sb) {
    constexpr int NSAMP = 16 * NS;
    f32x4 r[NBK], A[NBK];
#pragma unroll
    for (int i = 0; i < NBK; ++i) A[i] = Ap[(size_t)i * 64];
#pragma unroll
    for (int i = 0; i < NBK; ++i) r[i] = kuf[(i * 4 + gq) * NSAMP + tcol];
    int q = NBK;
    float ssq = 0.f;
#pragma unroll
    for (int bj = 0; bj < NBK; ++bj) {
        f32x4 An[NBK];
#pragma unroll
        for (int i = 0; i < NBK - bj - 1; ++i) An[i] = Ap[(size_t)(q + i) * 64];
        q += NBK - bj - 1;
        f32x4 res = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) res = __builtin_amdgcn_mfma_f32_16x16x4f32(A[0][s], r[bj][s], res, 0, 0, 0);
        if constexpr (H16) {
            f16x4 b1, b2;
            split_b16(res, sb, b1, b2);
            if constexpr (P16) {
                char* pl = reinterpret_cast<char*>(at) + ((size_t)((4 * bj + 2 * (gq >> 1)) * NSAMP + tcol)) * 16 + 8 * (gq & 1);
                *reinterpret_cast<f16x4*>(pl) = b1; *reinterpret_cast<f16x4*>(pl + NSAMP * 16) = b2;
            }
            // the lane's 16 bytes [h1 x 4 | h2 x 4] ARE an A operand of v_mfma_f32_16x16x32_f16 (k-slots 8 gq + e <-> h1, 8 gq + 4 + e <-> h2 of
            // k = 4 gq + e): against B = [b1 | b1] one instruction gives h1 b1' + h2 b1', against [b2 | b2] the other two terms -- the whole
            // product in two MFMAs of 16 clocks (the K = 16 form took three for three of the four terms)
            using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
            const f16x8 bb1 = __builtin_shufflevector(b1, b1, 0, 1, 2, 3, 4, 5, 6, 7), bb2 = __builtin_shufflevector(b2, b2, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int bi = bj + 1; bi < NBK; ++bi) r[bi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[bi - bj]), bb1, r[bi], 0, 0, 0);
#pragma unroll
            for (int bi = bj + 1; bi < NBK; ++bi) r[bi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[bi - bj]), bb2, r[bi], 0, 0, 0);
        } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int bi = bj + 1; bi < NBK; ++bi)
                r[bi] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[bi - bj][s], res[s], r[bi], 0, 0, 0);
        }
        }
        if constexpr (!P16) at[(bj * 4 + gq) * NSAMP + tcol] = res;
        ssq += colsumsq4(res);
        if (a_out_row) *((gout4)(a_out_row + 16 * bj + 4 * gq)) = res;
#pragma unroll
        for (int i = 0; i < NBK - bj - 1; ++i) A[i] = An[i];
    }
    return ssq;
}

// ---- stage 1 of a layer with exactly eight 16-row blocks (iwvi_common.h: INV8): a = X k with the explicit inverse X = Lm^-1, a triangular
// PRODUCT -- every wave owns one block row (waves w and w + 4 share a SIMD: rows (i, 7 - i), seven off-diagonal blocks per SIMD), nothing
// is a chain of dependent column solves.  The diagonal block of a row is an fp32 product on the row's own k (in front of the barrier), the
// blocks left of it take split-f16 operands exactly like the super-block solve's inverse part (M > 240): every row publishes s_r k as
// [h1 x 4 | h2 x 4] in place of k, block (i, q) costs two v_mfma_f32_16x16x32_f16 per sub-tile.  Ab: the layer's stream (LDS or L2) + lane.
// The wave's share of |a|^2 goes to slot 2 + wave of `asq` (epilogue (i) adds the eight in a fixed order).
template <int NS, bool P16, class AP>
__device__ __forceinline__ void stage1_inv8(AP Ab, f32x4* at, int wave, int gq, int jq, float sa, float* asq, gout1 o_a, long long t0, int nvalid, int Mp) {
    constexpr int NSAMP = 16 * NS;
    using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
    const int rw = wave < 4 ? wave : 11 - wave;
    const float s_i = 1024.0f / sa, s_r = 32.0f / (s_i * s_i);    // sa = 2^(10 - lg): s_i = 2^lg scales the packed X, |s_r k| <= 32
    const f32x4 Dg = Ab[(size_t)rw * 64];
    f32x4 Ti[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) Ti[u] = Ab[(size_t)(8 + rw * (rw - 1) / 2 + (u < rw ? u : (rw > 0 ? rw - 1 : 0))) * 64];
    f32x4 acc[NS], rs[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) acc[t] = at[(rw * 4 + gq) * NSAMP + 16 * t + jq];
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        rs[t] = acc[t] * s_r;
        f16x4 h1, h2;
        split_b16(rs[t], 1.0f, h1, h2);
        A16 pk; pk.h1 = h1; pk.h2 = h2;
        f32x4 w_;
        __builtin_memcpy(&w_, &pk, 16);
        at[(rw * 4 + gq) * NSAMP + 16 * t + jq] = w_;             // (this wave read its own block row of k: in place)
        rs[t] = rs[t] * s_i;
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Dg[s], rs[t][s], acc[t], 0, 0, 0);
    }
    __syncthreads();                                              // every row's s_r k published
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        if (q < rw) {                                             // (wave-uniform)
            const A16 ah = as_a16(Ti[q]);
            const f16x8 a1 = __builtin_shufflevector(ah.h1, ah.h1, 0, 1, 2, 3, 4, 5, 6, 7);
            const f16x8 a2 = __builtin_shufflevector(ah.h2, ah.h2, 0, 1, 2, 3, 4, 5, 6, 7);
            f32x4 b[NS];
#pragma unroll
            for (int t = 0; t < NS; ++t) b[t] = at[(q * 4 + gq) * NSAMP + 16 * t + jq];
#pragma unroll
            for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(f16x8, b[t]), acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, __builtin_bit_cast(f16x8, b[t]), acc[t], 0, 0, 0);
        }
    }
    const float un = 1.0f / (s_r * s_i);
    __syncthreads();                                              // every row has read the published k: its rows of the tile may take a
    float ssq[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        acc[t] = acc[t] * un;
        const int tcol = 16 * t + jq;
        if constexpr (P16) {                                      // the two f16 planes of 2^ea a (lane (gq, jq): 8 bytes of each vector)
            f16x4 h1, h2;
            split_b16(acc[t], sa, h1, h2);
            char* pl = reinterpret_cast<char*>(at) + ((size_t)((4 * rw + 2 * (gq >> 1)) * NSAMP + tcol)) * 16 + 8 * (gq & 1);
            *reinterpret_cast<f16x4*>(pl) = h1; *reinterpret_cast<f16x4*>(pl + NSAMP * 16) = h2;
        } else at[(rw * 4 + gq) * NSAMP + tcol] = acc[t];
        ssq[t] = colsumsq4(acc[t]);
        if (o_a && tcol < nvalid) *((gout4)(o_a + (size_t)(t0 + tcol) * Mp + 16 * rw + 4 * gq)) = acc[t];
    }
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const float sq = xgroup_sum_mfma(ssq[t]);
        if (gq == 0) asq[(2 + wave) * NSAMP + 16 * t + jq] = sq;
    }
}

// prologue copy list, entries [c_lo, c_hi): one entry per wave at a time, all DMA loads in flight together
// The compiler orders every LDS access behind every pending LDS-DMA (an `s_waitcnt vmcnt(0)` in front of each ds_read / ds_write that follows a
// global_load_lds): an entry read from the LDS table between two copies makes the second wait until the first has landed.  So: this wave's
// entries first (and the zero fills, which are LDS stores), then the copies back to back.
constexpr int FW_COPY_PER_WAVE = FW_MAX_COPY / FW_WAVES;
__device__ __forceinline__ void fw_copy_load(const FwCopy* CT, float* sm, int c_lo, int c_hi, int wave, int lane, FwCopy (&ce)[FW_COPY_PER_WAVE],
                                             int nwaves = FW_WAVES) {
#pragma unroll
    for (int k = 0; k < FW_COPY_PER_WAVE; ++k) {
        const int ci = c_lo + wave + k * nwaves;
        ce[k].src = nullptr; ce[k].n = 0; ce[k].dst = 0;
        if (ci < c_hi) {
            ce[k] = uniform_words(CT[ci]);                         // one 16-byte LDS read per entry
            if (!ce[k].src) { float* dst = sm + ce[k].dst; for (int i = lane; i < ce[k].n; i += 64) dst[i] = 0.f; }   // absent operand (e.g. no encoder bias)
        }
    }
}
__device__ __forceinline__ void fw_copy_issue(float* sm, int lane, const FwCopy (&ce)[FW_COPY_PER_WAVE]) {
#pragma unroll
    for (int k = 0; k < FW_COPY_PER_WAVE; ++k) {
        const float* src = ce[k].src;
        if (!src) continue;
        const int n = ce[k].n;
        float* dst = sm + ce[k].dst;
        if (n < 0) { for (int i0 = 0; i0 < (-n >> 2); i0 += 64) if (i0 + lane < (-n >> 2))
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * (i0 + lane)),
                                                 (__attribute__((address_space(3))) void*)(dst + 4 * i0), 16, 0, 0); }
        else { for (int i0 = 0; i0 < n; i0 += 64) if (i0 + lane < n)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i0 + lane),
                                                 (__attribute__((address_space(3))) void*)(dst + i0), 4, 0, 0); }
    }
}
__device__ __forceinline__ void fw_copy_entries(const FwCopy* CT, float* sm, int c_lo, int c_hi, int wave, int lane) {
    FwCopy ce[FW_COPY_PER_WAVE];
    fw_copy_load(CT, sm, c_lo, c_hi, wave, lane, ce);
    fw_copy_issue(sm, lane, ce);
}

// ---- one ticket per workgroup: the last arriver advances the noise stream (every workgroup has read the step counter by
//      then) and, if asked, finishes the IW-ELBO reduction of models.py:138-150.  Every workgroup of the launch arrives exactly
//      once.
// The packed arrival (FwElbo::fast; the headline shape): the workgroup's partial sum of the per-point log p travels IN the ticket -- one
// 64-bit atomic add of  fixed(part) << 18 | overflow << 9 | 1  on rng_state[1]: bits 0-8 count the arrivals, bits 18-63 accumulate the
// partial sums in units of 2^-20 (integer adds commute: the total does not depend on the order of arrival).  The workgroup whose add returns
// the count grid-1 holds the complete sum in the returned value and finishes models.py:150 at once.  Before this the end of a launch was
// three dependent round trips behind the last workgroup's arithmetic -- its partial's write-through store drained, the ticket, the loads of
// all partials by the last arriver -- and a fourth for the global KL terms (now summed in the prologue): ~4 us of every evaluation.
// A partial too large for its share of the 46-bit SIGNED field (+-2^45 units; |part| >= 2^16: with at most 511 arrivals the sum of the
// shares stays below 2^45) is stored
// exactly, tagged with the evaluation's number and drained BEFORE the add, which then carries 0 and counts in bits 9-17; the last arriver
// adds those partials from memory (a cold path: one more round trip, only when it happens).
constexpr double FX_UNIT = 1048576.0;                              // 2^20 units per 1.0
constexpr double FX_PART_MAX = 68719476736.0;                      // 2^36 units = 2^16: a workgroup's share (511 * 2^36 < 2^45, the signed field's range)
// tag of an exactly stored partial: the evaluation's number mixed with the address of the model's arrival word -- the scratch `ws` is
// whatever the caller's allocator recycled, and a block last used by ANOTHER model at the same step count must not match
__device__ __forceinline__ unsigned long long fx_tag(const unsigned long long* rng, unsigned long long step) {
    return (step + 1ULL) ^ ((unsigned long long)(uintptr_t)rng * 0x9E3779B97F4A7C15ULL);
}
template <int NS>
__device__ __forceinline__ void fw_arrive_fast(const FwArgs& gk, const FwElboHot& E, unsigned long long* rng, int nchunks, int tid, int chunk_id,
                                               double part, unsigned long long step) {
    if (tid >= 64) return;
    // the global KL terms (models.py:150): entry `lane` of the klg arrays laid end to end, requested by EVERY workgroup just before its
    // ticket -- whichever arrives last has them back together with the ticket's answer (one round trip, not two).  The first four arrays'
    // pointers and counts as one batch of scalar loads; further arrays (a stack of more than four GP layers) one by one
    double kl_lane = 0.0;
    if (tid < E.kl_total) {
        struct KlHead { const double* p[4]; int n[4]; };
        const KlHead kh = opaque_block(KlHead{{gk.h.e.klg[0], gk.h.e.klg[1], gk.h.e.klg[2], gk.h.e.klg[3]},
                                               {gk.h.e.klg_n[0], gk.h.e.klg_n[1], gk.h.e.klg_n[2], gk.h.e.klg_n[3]}});
        int idx = tid;
        const double* src = nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = i < E.n_glob ? kh.n[i] : 0;
            if (!src && idx < n) src = kh.p[i] + idx;
            idx -= n;
        }
        for (int i = 4; i < E.n_glob; ++i) {
            if (!src && idx < gk.h.e.klg_n[i]) src = gk.h.e.klg[i] + idx;
            idx -= gk.h.e.klg_n[i];
        }
        if (src) kl_lane = *((const __attribute__((address_space(1))) double*)src);
    }
    part = readlane_d(part, 0);
    const double sc = part * FX_UNIT;
    const bool ovf = !(fabs(sc) < FX_PART_MAX);                    // (a NaN partial goes the exact way too)
    const long long fx = ovf ? 0LL : __double2ll_rn(sc);
    if (ovf && chunk_id >= 0 && tid == 0) {
        __hip_atomic_store(E.ws + chunk_id, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(E.ws) + nchunks + chunk_id, fx_tag(rng, step), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long add = ((unsigned long long)fx << 18) + (ovf ? 512ULL : 0ULL) + 1ULL;
    unsigned long long old = 0ULL;
    if (tid == 0) old = __hip_atomic_fetch_add(&rng[1], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)old) |
          ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(old >> 32)) << 32);
    if ((unsigned)(old & 511ULL) != gridDim.x - 1) return;
    double kl_sum = 0.0;                                           // (in the arrays' order, like the loop it replaces)
    for (int k = 0; k < E.kl_total; ++k) kl_sum += readlane_d(kl_lane, k);
    if (tid != 0) return;
    __hip_atomic_store(&rng[1], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&rng[0], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double tot = (double)(((long long)old >> 18) + fx) * (1.0 / FX_UNIT);                 // (arithmetic shift: the field's sign)
    if ((unsigned)(old >> 9 & 511ULL) + (ovf ? 1u : 0u)) {
        for (int c = 0; c < nchunks; ++c)
            if (__hip_atomic_load(reinterpret_cast<unsigned long long*>(E.ws) + nchunks + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fx_tag(rng, step))
                tot += __hip_atomic_load(E.ws + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const double val = tot * E.scale - kl_sum;                                                 // models.py:150
    *E.elbo = val;
}

template <int NS>
__device__ __forceinline__ void fw_arrive(const FwArgs& gk, float* sm, int tid, int chunk_id) {
    constexpr int NSAMP = 16 * NS;
    const FwHead& g = gk.h;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int* counters = reinterpret_cast<int*>(sm + g.lds.cnt);
    const bool local_lse = g.e.enabled && g.e.ws && !g.e.mode_vi && g.e.stride_k == 1 && g.e.stride_b == g.e.K &&
                           (NSAMP % g.e.K) == 0;
    if (g.rng_state) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have left
        __syncthreads();
        if (tid == 0) {
            const unsigned long long tk = __hip_atomic_fetch_add(&g.rng_state[1], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (tk == (unsigned long long)gridDim.x - 1);
            if (last) {
                __hip_atomic_store(&g.rng_state[1], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&g.rng_state[0], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            counters[2] = last;
        }
        __syncthreads();
        if (g.e.enabled && counters[2]) {
            const FwElbo& E = g.e;
            double acc = 0.0;
            double acc_ds = 0.0;                                   // adjoint heads: the chunks' shares of d ELBO / d lik_variance (ws[nchunks ..))
            if (local_lse) {
                for (int i = tid; i < g.nchunks; i += FW_THREADS)
                    acc += __hip_atomic_load(E.ws + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (E.adj_sums)
                    for (int i = tid; i < g.nchunks; i += FW_THREADS)
                        acc_ds += __hip_atomic_load(E.ws + g.nchunks + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const int K = E.K;
                for (long long b = tid; b < E.B; b += FW_THREADS) {
                    const float* row = g.out_logw + b * E.stride_b;
                    float m = -INFINITY, ssum = 0.f, lsum = 0.f;
                    for (int k0 = 0; k0 < K; k0 += 8) {
                        float Lv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u)                                    // 8 independent write-through-coherent loads
                            Lv[u] = (k0 + u < K) ? __hip_atomic_load(row + (k0 + u) * E.stride_k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -INFINITY;
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            if (k0 + u < K) {
                                lsum += Lv[u];
                                if (Lv[u] > m) { ssum = ssum * __expf(m - Lv[u]) + 1.f; m = Lv[u]; }
                                else ssum += __expf(Lv[u] - m);
                            }
                        }
                    }
                    float lp;
                    if (E.mode_vi) lp = lsum / (float)K;                               // models.py:84
                    else {
                        lp = m + logf(ssum) - logf((float)E.K_total);                  // models.py:148
                        if (E.ms) { E.ms[2 * b] = m; E.ms[2 * b + 1] = ssum; }
                    }
                    if (E.logp) E.logp[b] = lp;
                    acc += (double)lp;
                }
            }
            // deterministic block sum: lanes by shuffles, then the 8 wave partials in a fixed order
            for (int o = 32; o > 0; o >>= 1) { acc += __shfl_xor(acc, o); acc_ds += __shfl_xor(acc_ds, o); }
            double* wsum = reinterpret_cast<double*>(sm + g.lds.xa);
            if (lane == 0) { wsum[wave] = acc; wsum[FW_WAVES + wave] = acc_ds; }
            __syncthreads();
            if (tid == 0 && (E.elbo || E.adj_sums)) {
                double tot = 0.0, tds = 0.0, kl = 0.0;
                for (int w = 0; w < FW_WAVES; ++w) { tot += wsum[w]; tds += wsum[FW_WAVES + w]; }
                for (int i = 0; i < E.n_glob; ++i)
                    for (int c = 0; c < E.klg_n[i]; ++c) kl += E.klg[i][c];
                const double val = tot * E.scale - kl;                                 // models.py:150
                if (E.elbo) *E.elbo = val;
                if (E.adj_sums) { E.adj_sums[0] = tot; E.adj_sums[1] = tds; E.adj_sums[2] = val; }
            }
        }
    }
}

// S16: stage 2 of every GP layer on split-f16 operands (iwvi_common.h: s16_*); BIG: some GP layer has more than 8 block rows (M > 128) --
// the launch variants for M <= 128 do not carry the column-at-a-time / super-block solves and the in-place plane conversion
// LEAN (the bound's own evaluation: every GP layer RBF, every latent-variable layer's encoder evaluated by the precompute launch, all noise drawn in
// the kernel, no per-layer output asked for, the packed arrival): none of those alternatives is compiled into the variant
// ---- float64 stage-1 route (IWVI_LAYER_F64_STAGE1; only instantiated in the F64 kernel variants) ------------------------------------
// The reference computes Kuf, A = Lm^-1 Kuf and Kdiag - sum A^2 in float64 (temp_workaround.py:44,51,59; settings.float_type).  With
// K_uu ill-conditioned (cond(Lm) ~ 1e4: M >~ 100 inducing points in a 1-3-dimensional box) a float32 k alone moves the mean by 5e-4 and
// the float32 substitution by 1e-2 .. 1e-1 (profiles/r04h_split16_error.txt); a rounded to float32 AFTER a float64 solve costs 7e-8.
//   Gram     k[m][t] = var exp(-|z~_m - x~_t|^2 / 2) from the float32 z~ the factorisation saw (state: Zs) and the layer's float32 x~ rows,
//            differenced and exponentiated in float64 (precompute_dev.h: exp_neg4), stored as the B operand of the product below:
//            k64[(bk * 4 + g) * NSAMP + t][s] = k[16 bk + 4 g + s][t]
//   solve    a = Lm^-1 k as a lower-triangular float64 product with the DENSE inverse (state: Linv, row-major; k_linv): wave w owns the
//            block rows (p, nbk - 1 - p), p = w, w + 8, .. (nbk + 1 blocks per pair: level), all NS sub-tiles; per 16 x 16 block four
//            v_mfma_f64_16x16x4_f64 per sub-tile.  A rows are fed PERMUTED (lane i supplies row 4 (i & 3) + (i >> 2)) so that the
//            accumulator registers of lane (g, j) are rows 4 g .. 4 g + 3 of the block (the f64 C/D map is row = g + 4 reg): the
//            float32 a tile stage 2 reads is written without a shuffle.
//   |a|^2    per sample in float64, the waves' shares summed in a fixed order; the layer's variance leaves as dvar = var - |a|^2.
template <int NS>
__device__ __forceinline__ void f64_gram(const FwHot& G, const float* Zs, const float* xt, int XSTR, int D, float variance,
                                         double* k64, int wave, int gq, int jq) {
    constexpr int NSAMP = 16 * NS;
    const int nbk = G.nbk, M = G.M;
    for (int idx = wave; idx < nbk * NS; idx += FW_WAVES) {
        const int bk = idx / NS, t = idx - bk * NS;
        const int m0 = 16 * bk + 4 * gq;
        const float* xr = xt + (16 * t + jq) * XSTR;
        double r2[4] = {0.0, 0.0, 0.0, 0.0};
        for (int d = 0; d < D; ++d) {
            const double xd = (double)xr[d];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const double df = (double)Zs[(size_t)(m0 + e) * IWVI_MAX_D + d] - xd; r2[e] = fma(df, df, r2[e]); }
        }
        double kv[4];
        if (G.kern_type == IWVI_KERN_RBF) {
            double hx[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) hx[e] = 0.5 * r2[e];
            exp_neg4(hx, kv);
#pragma unroll
            for (int e = 0; e < 4; ++e) kv[e] *= (double)variance;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) kv[e] = kern_value(r2[e], G.kern_type, (double)variance);
        }
        double* dst = k64 + ((size_t)(bk * 4 + gq) * NSAMP + 16 * t + jq) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e] = (m0 + e < M) ? kv[e] : 0.0;          // padded inducing rows: k = 0
    }
}
template <int NS>
__device__ __forceinline__ void f64_solve(const FwHot& G, const double* Linv, const double* k64, f32x4* at, double* part64, float* asq,
                                          float variance, gout1 o_a, long long t0, int nvalid, int tid, int wave, int gq, int jq) {
    constexpr int NSAMP = 16 * NS;
    using f64x4v = __attribute__((ext_vector_type(4))) double;
    using f64x2v = __attribute__((ext_vector_type(2))) double;
    const int nbk = G.nbk, Mp = G.Mp;
    const int prow = 4 * (jq & 3) + (jq >> 2);                    // the row of a block this lane feeds as A operand (see above)
    double ssq[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) ssq[t] = 0.0;
    for (int p = wave; p < (nbk + 1) / 2; p += FW_WAVES) {
        for (int half = 0; half < 2; ++half) {
            const int bi = half == 0 ? p : nbk - 1 - p;
            if (half == 1 && bi == p) break;                      // (odd block count: the middle row once)
            typedef const __attribute__((address_space(1))) f64x2v* gptr2d;
            gptr2d Ar = (gptr2d)(Linv + (size_t)(16 * bi + prow) * Mp + 4 * gq);
            f64x4v acc[NS];
#pragma unroll
            for (int t = 0; t < NS; ++t) acc[t] = f64x4v{0.0, 0.0, 0.0, 0.0};
            f64x2v a01 = Ar[0], a23 = Ar[1];
            for (int bk = 0; bk <= bi; ++bk) {
                const f64x2v c01 = a01, c23 = a23;
                const int nx = bk + 1 <= bi ? bk + 1 : bk;
                a01 = Ar[8 * nx]; a23 = Ar[8 * nx + 1];           // (16 doubles per block column)
                const double* kb = k64 + ((size_t)(bk * 4 + gq) * NSAMP + jq) * 4;
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                    const f64x4v b = *reinterpret_cast<const f64x4v*>(kb + (size_t)64 * t);
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(c01[0], b[0], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(c01[1], b[1], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(c23[0], b[2], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(c23[1], b[3], acc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                const int tcol = 16 * t + jq;
                const f32x4 af = {(float)acc[t][0], (float)acc[t][1], (float)acc[t][2], (float)acc[t][3]};
                at[(bi * 4 + gq) * NSAMP + tcol] = af;
                ssq[t] = fma(acc[t][0], acc[t][0], fma(acc[t][1], acc[t][1], fma(acc[t][2], acc[t][2], fma(acc[t][3], acc[t][3], ssq[t]))));
                if (o_a && tcol < nvalid) *((gout4)(o_a + (size_t)(t0 + tcol) * Mp + 16 * bi + 4 * gq)) = af;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NS; ++t) {                                // this wave's share of |a|^2 per sample (0 for a wave without rows)
        double v = ssq[t];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (gq == 0) part64[wave * NSAMP + 16 * t + jq] = v;
    }
    __syncthreads();
    if (tid < NSAMP) {
        double sq = 0.0;
#pragma unroll
        for (int w = 0; w < FW_WAVES; ++w) sq += part64[w * NSAMP + tid];
        asq[tid] = (float)((double)variance - sq);                // dvar = sigma^2 - |a|^2, the cancellation in float64 (epilogue (i))
        asq[NSAMP + tid] = 0.f;
    }
}

template <int NS, bool S16, bool BIG, int LEAN_MODE = 0, bool F64 = false>
__global__ __launch_bounds__(FW_THREADS) void k_dgp_forward(const FwArgs gk) {
    static_assert(!F64 || (!S16 && BIG && LEAN_MODE == 0), "the float64 stage-1 variants: fp32 stage 2, every solve form, no compiled-in shapes");
    constexpr int NSAMP = 16 * NS;
    constexpr bool SHP = LEAN_MODE != 0;     // the headline stack's shapes and sources compiled in (all RBF, M = 128, D <= 10, operands staged, encoders
                                             // precomputed, noise drawn here, whole chunks): mode 1 and mode 2
    constexpr bool LEAN = LEAN_MODE == 1;    // ... and the bound's own evaluation: no per-layer outputs, the packed arrival, the half-wave tail.
                                             // Mode 2 keeps the outputs and the general tail: the forward of a value + gradient evaluation
    int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int gq = lane >> 4, jq = lane & 15;
    // Round 6 (BIG variants): the thread index becomes opaque again at every phase boundary, so that the lane-dependent tile addresses of a
    // phase are formed IN that phase (a handful of integer instructions) instead of once, in front of the layer loop.  Hoisted, they were ~80
    // registers live across every phase of every layer; the super-block solve and stage 2 then had to spill 12-14 of them around
    // themselves (configs[3] / [4]: 60 / 52 B of scratch per lane = 65 / 424 MB of spill stores per launch).
#define FW_REBASE() do { if constexpr (BIG || REBASE_ALL) { asm volatile("" : "+v"(tid)); lane = tid & 63; gq = lane >> 4; jq = lane & 15; } } while (0)
    const FwHead& g = gk.h;                                       // scalar path (first kernarg lines)
    const int XSTR = g.xstr;
    float* sm = reinterpret_cast<float*>(fw_smem);
    {
        // the header and the hot layer descriptors, FIRST thing and waited for at once: the header's fields are read all over
        // the prologue, and the first reads of lines that had not arrived yet were cold round trips one behind the other (the branch on
        // n_early alone cost one) -- one cold round trip for everything, scalar-cache hits from then on
        unsigned warm = 0;
        const unsigned* kw = reinterpret_cast<const unsigned*>(&gk);
#pragma unroll
        for (int i = 0; i < (int)((sizeof(FwHead) + 63) / 64); ++i) warm |= kw[16 * i];
        const unsigned* hw = reinterpret_cast<const unsigned*>(&gk.H[0]);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(FwHot) * IWVI_MAX_STACK / 64); ++i) warm |= hw[16 * i];
        asm volatile("" :: "s"(warm));
    }
    const int chunk_id = (int)blockIdx.x;
    const long long t0 = (long long)chunk_id * NSAMP;
    const int nvalid = SHP ? NSAMP : (int)((g.T - t0) < (long long)NSAMP ? (g.T - t0) : (long long)NSAMP);   // (LEAN: T is a multiple of the chunk)

    // ---- the last n_early waves (those that draw no noise below) issue every copy of the prologue; the others fetch the layer table and the
    //      chunk's rows (below: "the prologue's copies")
    const int n_early = g.n_early;
    const bool early_wave = wave >= FW_WAVES - n_early;
    const int ethreads = (FW_WAVES - n_early) * 64;               // threads that run the small loads in front of the first barrier
    // ---- layer table: kernarg -> LDS, every dword in flight at once ------------------------------------
    if (!early_wave) {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t* ka = (const uint32_t*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(FwArgs, L) / 4;
#else
        const uint32_t* ka = nullptr;
#endif
        uint32_t* dst = reinterpret_cast<uint32_t*>(sm + g.lds.ltab);
        // layers | noise plan | copy list are contiguous in FwArgs; copy only what is in use
        const int nl = g.n_layers * (int)(sizeof(FwLayer) / 4);
        for (int i = tid; i < nl; i += ethreads) dst[i] = ka[i];
        constexpr int offN = (int)((offsetof(FwArgs, N) - offsetof(FwArgs, L)) / 4), offC = (int)((offsetof(FwArgs, C) - offsetof(FwArgs, L)) / 4);
        const int nn = g.n_layers * (int)(sizeof(FwNoise) / 4), nc = g.ncopy * (int)(sizeof(FwCopy) / 4);
        for (int i = tid; i < nn; i += ethreads) dst[offN + i] = ka[offN + i];
        for (int i = tid; i < nc; i += ethreads) dst[offC + i] = ka[offC + i];
    }
    const FwLayer* LT = reinterpret_cast<const FwLayer*>(sm + g.lds.ltab);
    const FwNoise* NT = reinterpret_cast<const FwNoise*>(sm + g.lds.ltab + (offsetof(FwArgs, N) - offsetof(FwArgs, L)) / 4);
    const FwCopy* CT = reinterpret_cast<const FwCopy*>(sm + g.lds.ltab + (offsetof(FwArgs, C) - offsetof(FwArgs, L)) / 4);
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
    float* znoise = sm + g.lds.znoise;
    float* xyrows = sm + g.lds.xyrows;
    float* scratch = sm + g.lds.scratch;

    // (through the scalar cache: a vector load here would make every wave wait for vmcnt(0), the early waves for their copies; the counter
    //  only moves when the last workgroup of a launch arrives, after every workgroup of that launch has read it)
    const unsigned long long step = g.rng_state ? *((const __attribute__((address_space(4))) unsigned long long*)g.rng_state) : 0ULL;
    FW_STAMP(0);
    if (!SHP && g.dbg_exit == 1) return;

    // ================= prologue: everything small -> LDS, all loads in flight at once ==================
    const unsigned ut0 = (unsigned)t0, uT = (unsigned)g.T;
    const unsigned p_first = ut0 / g.row_div;
    const unsigned p_last = (ut0 + nvalid - 1) / g.row_div;
    const int npts = (int)(p_last - p_first) + 1;                 // distinct data points in this chunk
    // data row of sample j of the chunk: ((t0 + j) / row_div) % row_mod without a per-lane division: the uniform parts
    // once, the small remainders through a reciprocal (row_div, row_mod < 2^31; j < NSAMP)
    const unsigned rem0 = ut0 - p_first * (unsigned)g.row_div;    // (t0 + j) / row_div = p_first + (rem0 + j) / row_div
    const unsigned pf_mod = p_first % (unsigned)g.row_mod;
    const float rcp_div = 1.0f / (float)g.row_div;
    const bool small_div = g.row_div < (1 << 18), wide_mod = g.row_mod >= NSAMP;
    auto point_of = [&](int j) -> unsigned {                      // index of sample j's data point relative to p_first
        const unsigned tj = ut0 + j < uT ? (unsigned)j : (uT - 1 - ut0);
        return small_div ? (unsigned)div_small((int)(rem0 + tj), rcp_div) : (rem0 + tj) / (unsigned)g.row_div;
    };
    auto row_of = [&](unsigned dp) -> unsigned {                  // data row of point p_first + dp
        unsigned r = pf_mod + dp;
        if (wide_mod) { if (r >= (unsigned)g.row_mod) r -= (unsigned)g.row_mod; } else r %= (unsigned)g.row_mod;
        return r;
    };
    // ---- the prologue's big copies (the first solve stream: 36 KiB at M = 128; every layer's Z~ image: 8 KiB each), first thing, by the waves
    //      that draw no noise.  These operands were written by the launch before this one: every XCD's first read of a line goes past its L2,
    //      and a CU takes in ~25 KB per us of such copies -- issued behind the first barrier by every wave, between its table reads and its
    //      draws, they were 2.5 us of this kernel.  Now they land while the other waves fetch the table and draw.  The issuing waves touch no
    //      LDS in front of the first barrier (the compiler would make them wait for the copies there) and wait at their first table read behind
    //      it.  (Measured and rejected: the small copies and the row gathers issued here as well, entries read from the kernel arguments
    //      -- these waves then reach the first barrier at 2.0-2.3 us instead of 1.0, in either order; the small copies given to the drawing
    //      waves -- in front of their draws they queue behind the big ones in the CU's address unit, 0.6 us per wave; behind their draws: no change.)
    if (early_wave) {
        const int dw = FW_WAVES - 1 - wave, dth = n_early * 64;
        if (g.ls_first >= 0) {
            const FwHot& G0 = gk.H[g.ls_first];
            const float* lsrc = reinterpret_cast<const float*>(G0.LsP);
            float* ldst = sm + G0.ls_off;
            const int n4 = (tri_blocks(G0.nbk) * BLK16) >> 2;
            for (int i0 = dw * 64; i0 < n4; i0 += dth)
                if (i0 + lane < n4)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(lsrc + 4 * (i0 + lane)),
                                                     (__attribute__((address_space(3))) void*)(ldst + 4 * i0), 16, 0, 0);
        }
        for (unsigned zm = g.zt_mask; zm; zm &= zm - 1) {
            const FwHot& Hz = gk.H[__builtin_ctz(zm)];
            const float* zsrc = Hz.ZtP;
            float* zdst = sm + Hz.zt_off;
            const int n4 = Hz.nbk * Hz.nsteps * 16;
            for (int i0 = dw * 64; i0 < n4; i0 += dth)
                if (i0 + lane < n4)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(zsrc + 4 * (i0 + lane)),
                                                     (__attribute__((address_space(3))) void*)(zdst + 4 * i0), 16, 0, 0);
        }
    }
    if (!early_wave) {
    if (tid < NSAMP) {
        const unsigned dp = point_of(tid);
        rowi[tid] = (int)row_of(dp);
        pidx[tid] = (int)dp;
        lw[tid] = (!SHP && g.lw_init && tid < nvalid) ? g.lw_init[t0 + tid] : 0.f;
    }
    // the chunk's rows of X (models.py:113 / :50 tiling done here) and of the encoder input
    for (int idx = tid; idx < NSAMP * g.Dx; idx += ethreads) {
        const int d = idx / NSAMP, j = idx - d * NSAMP;            // (compile-time divisor)
        const unsigned row = (!SHP && g.x_per_sample) ? ut0 + (unsigned)(j < nvalid ? j : nvalid - 1) : row_of(point_of(j));
        xin[j * XSTR + d] = (j < nvalid) ? g.X[(size_t)row * g.Dx + d] : 0.f;
    }
    if (g.XY) {
        const int xs = up4(g.XYdim);
        const float rcp = 1.0f / (float)g.XYdim;
        for (int idx = tid; idx < npts * g.XYdim; idx += ethreads) {
            const int p = div_small(idx, rcp), i = idx - p * g.XYdim;
            xyrows[p * xs + i] = g.XY[(size_t)row_of((unsigned)p) * g.XYdim + i];
        }
    }
    }
    // layer table (and rowi / pidx) visible.  Not __syncthreads(): its release half waits for vmcnt(0), i.e. for the early waves' copies --
    // this wave's LDS stores are what the others need, and those are done at lgkmcnt(0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    FW_STAMP(56);
    // ---- everything small -> LDS, the duties split over the waves: each of these steps is a short chain of dependent latencies (a table
    //      read, an address, a copy), and one wave running all of them one after the other was 3.4 us of this kernel.  The first
    //      `noise_drawn` threads draw the noise (one Philox item each), from wave 0 up; the copies go to the waves that draw nothing, from
    //      the last wave down (to every wave when fewer than two are free).  In a wave that does both, whatever reads LDS comes first and
    //      the copies follow back to back: the compiler puts `s_waitcnt vmcnt(0)` in front of every ds_read / ds_write that follows a
    //      global_load_lds, so a table entry read between two copies makes the second wait until the first has landed.
    const int ncopy0 = g.ncopy;
    const bool noise_any_src = !SHP && g.noise_any_src;
    const int draw_waves = noise_any_src ? FW_WAVES : ((g.noise_drawn + 63) >> 6 < FW_WAVES ? (g.noise_drawn + 63) >> 6 : FW_WAVES);
    const int ndma = n_early > 0 ? n_early : FW_WAVES;            // (the host has checked that the copy list fits those waves)
    const int dwave = FW_WAVES - 1 - wave;                        // the copies' wave index: 0 = the last wave
    const bool dma_wave = dwave < ndma, late_big = n_early == 0;  // (without early waves the big copies are issued here too, by every wave)
    const int dthreads = ndma * 64;
    FwCopy ce0[FW_COPY_PER_WAVE];                                 // this wave's entries of the copy list
    // noise of every layer -> znoise[z_off + r * NSAMP + j]: 4 normals per Philox call, the (layer, 4-component group, sample) items of all
    // layers laid end to end over the workgroup's threads.  A thread's first item is looked up now (table reads), drawn in registers after
    // this wave's copies are issued and stored last; further items (more than one per thread: rare) follow
    int it_li = -1, it_k = 0, it_dims = 0, it_zoff = 0, it_zero = 0;
    {
    if (wave < draw_waves && !noise_any_src) {                  // the plan from the header: no LDS round trip per layer
        int base = 0, cnts[IWVI_MAX_STACK], zoffs[IWVI_MAX_STACK], dms[IWVI_MAX_STACK];
#pragma unroll
        for (int li = 0; li < IWVI_MAX_STACK; ++li) { cnts[li] = g.nz_cnt[li]; zoffs[li] = g.nz_zoff[li]; dms[li] = g.nz_dims[li]; }   // (three wide scalar loads, one wait)
        const unsigned zmask = g.nz_zero_mask;
#pragma unroll
        for (int li = 0; li < IWVI_MAX_STACK; ++li) {              // (layers beyond the stack: 0 items)
            int k = tid - base;
            if (k < 0) k += FW_THREADS;
            if (it_li < 0 && k < cnts[li]) { it_li = li; it_k = k; it_dims = dms[li]; it_zoff = zoffs[li]; it_zero = (int)(zmask >> li & 1u); }
            base = (base + cnts[li]) & (FW_THREADS - 1);
        }
    } else if (wave < draw_waves) {
        int base = 0;                                              // first thread of the current layer's items
        for (int li = 0; li < g.n_layers; ++li) {
            const FwNoise nz = uniform_words(NT[li]);              // one LDS round trip per layer
            if (nz.src) {                                          // injected [T, dims]: a gather by LDS-DMA (the source is per lane)
                float* zdst = znoise + nz.z_off;
                for (int i0 = (tid & ~63); i0 < nz.dims * NSAMP; i0 += FW_THREADS) {
                    const int i = i0 + lane, r = i / NSAMP, j = i - r * NSAMP;
                    if (i < nz.dims * NSAMP && j < nvalid)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(nz.src + (size_t)(t0 + j) * nz.dims + r),
                                                         (__attribute__((address_space(3))) void*)(zdst + i0), 4, 0, 0);
                }
                continue;
            }
            const int cnt = ((nz.dims + 3) >> 2) * NSAMP;
            int k = tid - base;
            if (k < 0) k += FW_THREADS;
            if (it_li < 0 && k < cnt) { it_li = li; it_k = k; it_dims = nz.dims; it_zoff = nz.z_off; it_zero = nz.zero; }
            base = (base + cnt) & (FW_THREADS - 1);
        }
    }
    FW_STAMP(40);
    if (dma_wave) fw_copy_load(CT, sm, 0, ncopy0, dwave, lane, ce0, ndma);
    }
    // the chunk's targets y (a gather by LDS-DMA: no register, nothing waits here)
    if (g.out_logw && dma_wave) {
        float* yrows = sm + g.lds.yrows;
        const int Dy_ = g.Dy;
        for (int i0 = dwave * 64; i0 < NSAMP * Dy_; i0 += dthreads) {
            const int idx = i0 + lane, d = idx / NSAMP, j = idx - d * NSAMP;
            if (idx < NSAMP * Dy_ && j < nvalid)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.Y + (size_t)row_of(point_of(j)) * Dy_ + d),   // (= rowi[j], without the LDS read)
                                                 (__attribute__((address_space(3))) void*)(yrows + i0), 4, 0, 0);
        }
    }
    {
    if (dma_wave) {
    // precomputed encoder outputs of the chunk's distinct data points -> the LV layer's constant block
    for (unsigned pm = g.pre_enc_mask; pm; pm &= pm - 1) {         // (no walk over the layer table: each step of it is a scalar-cache round trip)
        const FwHot& Hp = gk.H[__builtin_ctz(pm)];
        const float* eo = reinterpret_cast<const float*>(Hp.LrTP);
        const int no = 2 * Hp.R;
        float* dst = sm + Hp.c_off;
        const float rcp = 1.0f / (float)no;
        for (int i0 = dwave * 64; i0 < npts * no; i0 += dthreads) {        // a gather by LDS-DMA: nothing waits here
            const int idx = i0 + lane;
            const int p = div_small(idx < npts * no ? idx : 0, rcp), o = idx - p * no;
            if (idx < npts * no)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(eo + (size_t)row_of((unsigned)p) * no + o),
                                                 (__attribute__((address_space(3))) void*)(dst + i0), 4, 0, 0);
        }
    }
    fw_copy_issue(sm, lane, ce0);                                 // copy list: all DMA loads in flight together
    if (g.ls_first >= 0 && late_big) {                             // the largest piece (36 KiB at M = 128), 1 KiB per wave-instruction
        const FwHot& G0 = gk.H[g.ls_first];
        const float* lsrc = reinterpret_cast<const float*>(G0.LsP);
        float* ldst = sm + G0.ls_off;
        const int n4 = (tri_blocks(G0.nbk) * BLK16) >> 2;
        for (int i0 = dwave * 64; i0 < n4; i0 += dthreads) {
            const int i = i0 + lane;
            if (i < n4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(lsrc + 4 * i),
                                                 (__attribute__((address_space(3))) void*)(ldst + 4 * i0), 16, 0, 0);
        }
    }
    }
    if (wave < draw_waves) {
        const int q0 = it_k / NSAMP, j0 = it_k - q0 * NSAMP;
        float v0[4] = {0.f, 0.f, 0.f, 0.f};
        if (it_li >= 0 && !it_zero && j0 < nvalid) draw_normal4(g.seed, step, g.layer_base + it_li, t0 + j0, q0, v0);
        if (it_li >= 0) {
            float* zdst = znoise + it_zoff;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q0 + e < it_dims) zdst[(4 * q0 + e) * NSAMP + j0] = v0[e];
        }
        if (g.noise_drawn > FW_THREADS) {                          // more items than threads: the rest, as they come
            int base = 0;
            for (int li = 0; li < g.n_layers; ++li) {
                const FwNoise nz = uniform_words(NT[li]);
                if (nz.src) continue;
                const int dims = nz.dims;
                float* zdst = znoise + nz.z_off;
                const int cnt = ((dims + 3) >> 2) * NSAMP;
                int k = tid - base;
                if (k < 0) k += FW_THREADS;
                for (; k < cnt; k += FW_THREADS) {
                    if (li == it_li && k == it_k) continue;
                    const int q = k / NSAMP, j = k - q * NSAMP;
                    float v[4] = {0.f, 0.f, 0.f, 0.f};
                    if (!nz.zero && j < nvalid) draw_normal4(g.seed, step, g.layer_base + li, t0 + j, q, v);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * q + e < dims) zdst[(4 * q + e) * NSAMP + j] = v[e];
                }
                base = (base + cnt) & (FW_THREADS - 1);
            }
        }
    }
    FW_STAMP(57);
    }
    FW_STAMP(58);
    // the launch's device scalars -> LDS (likelihood variance: read in the tail; kernel variances: read at the top of a layer -- a global round
    // trip has nothing to hide behind in either place): one per lane of the last wave, whose copies are all issued.  (A scalar load would sit
    // in front of every later lgkmcnt(0) of its wave for its whole round trip; requested at the top of the kernel instead, the address
    // arithmetic kept this wave from the first barrier: +0.3 us for everybody.)
    if (wave == FW_WAVES - 1) {
        if (lane == 63) { float lv = g.lik_variance; if (g.lik_var_dev) lv = *((gptr1)g.lik_var_dev); sm[g.lds.cnt + 8] = lv; }
        if (lane >= 48 && lane < 48 + IWVI_MAX_STACK && (g.var_dev_mask >> (lane - 48) & 1u)) sm[g.lds.cnt + 12 + (lane - 48)] = *((gptr1)g.var_dev[lane - 48]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FW_STAMP(59);
    __syncthreads();
    FW_STAMP(1);
    if (!SHP && g.dbg_exit == 2) return;

    int xt_for = -1;                                              // layer whose Gram operand x~ is already in `xt`
    for (int li = 0; li < g.n_layers; ++li) {
        FW_REBASE();
        const FwLayer& L = LT[li];
        // the layer's 32 hot words: ONE pair of wide scalar loads from the (warmed) kernel-argument lines, then opaque registers.  Read through
        // a reference, every field was re-loaded next to each use (the compiler rematerialises loads of the kernel arguments instead of
        // keeping them) -- nine dependent scalar-cache round trips at the top of a layer, ~0.5 us per layer boundary
        const FwHot Hv = opaque_block(gk.H[li]);
        const FwHot& H = Hv;
        const FwHot& G = H;
        const int D = H.D;
        const float* cst = sm + H.c_off;
        const float* zl = znoise + H.z_off;
        // optional outputs: from the LDS table, and only when the launch has any
        gout1 o_sample = nullptr, o_mean = nullptr, o_var = nullptr, o_noise = nullptr, o_gmv = nullptr, o_a = nullptr, o_u = nullptr, o_kl = nullptr;
        if (!LEAN && (H.flags & FWF_ANY_OUT)) {
            o_sample = (gout1)ufirst(L.sample); o_mean = (gout1)ufirst(L.mean); o_var = (gout1)ufirst(L.var);
            o_noise = (gout1)ufirst(L.noise_out);
            if (H.type == IWVI_LAYER_GP) { o_gmv = (gout1)ufirst(L.gmv_out); o_a = (gout1)ufirst(L.gp.a_out); o_u = (gout1)ufirst(L.gp.u_out); }
            else o_kl = (gout1)ufirst(L.lv.kl_local);
        }
        // a following GP layer gets its Gram operand from this layer's last phase (no phase of its own)
        const bool nx_gp = (H.flags & FWF_NX_GP) != 0, nx_rbf = SHP || (H.flags & FWF_NX_RBF) != 0;
        const float* nx_cst = sm + H.nx_c_off;
        const int nx_nsteps = SHP ? 3 : H.nx_nsteps;
        if (H.type == IWVI_LAYER_LV) {
            // ================= LatentVariableLayer (layers.py:72-105) =================================
            const FwLv& V = L.lv;
            const int Lw = H.R, Do = D + Lw, n_enc = H.nbk, sampled_kl = (H.flags & FWF_SAMPLED_KL) ? 1 : 0;
            const int mdim = up4(H.Mp);
            float* act0 = scratch;
            float* act1 = act0 + NSAMP * mdim;
            const float* in = xyrows; int in_str = up4(g.XYdim);
            float* out = act0;
            const bool pre_enc = SHP || (H.flags & FWF_PRE_ENC) != 0;    // encoder already evaluated by iwvi_model_precompute
            const int enc_actv = pre_enc ? 0 : ufirst(V.act);
            if (pre_enc) { in = cst; in_str = 2 * Lw; }
            else {
                int off = 0;
                for (int l = 0; l < n_enc; ++l) {                                    // encoder, once per data point
                    const int din = ufirst(V.dims[l]), dout = ufirst(V.dims[l + 1]);
                    const float* W = cst + off; const float* b = W + din * dout;
                    for (int idx = tid; idx < npts * dout; idx += FW_THREADS) {
                        const int p = idx / dout, o = idx - p * dout;
                        float acc = b[o];
#pragma unroll 8
                        for (int i = 0; i < din; ++i) acc = fmaf(in[p * in_str + i], W[i * dout + o], acc);
                        if (l < n_enc - 1) acc = enc_act(acc, enc_actv);             // layers.py:143-144
                        if (din == dout) acc += in[p * in_str + o];                  // layers.py:146-147
                        out[p * mdim + o] = acc;
                    }
                    off += din * dout + dout;
                    __syncthreads();
                    in = out; in_str = mdim;
                    out = (out == act0) ? act1 : act0;
                }
            }
            FW_STAMP(2 + li * 6 + 1);
            // `in` rows hold [means (Lw) | raw (Lw)] per distinct point
            for (int idx = tid; idx < NSAMP * D; idx += FW_THREADS) {
                const int c = idx / NSAMP, j = idx - c * NSAMP;    // column-major: the divisor is a compile-time constant
                const float v = xin[j * XSTR + c];
                xout[j * XSTR + c] = v;
                if (j < nvalid) {
                    const long long t = t0 + j;
                    if (o_sample) o_sample[t * Do + c] = v;
                    if (o_mean) o_mean[t * Do + c] = v;
                    if (o_var) o_var[t * Do + c] = 0.f;
                }
            }
            if (wave < NS) {                                       // wave t: sub-tile t; its lanes gq == 0: one sample each
                const int j = 16 * wave + jq;
                if (gq == 0) {
                    const long long t = t0 + j;
                    float klsum = 0.f;
                    for (int l = 0; l < Lw; ++l) {
                        float mu = 0.f, sg = 1.f;                                        // prior (layers.py:73-81)
                        if (n_enc > 0 || pre_enc) { mu = in[pidx[j] * in_str + l]; sg = softplus_f(in[pidx[j] * in_str + Lw + l] - 3.f); }
                        const float z = (j < nvalid) ? zl[l * NSAMP + j] : 0.f;
                        const float w = fmaf(z, sg, mu);                                 // layers.py:86-87
                        float kl;
                        if (sampled_kl) kl = -0.5f * z * z - __logf(sg) + 0.5f * w * w;  // log q(W) - log p(W), :98-100
                        else kl = 0.5f * (sg * sg + mu * mu - 1.f) - __logf(sg);         // KL(N(mu,sg)||N(0,1)), :101-103
                        klsum += kl;
                        xout[j * XSTR + D + l] = w;
                        xin[j * XSTR + D + l] = w;                                       // completes row j for x~ below
                        if (j < nvalid) {
                            if (o_kl) o_kl[t * Lw + l] = kl;
                            if (o_noise) o_noise[t * Lw + l] = z;
                            if (o_sample) o_sample[t * Do + D + l] = w;                  // layers.py:89-91
                            if (o_mean) o_mean[t * Do + D + l] = mu;
                            if (o_var) o_var[t * Do + D + l] = sg * sg;
                        }
                    }
                    lw[j] += klsum;
                }
                // (LDS operations of one wave complete in order: the row written above is what is read here)
                if (nx_gp) xt_subtile(xin + j * XSTR, xt + j * XSTR, nx_cst, nx_cst + 32, Do, nx_nsteps, nx_rbf, gq);
            }
            if (nx_gp) xt_for = li + 1;
            __syncthreads();
            FW_STAMP(2 + li * 6 + 5);
        } else {
            // ================= GPLayer (layers.py:35-50) ==============================================
            const int nbk = SHP ? 8 : G.nbk, R = G.R, P = G.P, nsteps = SHP ? 3 : G.nsteps;   // (LEAN: M = 128, D <= 10 -- see launch)
            const unsigned long long s2_rec = L.gp.s2w[wave];    // stage 2's run of this wave (used behind the Gram: the read is long back by then)
            float g_variance = G.variance;
            if (g.var_dev_mask >> li & 1u) g_variance = sm[g.lds.cnt + 12 + li];          // (a device scalar: fetched in the prologue)
            f32x4* kuf = reinterpret_cast<f32x4*>(scratch);
            f32x4* at = kuf;                                       // solved in place (stage 1)
            float* usq = scratch + (size_t)G.Mp * NSAMP;           // [wave][r][NSAMP]
            const bool rbf = SHP || G.kern_type == IWVI_KERN_RBF;
            // split-f16 solve (even nbk <= 8; see split_b16): the Gram tile in units of U (1 otherwise), a_bj scaled by sb for the updates
            const float st1_u = cst[IWVI_CST_U], st1_sb = cst[IWVI_CST_SB];
            const float* invls = cst; const float* zc = cst + 32;
            const float* Wm = cst + gpc_W(); const float* mfA = cst + gpc_A(P, R); const float* mfb = cst + gpc_b(D, P, R);
            // Gram form: the expanded |x|^2 + |z|^2 - 2 x.z (one small MFMA product) has an absolute error of
            // ~eps * (|x|^2 + |z|^2); use it only while the inducing cloud is compact in lengthscale units,
            // else difference the coordinates directly (error ~eps * r^2)
            const bool gram_mfma = __float_as_int(cst[64]) <= __float_as_int(4.0f);   // zmax2 >= 0: int compare is exact

            // ---- x~ (only when the layer before did not already leave it in `xt`) -------------------------
            if (xt_for != li) {
                if (wave < NS) xt_subtile(xin + (16 * wave + jq) * XSTR, xt + (16 * wave + jq) * XSTR, invls, zc, D, nsteps, rbf, gq);
                __syncthreads();
            }
            // (the layer's forward-substitution stream is already on its way to LDS: prologue for the first GP
            // layer, the previous GP layer's stage 2 for the others)
            FW_STAMP(2 + li * 6 + 0);
            FW_REBASE();
            // ---- Gram: kuf block bi = kernel(Z_bi, x), written in B-operand order ---------------------------
            DBG_WSTAMP(32);
            // float64 stage-1 route (F64 variants, layers flagged IWVI_LAYER_F64_STAGE1): the Gram tile in float64, behind the |u|^2 slots
            bool f64_l = false;
            double* k64 = nullptr; double* part64 = nullptr;
            const double* f64_Linv = nullptr;
            if constexpr (F64) {
                f64_l = (H.flags & FWF_F64) != 0;
                if (f64_l) {
                    k64 = reinterpret_cast<double*>(usq + (size_t)FW_WAVES * R * NSAMP);
                    part64 = k64 + (size_t)G.Mp * NSAMP;
                    const StateLayout sl = state_layout(G.M, R);
                    const char* st = reinterpret_cast<const char*>(G.LsP) - sl.off_LsP;
                    f64_Linv = reinterpret_cast<const double*>(st + sl.off_Linv);
                    f64_gram<NS>(G, reinterpret_cast<const float*>(st + sl.off_Zs), xt, XSTR, D, g_variance, k64, wave, gq, jq);
                    f32x4* uz = reinterpret_cast<f32x4*>(usq);      // (stage 2 fills only some of the |u|^2 slots)
                    for (int i = tid; i < (FW_WAVES * R * NSAMP) / 4; i += FW_THREADS) uz[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            // Z~ from L2 (the large stacks: five layers' images do not fit the LDS beside a 96-KiB tile): every wave reaches this phase behind the
            // same barrier, so a load in front of its MFMAs is a round trip all eight wait for together -- and the k-step loop had one
            // per step (run-time trip count: load, wait, NS MFMAs; 12 round trips per wave at M = 512 = most of the phase's 5.2 us).
            // Round 5: the first GZP k-steps of the NEXT block row are requested while this one's MFMAs issue.
            constexpr int GZP = 3;                                // (D <= 10; the steps beyond are loaded in place)
            float znext[GZP] = {0.f, 0.f, 0.f};
            // (only in the variants of at most three sub-tiles -- the ones whose Z~ images do not fit: at NS = 5 the images are staged, and
            //  the extra live registers cost configs[3] 1.4 %)
            constexpr bool GZ_ON = BIG && NS <= 3;
            const bool z_l2 = GZ_ON && !SHP && gram_mfma && G.zt_off < 0 && !f64_l;
            if (GZ_ON && z_l2 && wave < nbk) {
                gptr1 zq = (gptr1)G.ZtP + (size_t)wave * nsteps * 64 + lane;
#pragma unroll
                for (int u = 0; u < GZP; ++u) znext[u] = zq[(u < nsteps ? u : nsteps - 1) * 64];
            }
            for (int bi = wave; bi < (f64_l ? 0 : nbk); bi += FW_WAVES) {
                f32x4 acc[NS];
#pragma unroll
                for (int t = 0; t < NS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (GZ_ON && z_l2) {
                    float zc[GZP];
#pragma unroll
                    for (int u = 0; u < GZP; ++u) zc[u] = znext[u];
                    FW_PIN_LOADS();
                    {
                        const int bn = bi + FW_WAVES < nbk ? bi + FW_WAVES : bi;
                        gptr1 zq = (gptr1)G.ZtP + (size_t)bn * nsteps * 64 + lane;
#pragma unroll
                        for (int u = 0; u < GZP; ++u) znext[u] = zq[(u < nsteps ? u : nsteps - 1) * 64];
                    }
                    FW_PIN_LOADS();
#pragma unroll
                    for (int u = 0; u < GZP; ++u) {
                        if (u < nsteps) {
#pragma unroll
                            for (int t = 0; t < NS; ++t)
                                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(zc[u], xt[(16 * t + jq) * XSTR + 4 * u + gq], acc[t], 0, 0, 0);
                        }
                    }
                    gptr1 zp = (gptr1)G.ZtP + (size_t)bi * nsteps * 64 + lane;
                    for (int s = GZP; s < nsteps; ++s) {
                        const float a = zp[s * 64];
#pragma unroll
                        for (int t = 0; t < NS; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xt[(16 * t + jq) * XSTR + 4 * s + gq], acc[t], 0, 0, 0);
                    }
                } else
                if (gram_mfma) {
                    if (SHP || G.zt_off >= 0) {
                        const float* zp = sm + G.zt_off + (size_t)bi * nsteps * 64 + lane;      // staged in LDS
                        for (int s = 0; s < nsteps; ++s) {
                            const float a = zp[s * 64];
#pragma unroll
                            for (int t = 0; t < NS; ++t)
                                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xt[(16 * t + jq) * XSTR + 4 * s + gq], acc[t], 0, 0, 0);
                        }
                    } else {
                        gptr1 zp = (gptr1)G.ZtP + (size_t)bi * nsteps * 64 + lane;              // from L2
                        for (int s = 0; s < nsteps; ++s) {
                            const float a = zp[s * 64];
#pragma unroll
                            for (int t = 0; t < NS; ++t)
                                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xt[(16 * t + jq) * XSTR + 4 * s + gq], acc[t], 0, 0, 0);
                        }
                    }
                } else {
                    // direct form: Z~ holds c*zs (RBF, c = log2 e) or -2*zs (Matern52) for rows 16bi+4g .. +3 of
                    // feature d at [(bi*nsteps + d/4)*64 + 16*(d%4) + 4g .. +3]; acc becomes what the MFMA form yields
                    const float sx = rbf ? 1.4426950408889634f : -2.f;
                    for (int d = 0; d < D; ++d) {
                        const size_t zi = ((size_t)bi * nsteps + (d >> 2)) * 64 + 16 * (d & 3) + 4 * gq;
                        const f32x4 z4 = (SHP || G.zt_off >= 0) ? *reinterpret_cast<const f32x4*>(sm + G.zt_off + zi)
                                                         : *((gptr4)(G.ZtP + zi));
#pragma unroll
                        for (int t = 0; t < NS; ++t) {
                            const float xv = sx * xt[(16 * t + jq) * XSTR + d];
#pragma unroll
                            for (int e = 0; e < 4; ++e) { const float df = xv - z4[e]; acc[t][e] = fmaf(df, df, acc[t][e]); }
                        }
                    }
                    // RBF: sum = c^2 r^2 -> exponent -r^2 c / 2 + log2 var;  Matern52: sum = 4 r^2 -> r^2
                    const float sc = rbf ? -0.5f / 1.4426950408889634f : 0.25f;
                    const float of = rbf ? __log2f(g_variance) : 0.f;
#pragma unroll
                    for (int t = 0; t < NS; ++t)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[t][e] = fmaf(acc[t][e], sc, of);
                }
                // padded inducing rows (M not a multiple of 16) must give k = 0: the forward substitution carries them
                const int mrow = 16 * bi + 4 * gq;
                if (rbf) {
#pragma unroll
                    for (int t = 0; t < NS; ++t) {
                        f32x4 k;
#pragma unroll
                        for (int e = 0; e < 4; ++e) k[e] = (SHP || mrow + e < G.M) ? __builtin_amdgcn_exp2f(acc[t][e]) : 0.f;   // log2(var) folded in
                        k *= st1_u;
                        kuf[(bi * 4 + gq) * NSAMP + 16 * t + jq] = k;
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < NS; ++t) {
                        f32x4 k;
#pragma unroll
                        for (int e = 0; e < 4; ++e) k[e] = (mrow + e < G.M) ? kern_from_acc(acc[t][e], G.kern_type, g_variance) : 0.f;
                        k *= st1_u;
                        kuf[(bi * 4 + gq) * NSAMP + 16 * t + jq] = k;
                    }
                }
            }
            DBG_WSTAMP(33);
            // stage 2's first operands are requested here, in the shadow of the Gram phase's barrier (nothing in them
            // depends on the Gram or the solve), so that its MFMAs start right behind the barrier that ends stage 1
            // (this wave's run: the record read at the top of the layer -- runs are dealt to waves by load, not in order)
            const unsigned s2_lo = (unsigned)ufirst((int)(unsigned)s2_rec), s2_hi = (unsigned)ufirst((int)(unsigned)(s2_rec >> 32));
            const int s2_nblocks = (int)(s2_hi & 0xffffu), s2_r0 = (int)(s2_hi >> 16 & 0xffu), s2_bi0 = (int)(s2_hi >> 24 & 63u);
            const int s2_mw0 = (s2_hi >> 30 & 1u) ? wave : -1, s2_mw1 = (s2_hi >> 31) ? wave : -1;
            constexpr bool s16 = S16;
            gptr4 s2_P = (gptr4)G.LrTP + (size_t)s2_lo + lane;
            // the NEXT GP layer's forward-substitution stream is fetched into registers now and parked in the staging
            // buffer once this layer's solve is over (the buffer is busy until then; an LDS-DMA left pending across
            // stage 2 would make every LDS read there wait for all outstanding loads)
            constexpr int LSN = (36 * 64 + FW_THREADS - 1) / FW_THREADS;   // tri_blocks(8) packed blocks of 64 float4
            f32x4 lsn[LSN];
            const int lsn4 = H.nx_ls_off >= 0 ? H.nx_ls_n >> 2 : 0;
#pragma unroll
            for (int i = 0; i < LSN; ++i) {
                lsn[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (tid + i * FW_THREADS < lsn4) {
                    lsn[i] = ((gptr4)H.nx_ls)[tid + i * FW_THREADS];
                }
            }
            f32x4 ring[4];
            ring[0] = ring[1] = ring[2] = ring[3] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 ring2e[2];                                      // split-f16: second plane of the two slabs requested ahead of stage 1
            ring2e[0] = ring2e[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            // split-f16, M <= 128: the q_mu^T slabs, requested during stage 1 by a wave that is idle there.  (Not in the BIG variants: eight more
            // registers live across the super-block solve cost configs[3] a third of its stage 1 in spills -- 28100 -> 37500 clocks.)
            f32x4 Qpre[BIG ? 1 : 4][2];
            bool q_pre = false;
#pragma unroll
            for (int u = 0; u < (BIG ? 1 : 4); ++u) Qpre[u][0] = Qpre[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (s2_nblocks > 0 && s16) {
                const size_t o1 = (size_t)(1 < s2_nblocks ? 1 : s2_nblocks - 1) * 128;
                ring[0] = s2_P[0]; ring2e[0] = s2_P[64]; ring[1] = s2_P[o1]; ring2e[1] = s2_P[o1 + 64];
            }
            if (s2_nblocks > 0 && !s16) {
                ring[0] = s2_P[0];
                ring[1] = s2_P[(size_t)(1 < s2_nblocks ? 1 : s2_nblocks - 1) * 64];
                ring[2] = s2_P[(size_t)(2 < s2_nblocks ? 2 : s2_nblocks - 1) * 64];
            }
            DBG_WSTAMP(34);
            __syncthreads();                                      // (the staged solve stream was complete before this layer began)
            DBG_WSTAMP(35);
            FW_STAMP(2 + li * 6 + 1);
            FW_REBASE();

            // ---- stage 1: a = Lm^-1 k, right-looking blocked forward substitution, one wave per sub-tile ------
            // packed stream, column bj: [Dinv(bj), -L(bj+1,bj) .. -L(nbk-1,bj)];  a_bj = Dinv_bj r_bj, then
            // r_bi += (-L(bi,bj)) a_bj for every bi > bj: independent MFMA chains, B operand = a_bj in registers.
            // r lives in the `at` tile (first touched from the Gram tile); the freshly updated r_{bj+1} is handed
            // to the next column in registers, so the dependent chain never waits for LDS.
            // (five sub-tiles on four SIMDs: two solves share SIMD 0.  Cutting the fifth solve into a 2 x 2 block system over four waves paid
            // while its updates were fp32 MFMAs; with the split-f16 updates the hand-offs cost what the cut saves: measured equal, removed.)
            if (F64 && f64_l) {
                if constexpr (F64)
                    f64_solve<NS>(G, f64_Linv, k64, at, part64, asq, g_variance, o_a, t0, nvalid, tid, wave, gq, jq);
            } else
            if (BIG && nbk >= FW_SB_MIN_NBK) {
                // ---- M >= 256: super-block solve.  Per super-block I (8 block rows): r_I = k_I - L(I, <I) a_<I (dense product,
                // one block row per wave, in place), then a_I = (L_II)^-1 r_I (triangular product with the packed inverse of
                // the 128 x 128 diagonal super-block; every wave reads r_I, barrier, writes a_I in place).  No wave carries a
                // dependent chain, all eight are busy; the operand stream (csrc/precompute.hip) comes from L2 in job order.
                // S16 launches (round 4): the dense product runs on split-f16 operands like stage 2 -- the slabs of 2^ea (-L(bi, <8I)) from
                // the state (k_pack_ls16), the solved a_<I read as the f16 planes stage 2 reads -- three v_mfma_f32_16x16x32_f16 (16 clocks
                // each) per 16 x 32 slab instead of eight fp32 MFMAs of 32: 47 % of this stage's blocks at M = 256, 73 % at M = 512 (the
                // products with the inverse diagonal super-blocks stay fp32: an error there is amplified by cond(L_II)).  a_I is written
                // straight as its two f16 planes (each block's planes live in the block's own four tile rows), so stage 2 converts nothing.
                {
                    f32x4* uz = reinterpret_cast<f32x4*>(usq);
                    for (int i = tid; i < (FW_WAVES * R * NSAMP) / 4; i += FW_THREADS) uz[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                gptr4 Ap = (gptr4)G.LsP + lane;
                const int nsb = (nbk + 7) >> 3;
                // operands in flight per wave (registers: 8 / 4 per slab / block): the five-sub-tile variants have none to spare
                // (kernel_resources.py: a spill there cost configs[3] a third of this stage in round 4)
                constexpr int SBD = dbgSBD > 0 ? dbgSBD : (NS <= 3 ? 4 : 2), SBT = dbgSBT > 0 ? dbgSBT : (NS <= 3 ? 4 : 2);
                float ssq[NS];
#pragma unroll
                for (int t = 0; t < NS; ++t) ssq[t] = 0.f;
                int off = 0, off16 = 0, offt = 0;
                // block row of a wave within a super-block: waves w and w + 4 share a SIMD, and row q of the inverse part costs q + 1 blocks --
                // rows (i, 7 - i) per SIMD level it (9 blocks each instead of 6 / 8 / 10 / 12)
                const int rw = wave < 4 ? wave : 11 - wave;
                const float sa_sb = cst[IWVI_CST_SA];
                for (int I = 0; I < nsb; ++I) {
                    const int r0 = 8 * I, nr = (nbk - r0 < 8) ? nbk - r0 : 8;
                    const bool mine = rw < nr;
                    const int bi = r0 + rw;
                    f32x4 acc[NS];
                    DBG_WSTAMP(48 + 6 * I);
                    f32x4 Dg = {0.f, 0.f, 0.f, 0.f};              // S16: the row's own diagonal block of (L_II)^-1 (fp32), asked for ahead of the dense part
                    if constexpr (S16) {
                        if (mine) { Dg = Ap[(size_t)(off + nr * r0 + rw * (rw + 1) / 2 + rw) * 64]; FW_PIN_LOADS(); }
                    }
                    if (S16 && mine && r0 > 0) {
                        using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
                        const int nst = r0 >> 1;                  // slabs of this block row
                        gptr4 P16 = (gptr4)G.LsP + G.ls16_off + (size_t)(off16 + rw * nst) * 128 + lane;
                        const f32x4* p1 = at + (size_t)(2 * gq) * NSAMP + jq;   // h1 vector of chunk kc, sub-tile t: p1[kc * 8 * NSAMP + 16 t]; h2: the next row
                        const f32x4* p2 = p1 + NSAMP;
#pragma unroll
                        for (int t = 0; t < NS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                        // the slab stream comes straight from L2: SBD slabs in flight (round 5; one until then: VERDICT r04 weak #4).  nst = 4 I is a multiple of 4.
                        f32x4 A1[SBD], A2[SBD];
#pragma unroll
                        for (int u = 0; u < SBD; ++u) { const size_t o_ = (size_t)(u < nst ? u : nst - 1) * 128; A1[u] = P16[o_]; A2[u] = P16[o_ + 64]; }
                        for (int kc0 = 0; kc0 < nst; kc0 += SBD) {
#pragma unroll
                            for (int u = 0; u < SBD; ++u) {
                                const int kc = kc0 + u;
                                const f16x8 a1 = __builtin_bit_cast(f16x8, A1[u]), a2 = __builtin_bit_cast(f16x8, A2[u]);
                                f32x4 b1[NS], b2[NS];
#pragma unroll
                                for (int t = 0; t < NS; ++t) { b1[t] = p1[kc * 8 * NSAMP + 16 * t]; b2[t] = p2[kc * 8 * NSAMP + 16 * t]; }
#pragma unroll
                                for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(f16x8, b1[t]), acc[t], 0, 0, 0);
#pragma unroll
                                for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, __builtin_bit_cast(f16x8, b1[t]), acc[t], 0, 0, 0);
#pragma unroll
                                for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(f16x8, b2[t]), acc[t], 0, 0, 0);
                                // the slot is refilled BEHIND the products that read it (its registers are the load's destination: no copies at
                                // the end of the unrolled group, which would wait for every load in flight), SBD - 1 steps ahead of its next use
                                FW_PIN_LOADS();
                                const size_t nx = (size_t)(kc + SBD < nst ? kc + SBD : nst - 1) * 128;
                                A1[u] = P16[nx]; A2[u] = P16[nx + 64];
                                FW_PIN_LOADS();
                            }
                        }
                        const float inv_u = 1.0f / (sa_sb * sa_sb);   // both operands carry 2^ea (exact powers of two)
#pragma unroll
                        for (int t = 0; t < NS; ++t) acc[t] = kuf[(bi * 4 + gq) * NSAMP + 16 * t + jq] + acc[t] * inv_u;      // r(bi), in registers
                    } else if (S16 && mine) {
#pragma unroll
                        for (int t = 0; t < NS; ++t) acc[t] = kuf[(bi * 4 + gq) * NSAMP + 16 * t + jq];                        // first super-block: r = k
                    } else if (mine && r0 > 0) {
#pragma unroll
                        for (int t = 0; t < NS; ++t) acc[t] = kuf[(bi * 4 + gq) * NSAMP + 16 * t + jq];
                        gptr4 P = Ap + (size_t)(off + rw * r0) * 64;
                        f32x4 a_nx = P[0], a_n2 = P[(size_t)(1 < r0 ? 1 : 0) * 64];
                        for (int bj = 0; bj < r0; ++bj) {
                            const f32x4 a_cur = a_nx;
                            a_nx = a_n2;
                            a_n2 = P[(size_t)(bj + 2 < r0 ? bj + 2 : r0 - 1) * 64];
                            f32x4 b[NS];
#pragma unroll
                            for (int t = 0; t < NS; ++t) b[t] = at[(bj * 4 + gq) * NSAMP + 16 * t + jq];
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
#pragma unroll
                                for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s], b[t][s], acc[t], 0, 0, 0);
                            }
                        }
#pragma unroll
                        for (int t = 0; t < NS; ++t) at[(bi * 4 + gq) * NSAMP + 16 * t + jq] = acc[t];      // r(bi), in place of k(bi)
                    }
                    // ---- a_I = (L_II)^-1 r_I.  S16 (round 5, last): only the DIAGONAL block of a row is an fp32 product -- on the row's own r(bi),
                    // still in registers, in front of the barrier --; the blocks left of it take split-f16 operands like the dense part: the
                    // packed image holds, per lane, [h1 x 4 | h2 x 4] of 2^lg (L_II)^-1 (k_pack_ls16; lg = ceil(log2 sigma)), every row
                    // publishes r(bi) as [h1 x 4 | h2 x 4] of s_r r (s_r = 2^(5 - 2 lg): |r| <= sigma^2 -> <= 32) in place of k(bi), and
                    // block (w, q) costs two v_mfma_f32_16x16x32_f16 per sub-tile (A = [h1 | h1], then [h2 | h2], against B = [h1' | h2']:
                    // all four partial products) -- 32 clocks where the fp32 block took 128; 28 of a super-block's 36 blocks.  An error in
                    // these products is amplified by cond(L_II) like one in r itself, which the split-f16 dense part already carries at
                    // the same 2^-22; layers whose K_uu is ill-conditioned take the float64 route (F64).
                    if constexpr (S16) {
                        f32x4 Ti[SBT];
                        using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
                        const float s_i = 1024.0f / sa_sb, s_r = 32.0f / (s_i * s_i);        // sa_sb = 2^(10 - lg) for these layers
                        gptr4 Pt16 = (gptr4)G.LsP + G.ls16_off + (size_t)sb16_slabs(nbk) * 128 + (size_t)(offt + rw * (rw - 1) / 2) * 64 + lane;
                        if (mine) {
#pragma unroll
                            for (int u = 0; u < SBT; ++u) Ti[u] = Pt16[(size_t)(u < rw ? u : (rw > 0 ? rw - 1 : 0)) * 64];
                            f32x4 rs[NS];
#pragma unroll
                            for (int t = 0; t < NS; ++t) {
                                rs[t] = acc[t] * s_r;
                                f16x4 h1, h2;
                                split_b16(rs[t], 1.0f, h1, h2);
                                A16 pk; pk.h1 = h1; pk.h2 = h2;
                                f32x4 w_;
                                __builtin_memcpy(&w_, &pk, 16);
                                at[(bi * 4 + gq) * NSAMP + 16 * t + jq] = w_;
                                rs[t] = rs[t] * s_i;
                                acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                            }
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
#pragma unroll
                                for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Dg[s], rs[t][s], acc[t], 0, 0, 0);
                            }
                        }
                        DBG_WSTAMP(49 + 6 * I);
                        __syncthreads();                          // r_I complete (published by every row)
                        DBG_WSTAMP(50 + 6 * I);
                        if (mine) {
                            for (int q0 = 0; q0 < rw; q0 += SBT) {
#pragma unroll
                                for (int u = 0; u < SBT; ++u) {
                                    const int q = q0 + u;
                                    if (q < rw) {                 // (wave-uniform)
                                        const A16 ah = as_a16(Ti[u]);
                                        const f16x8 a1 = __builtin_shufflevector(ah.h1, ah.h1, 0, 1, 2, 3, 4, 5, 6, 7);
                                        const f16x8 a2 = __builtin_shufflevector(ah.h2, ah.h2, 0, 1, 2, 3, 4, 5, 6, 7);
                                        f32x4 b[NS];
#pragma unroll
                                        for (int t = 0; t < NS; ++t) b[t] = at[((r0 + q) * 4 + gq) * NSAMP + 16 * t + jq];
#pragma unroll
                                        for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(f16x8, b[t]), acc[t], 0, 0, 0);
#pragma unroll
                                        for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, __builtin_bit_cast(f16x8, b[t]), acc[t], 0, 0, 0);
                                        FW_PIN_LOADS();
                                        Ti[u] = Pt16[(size_t)(q + SBT < rw ? q + SBT : rw - 1) * 64];
                                        FW_PIN_LOADS();
                                    }
                                }
                            }
                            const float un = 1.0f / (s_r * s_i);
#pragma unroll
                            for (int t = 0; t < NS; ++t) acc[t] = acc[t] * un;
                        }
                    } else {
                        // the row's first blocks of the inverse super-block are requested BEFORE the barrier (nothing in them depends on r_I):
                        // the wait for the slowest row covers their round trip; SBT blocks in flight from then on (one until round 5).
                        // (Tried in round 5 and not kept: the short row's wave also taking the long row of its SIMD partner for the last
                        // sub-tiles -- per-wave times level out, 3700-5400 clocks instead of 1400-5600, but the phase is as long as before: it is
                        // bound by what one SIMD issues, not by the balance between its two waves; and the extra accumulators spill.)
                        f32x4 Ti[SBT];
                        gptr4 Pt = Ap + (size_t)(off + nr * r0 + rw * (rw + 1) / 2) * 64;
                        if (mine) {
#pragma unroll
                            for (int u = 0; u < SBT; ++u) Ti[u] = Pt[(size_t)(u <= rw ? u : rw) * 64];
                        }
                        DBG_WSTAMP(49 + 6 * I);
                        if (r0 > 0) __syncthreads();                  // r_I complete
                        DBG_WSTAMP(50 + 6 * I);
                        if (mine) {
#pragma unroll
                            for (int t = 0; t < NS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                            for (int q0 = 0; q0 <= rw; q0 += SBT) {
#pragma unroll
                                for (int u = 0; u < SBT; ++u) {
                                    const int q = q0 + u;
                                    if (q <= rw) {                    // (wave-uniform)
                                        const f32x4 a_cur = Ti[u];
                                        f32x4 b[NS];
#pragma unroll
                                        for (int t = 0; t < NS; ++t) b[t] = at[((r0 + q) * 4 + gq) * NSAMP + 16 * t + jq];
#pragma unroll
                                        for (int s = 0; s < 4; ++s) {
#pragma unroll
                                            for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s], b[t][s], acc[t], 0, 0, 0);
                                        }
                                        FW_PIN_LOADS();
                                        Ti[u] = Pt[(size_t)(q + SBT <= rw ? q + SBT : rw) * 64];
                                        FW_PIN_LOADS();
                                    }
                                }
                            }
                        }
                    }
                    DBG_WSTAMP(51 + 6 * I);
                    __syncthreads();                              // every row of the super-block has read r_I
                    DBG_WSTAMP(52 + 6 * I);
                    if (mine) {
#pragma unroll
                        for (int t = 0; t < NS; ++t) {
                            const int tcol = 16 * t + jq;
                            if constexpr (S16) {                  // the two f16 planes of 2^ea a (lane (gq, jq): 8 bytes of each vector)
                                f16x4 h1, h2;
                                split_b16(acc[t], sa_sb, h1, h2);
                                char* pl = reinterpret_cast<char*>(at) + ((size_t)((4 * bi + 2 * (gq >> 1)) * NSAMP + tcol)) * 16 + 8 * (gq & 1);
                                *reinterpret_cast<f16x4*>(pl) = h1; *reinterpret_cast<f16x4*>(pl + NSAMP * 16) = h2;
                            } else at[(bi * 4 + gq) * NSAMP + tcol] = acc[t];
                            ssq[t] += colsumsq4(acc[t]);
                            if (o_a && tcol < nvalid) *((gout4)(o_a + (size_t)(t0 + tcol) * G.Mp + 16 * bi + 4 * gq)) = acc[t];
                        }
                    }
                    DBG_WSTAMP(53 + 6 * I);
                    __syncthreads();                              // a_I visible (next super-block's product, stage 2)
                    off += nr * r0 + nr * (nr + 1) / 2;
                    off16 += nr * (r0 >> 1);
                    offt += nr * (nr - 1) / 2;
                }
                // |a|^2: every wave's share to its own slot, summed in a fixed order
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                    const float sq = xgroup_sum_mfma(ssq[t]);
                    if (gq == 0) asq[(2 + wave) * NSAMP + 16 * t + jq] = sq;
                }
                __syncthreads();
                if (tid < NSAMP) {
                    float sq = 0.f;
#pragma unroll
                    for (int w = 0; w < FW_WAVES; ++w) sq += asq[(2 + w) * NSAMP + tid];
                    asq[tid] = sq; asq[NSAMP + tid] = 0.f;
                }
            } else
            if ((SHP && INV8) || inv8_layer(nbk)) {
                // ---- eight blocks (the headline's M = 128): a = X k with the explicit inverse, one block row per wave (stage1_inv8)
                {
                    f32x4* uz = reinterpret_cast<f32x4*>(usq);      // every |u|^2 slot cleared first (stage 2 fills only some)
                    for (int i = tid; i < (FW_WAVES * R * NSAMP) / 4; i += FW_THREADS) uz[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if constexpr (S16 && !BIG) {
                    if (s2_mw0 == wave) {                         // the q_mu^T slabs of stage 2, requested now (their first read of an evaluation comes from HBM)
                        gptr4 Pq = (gptr4)G.QmuP + lane;
#pragma unroll
                        for (int u = 0; u < 4; ++u) { Qpre[u][0] = Pq[(size_t)u * 128]; Qpre[u][1] = Pq[(size_t)u * 128 + 64]; }
                        q_pre = true;
                    }
                }
                const float sa8 = cst[IWVI_CST_SA];
                if (SHP || G.ls_off >= 0) stage1_inv8<NS, S16>(reinterpret_cast<const f32x4*>(sm + G.ls_off) + lane, at, wave, gq, jq, sa8, asq, o_a, t0, nvalid, G.Mp);
                else stage1_inv8<NS, S16>((gptr4)G.LsP + lane, at, wave, gq, jq, sa8, asq, o_a, t0, nvalid, G.Mp);
            } else {
            if (wave >= NS) {
                // the waves without a sub-tile of their own clear every |u|^2 slot first (stage 2 fills only some)
                f32x4* uz = reinterpret_cast<f32x4*>(usq);
                for (int i = (wave - NS) * 64 + lane; i < (FW_WAVES * R * NSAMP) / 4; i += (FW_WAVES - NS) * 64) uz[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                // ... and the one that multiplies by q_mu^T in stage 2 (plan_stage2 puts it here when it can) requests those slabs now: their first
                // read of an evaluation comes from HBM (written by the precompute launch), 5000 clocks of that wave when requested in stage 2
                if constexpr (S16 && !BIG) {
                    if (s2_mw0 == wave && nbk <= 8) {
                        gptr4 Pq = (gptr4)G.QmuP + lane;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const size_t kq = (size_t)(u < (nbk >> 1) ? u : (nbk >> 1) - 1) * 128;
                            Qpre[u][0] = Pq[kq]; Qpre[u][1] = Pq[kq + 64];
                        }
                        q_pre = true;
                    }
                }
            }
            if (wave < NS) {
                const int tcol = 16 * wave + jq;                  // this lane's sample column
                const gptr4 Ap = (gptr4)G.LsP + lane;
                const gout1 arow = (o_a && tcol < nvalid) ? o_a + (size_t)(t0 + tcol) * G.Mp : (gout1)nullptr;
                float ssq = 0.f;
                if constexpr (SHP && !INV8) {
                    ssq = stage1_unrolled<NS, 8, true, S16>(reinterpret_cast<const f32x4*>(sm + G.ls_off) + lane, kuf, at, tcol, gq, arow, st1_sb);
                } else
                if (G.ls_off >= 0 && nbk <= 8) {
                    const f32x4* Al = reinterpret_cast<const f32x4*>(sm + G.ls_off) + lane;      // staged in LDS
                    switch (nbk) {
                        case 1: ssq = stage1_unrolled<NS, 1, false, false>(Al, kuf, at, tcol, gq, arow, st1_sb); break;
                        case 2: ssq = stage1_unrolled<NS, 2, true, S16>(Al, kuf, at, tcol, gq, arow, st1_sb); break;
                        case 3: ssq = stage1_unrolled<NS, 3, false, false>(Al, kuf, at, tcol, gq, arow, st1_sb); break;
                        case 4: ssq = stage1_unrolled<NS, 4, true, S16>(Al, kuf, at, tcol, gq, arow, st1_sb); break;
                        case 5: ssq = stage1_unrolled<NS, 5, false, false>(Al, kuf, at, tcol, gq, arow, st1_sb); break;
                        case 6: ssq = stage1_unrolled<NS, 6, true, S16>(Al, kuf, at, tcol, gq, arow, st1_sb); break;
                        case 7: ssq = stage1_unrolled<NS, 7, false, false>(Al, kuf, at, tcol, gq, arow, st1_sb); break;
                        default: ssq = stage1_unrolled<NS, 8, true, S16>(Al, kuf, at, tcol, gq, arow, st1_sb); break;
                    }
                } else
                switch (nbk) {
                    case 1: ssq = stage1_unrolled<NS, 1, false, false>(Ap, kuf, at, tcol, gq, arow, st1_sb); break;
                    case 2: ssq = stage1_unrolled<NS, 2, true, S16>(Ap, kuf, at, tcol, gq, arow, st1_sb); break;
                    case 3: ssq = stage1_unrolled<NS, 3, false, false>(Ap, kuf, at, tcol, gq, arow, st1_sb); break;
                    case 4: ssq = stage1_unrolled<NS, 4, true, S16>(Ap, kuf, at, tcol, gq, arow, st1_sb); break;
                    case 5: ssq = stage1_unrolled<NS, 5, false, false>(Ap, kuf, at, tcol, gq, arow, st1_sb); break;
                    case 6: ssq = stage1_unrolled<NS, 6, true, S16>(Ap, kuf, at, tcol, gq, arow, st1_sb); break;
                    case 7: ssq = stage1_unrolled<NS, 7, false, false>(Ap, kuf, at, tcol, gq, arow, st1_sb); break;
                    case 8: ssq = stage1_unrolled<NS, 8, true, S16>(Ap, kuf, at, tcol, gq, arow, st1_sb); break;
                    default: if constexpr (BIG) {
                        // generic column-at-a-time form (M > 128): right-hand sides in the LDS tile, the packed factor
                        // streamed from L2.  Column bj: a_bj = Dinv_bj r_bj (4 dependent MFMAs), then the updates of the
                        // rows below it four block rows at a time -- four independent accumulator chains in flight, their
                        // A blocks requested one group ahead -- so the phase is bound by MFMA issue, not by the latency of
                        // one dependent chain per block.  (Clamped indices: a short last group recomputes its last row,
                        // nothing of it is written back.)
                    int tcol_g = tcol;                              // opaque copy: this rarely taken path's tile addresses are formed here, not
                    asm volatile("" : "+v"(tcol_g));                // hoisted to the top of the kernel where they cost a live (spilled) register
                    const int tcol = tcol_g;
                    f32x4 xcur = kuf[gq * NSAMP + tcol];             // r_0
                    for (int bj = 0; bj < nbk; ++bj) {
                        const size_t col = (size_t)tri_upper_off(nbk, bj);
                        const int m = nbk - 1 - bj;                  // block rows below
                        const f32x4 Dv = Ap[col * 64];
                        f32x4 An[4];
#pragma unroll
                        for (int gi = 0; gi < 4; ++gi) An[gi] = Ap[(col + 1 + (m > 0 ? (gi < m ? gi : m - 1) : -1)) * 64];
                        f32x4 res = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int s = 0; s < 4; ++s) res = __builtin_amdgcn_mfma_f32_16x16x4f32(Dv[s], xcur[s], res, 0, 0, 0);
                        at[(bj * 4 + gq) * NSAMP + tcol] = res;
                        ssq += colsumsq4(res);
                        if (o_a && tcol < nvalid)
                            *((gout4)(o_a + (size_t)(t0 + tcol) * G.Mp + 16 * bj + 4 * gq)) = res;
                        for (int b0 = 0; b0 < m; b0 += 4) {
                            f32x4 A[4], y[4];
#pragma unroll
                            for (int gi = 0; gi < 4; ++gi) {
                                A[gi] = An[gi];
                                const int b = b0 + gi < m ? b0 + gi : m - 1;
                                y[gi] = at[((bj + 1 + b) * 4 + gq) * NSAMP + tcol];
                            }
                            if (b0 + 4 < m) {
#pragma unroll
                                for (int gi = 0; gi < 4; ++gi) An[gi] = Ap[(col + 1 + (b0 + 4 + gi < m ? b0 + 4 + gi : m - 1)) * 64];
                            }
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
#pragma unroll
                                for (int gi = 0; gi < 4; ++gi) y[gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[gi][s], res[s], y[gi], 0, 0, 0);
                            }
#pragma unroll
                            for (int gi = 0; gi < 4; ++gi) {
                                if (b0 + gi < m) {
                                    if (b0 + gi == 0) xcur = y[gi];      // next column's right-hand side, kept in registers
                                    else at[((bj + 1 + b0 + gi) * 4 + gq) * NSAMP + tcol] = y[gi];
                                }
                            }
                        }
                    }
                    }
                }
                ssq = xgroup_sum_mfma(ssq);
                if (gq < 2) asq[gq * NSAMP + tcol] = gq == 0 ? ssq : 0.f;   // slot 0 carries it; the variance reads two slots (the super-block solve fills both)
            }
            }                                                     // (nbk <= 8)
            if (g.stamps && lane == 0 && li == 1) g.stamps[(size_t)blockIdx.x * 128 + 118 + wave] = clock64();
            __syncthreads();
            FW_STAMP(2 + li * 6 + 2);
            FW_REBASE();
            // the solve is over: the next GP layer's stream leaves its registers for the staging buffer (free from here on)
#pragma unroll
            for (int i = 0; i < LSN; ++i)
                if (tid + i * FW_THREADS < lsn4) reinterpret_cast<f32x4*>(sm + H.nx_ls_off)[tid + i * FW_THREADS] = lsn[i];
            // ---- stage 2 on split-f16 operands (iwvi_common.h: s16_*): the a tile is rewritten IN PLACE as two f16 planes
            //      (h1 = f16(a 2^ea), h2 = f16(a 2^ea - h1); 16-B vectors [(kc*4 + g) * NSAMP + sample] = a[32 kc + 8 g .. + 7]), then every
            //      16 x 32 slab of L_r^T (two planes from L2) takes three v_mfma_f32_16x16x32_f16 per sub-tile -- h1 h1' + h1 h2' + h2 h1' --
            //      instead of eight fp32 MFMAs at twice their cycles (scripts/ubench/stage2_split16.hip: 281 vs 1436 clocks per slab).
            if constexpr (S16) {
                using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
                const int nvec = nbk * 2 * NSAMP;                 // vectors per plane
                // this wave's first slabs: requested now, so that the conversion below covers their L2 round trip
                // the slab stream pairs the row-blocks 2p and 2p+1 (same chunks of k, same B vectors): one STEP = [slab(2p, kc) h1 | h2 |
                // slab(2p+1, kc) h1 | h2], 4 KiB.  Step 0 was requested ahead of stage 1, step 1 is requested here, behind the solve.
                const int nstp = s2_nblocks >> 1;
                f32x4 A[3][4];
#pragma unroll
                for (int i = 0; i < 3; ++i) A[i][0] = A[i][1] = A[i][2] = A[i][3] = f32x4{0.f, 0.f, 0.f, 0.f};
                A[0][0] = ring[0]; A[0][1] = ring2e[0]; A[0][2] = ring[1]; A[0][3] = ring2e[1];
                if (nstp > 0) {
                    const size_t o_ = (size_t)(1 < nstp ? 1 : nstp - 1) * 256;
#pragma unroll
                    for (int i = 0; i < 4; ++i) A[1][i] = s2_P[o_ + 64 * i];
                }
                // In-place conversion: the eight values a[32 kc + 8 g .. + 7] of a sample are the two float4 rows 8 kc + 2 g and 8 kc + 2 g + 1
                // of the fp32 tile; their h1 vector goes back to the first, their h2 vector to the second (planes interleaved row by row):
                // every item reads and writes its own two slots -- no hazard, no temporaries, every thread busy.
                if (BIG && nbk > 8 && nbk < FW_SB_MIN_NBK) {      // (nbk <= 8 and the super-block solve write the planes themselves)
                    const float sa = cst[IWVI_CST_SA];
                    for (int v = tid; v < nvec; v += FW_THREADS) {
                        const int j = v % NSAMP, kg = v / NSAMP;          // kg = 4 kc + g
                        const int row = (2 * kg) * NSAMP + j;
                        const f32x4 x0 = at[row], x1 = at[row + NSAMP];
                        f16x8 h1, h2;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float a0 = x0[e] * sa, a1 = x1[e] * sa;
                            h1[e] = (_Float16)a0; h2[e] = (_Float16)(a0 - (float)h1[e]);
                            h1[4 + e] = (_Float16)a1; h2[4 + e] = (_Float16)(a1 - (float)h1[4 + e]);
                        }
                        at[row] = __builtin_bit_cast(f32x4, h1); at[row + NSAMP] = __builtin_bit_cast(f32x4, h2);
                    }
                    __syncthreads();
                }
                if (g.stamps && lane == 0 && li == 1) g.stamps[(size_t)blockIdx.x * 128 + 100 + wave] = clock64();
                const f32x4* p1 = at + (size_t)(2 * gq) * NSAMP + jq;   // h1 vector of chunk kc, sub-tile t: p1[kc * 8 * NSAMP + 16 t]; h2: the next row
                const f32x4* p2 = p1 + NSAMP;
                // (a) q_mu^T row-blocks assigned to this wave
                DBG_WSTAMP(36);
                for (int rb = 0; rb < G.nrb; ++rb) {
                    if ((rb == 0 ? s2_mw0 : s2_mw1) != wave) continue;
                    gptr4 P = (gptr4)G.QmuP + (size_t)rb * (nbk >> 1) * 128 + lane;
                    f32x4 acc[NS];
#pragma unroll
                    for (int t = 0; t < NS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    // the slabs of four chunks are requested together (one L2 round trip per group, not one per slab; the BIG variants of more than three sub-tiles: one chunk at a
                    // time -- they have no registers to spare, see BIG_ZERO below)
                    constexpr int QG = (BIG && NS > 3) ? 1 : 4;
                    for (int kc0 = 0; kc0 < (nbk >> 1); kc0 += QG) {
                        f32x4 Q[QG][2];
                        if (!BIG && q_pre && rb == 0) {           // (on their way since stage 1; kc0 = 0 is the only group then)
#pragma unroll
                            for (int u = 0; u < QG; ++u) { Q[u][0] = Qpre[u][0]; Q[u][1] = Qpre[u][1]; }
                        } else {
#pragma unroll
                        for (int u = 0; u < QG; ++u) {
                            const size_t kq = (size_t)(kc0 + u < (nbk >> 1) ? kc0 + u : (nbk >> 1) - 1) * 128;
                            Q[u][0] = P[kq]; Q[u][1] = P[kq + 64];
                        }
                        }
#pragma unroll
                        for (int u = 0; u < QG; ++u) {
                            const int kc = kc0 + u;
                            if (kc < (nbk >> 1)) {
                                const f16x8 a1 = __builtin_bit_cast(f16x8, Q[u][0]), a2 = __builtin_bit_cast(f16x8, Q[u][1]);
                                f32x4 b1[NS], b2[NS];
#pragma unroll
                                for (int t = 0; t < NS; ++t) { b1[t] = p1[kc * 8 * NSAMP + 16 * t]; b2[t] = p2[kc * 8 * NSAMP + 16 * t]; }
#pragma unroll
                                for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(f16x8, b1[t]), acc[t], 0, 0, 0);
#pragma unroll
                                for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(f16x8, b2[t]), acc[t], 0, 0, 0);
#pragma unroll
                                for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, __builtin_bit_cast(f16x8, b1[t]), acc[t], 0, 0, 0);
                            }
                        }
                    }
                    const float fm = cst[IWVI_CST_FMEAN];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 16 * rb + 4 * gq + e;
                        if (r < R) {
#pragma unroll
                            for (int t = 0; t < NS; ++t) meanp[r * NSAMP + 16 * t + jq] = acc[t][e] * fm;
                        }
                    }
                }
                // (b) this wave's contiguous run of (r, p) jobs -- row-blocks 2p and 2p+1 of L_r^T, nbk/2 - p steps -- as one linear stream
                DBG_WSTAMP(37);
                if (nstp > 0) {
                    int r = s2_r0, bp = s2_bi0 >> 1;
                    gptr4 P = s2_P;
                    f32x4 acc0[NS], acc1[NS];
                    float ssq[NS];
#pragma unroll
                    for (int t = 0; t < NS; ++t) { acc0[t] = acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f}; ssq[t] = 0.f; }
                    const int nkc = nbk >> 1;
                    int kc = bp, c = nkc - bp;
                    bool fresh = true;
                    // (the BIG variant of five sub-tiles clears its accumulators between jobs instead: the second copy of a step's first ten MFMAs
                    //  put it into scratch -- configs[3]: stage 1 28100 -> 37500 clocks)
                    constexpr bool BIG_ZERO = BIG && NS > 3;
                    // B vectors serve both row-blocks: h1 of the NEXT step is requested while this step's last ten MFMAs (on h2) issue, h2 of
                    // this step at its top, under the twenty MFMAs on h1 -- half the LDS reads per MFMA of one row-block per job
                    f32x4 b1[NS], b2[NS];
#pragma unroll
                    for (int t = 0; t < NS; ++t) b1[t] = p1[kc * 8 * NSAMP + 16 * t];
                    for (int q0 = 0; q0 < nstp; q0 += 3) {
#pragma unroll
                        for (int u = 0; u < 3; ++u) {
                            const int q = q0 + u;
                            if (q < nstp) {
#ifdef IWVI_S2_STEP_STAMPS
                                if (g.stamps && lane == 0 && li == 1 && blockIdx.x < 1900 && q < 32) g.stamps[(size_t)(g.nchunks + blockIdx.x * 8 + wave) * 128 + q] = clock64();
#endif
                                const size_t nx = (size_t)(q + 2 < nstp ? q + 2 : nstp - 1) * 256;
#pragma unroll
                                for (int i = 0; i < 4; ++i) A[(u + 2) % 3][i] = P[nx + 64 * i];
                                const f16x8 a10 = __builtin_bit_cast(f16x8, A[u][0]), a20 = __builtin_bit_cast(f16x8, A[u][1]);
                                const f16x8 a11 = __builtin_bit_cast(f16x8, A[u][2]), a21 = __builtin_bit_cast(f16x8, A[u][3]);
#pragma unroll
                                for (int t = 0; t < NS; ++t) b2[t] = p2[kc * 8 * NSAMP + 16 * t];
                                if (!BIG_ZERO && fresh) {          // first step of a job: accumulate onto the constant 0 (no registers to clear between jobs)
                                    const f32x4 Z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                                    for (int t = 0; t < NS; ++t) acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a10, __builtin_bit_cast(f16x8, b1[t]), Z, 0, 0, 0);
#pragma unroll
                                    for (int t = 0; t < NS; ++t) acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a11, __builtin_bit_cast(f16x8, b1[t]), Z, 0, 0, 0);
                                } else {
#pragma unroll
                                for (int t = 0; t < NS; ++t) acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a10, __builtin_bit_cast(f16x8, b1[t]), acc0[t], 0, 0, 0);
#pragma unroll
                                for (int t = 0; t < NS; ++t) acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a11, __builtin_bit_cast(f16x8, b1[t]), acc1[t], 0, 0, 0);
                                }
                                fresh = false;
#pragma unroll
                                for (int t = 0; t < NS; ++t) acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a20, __builtin_bit_cast(f16x8, b1[t]), acc0[t], 0, 0, 0);
#pragma unroll
                                for (int t = 0; t < NS; ++t) acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a21, __builtin_bit_cast(f16x8, b1[t]), acc1[t], 0, 0, 0);
                                // where the next step's B vectors live: next chunk of this job, else the first chunk of the next job (a select, not a branch)
                                const int bp_n = bp + 1 == nkc ? 0 : bp + 1;
                                const int kc_n = (c > 1) ? kc + 1 : bp_n;
#pragma unroll
                                for (int t = 0; t < NS; ++t) b1[t] = p1[kc_n * 8 * NSAMP + 16 * t];
#pragma unroll
                                for (int t = 0; t < NS; ++t) acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a10, __builtin_bit_cast(f16x8, b2[t]), acc0[t], 0, 0, 0);
#pragma unroll
                                for (int t = 0; t < NS; ++t) acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a11, __builtin_bit_cast(f16x8, b2[t]), acc1[t], 0, 0, 0);
                                ++kc;
                                if (--c == 0) {
                                    // row-blocks (r, 2p) and (r, 2p+1) complete: add their squares, start the next pair.  The squares are summed at the
                                    // scale of the operands and taken back to the scale of u once per r (fr is a power of two: the same bits as scaling
                                    // every accumulator first -- 40 multiplies per job); only the saved u needs the accumulators themselves scaled
                                    const float fr = cst[IWVI_CST_FR + r];
                                    if (o_u) {
#pragma unroll
                                        for (int t = 0; t < NS; ++t) {
                                            const int j = 16 * t + jq;
                                            if (j < nvalid) {
                                                gout1 ur = o_u + ((size_t)r * g.T + (t0 + j)) * G.Mp + 32 * bp + 4 * gq;
                                                *((gout4)ur) = acc0[t] * fr; *((gout4)(ur + 16)) = acc1[t] * fr;
                                            }
                                        }
                                    }
#pragma unroll
                                    for (int t = 0; t < NS; ++t) ssq[t] += colsumsq8(acc0[t], acc1[t]);
                                    if constexpr (BIG_ZERO) {
#pragma unroll
                                        for (int t = 0; t < NS; ++t) acc0[t] = acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                                    }
                                    fresh = true;
                                    ++bp;
                                    if (bp == nkc || q == nstp - 1) {
                                        const float fr2 = fr * fr;
                                        float sq[NS];              // (the sums first, then one predicated block of stores: the MFMAs issue back to back)
#pragma unroll
                                        for (int t = 0; t < NS; ++t) { sq[t] = xgroup_sum_mfma(ssq[t]) * fr2; ssq[t] = 0.f; }
                                        if (gq == 0) {
#pragma unroll
                                            for (int t = 0; t < NS; ++t) usq[(wave * R + r) * NSAMP + 16 * t + jq] = sq[t];
                                        }
                                        if (bp == nkc) { bp = 0; ++r; }
                                    }
                                    c = nkc - bp; kc = bp;
                                }
                            }
                        }
                    }
                }
            } else
            // ---- stage 2: u block (r, bi) = sum_{bk >= bi} LrT(bi, bk) a(bk), only |u|^2 kept; mean = q_mu^T a ----
            {
                if (wave >= FW_WAVES / 2) __builtin_amdgcn_s_setprio(1);   // the second-dispatched half loses issue arbitration otherwise
                // (a) q_mu^T row-blocks assigned to this wave: mean = q_mu^T a  (temp_workaround.py:68)
                for (int rb = 0; rb < G.nrb; ++rb) {
                    if ((rb == 0 ? s2_mw0 : s2_mw1) != wave) continue;
                    gptr4 P = (gptr4)G.QmuP + (size_t)rb * nbk * 64 + lane;
                    f32x4 acc[NS];
#pragma unroll
                    for (int t = 0; t < NS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    f32x4 a_cur = P[0];
                    const f32x4* Bp = at + (size_t)gq * NSAMP + jq;
                    for (int c = 0; c < nbk; ++c) {
                        const f32x4 a_nx = P[(size_t)(c + 1 < nbk ? c + 1 : c) * 64];
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
#pragma unroll
                            for (int t = 0; t < NS; ++t)
                                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s], Bp[16 * t][s], acc[t], 0, 0, 0);
                        }
                        a_cur = a_nx; Bp += 4 * NSAMP;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 16 * rb + 4 * gq + e;
                        if (r < R) {
#pragma unroll
                            for (int t = 0; t < NS; ++t) meanp[r * NSAMP + 16 * t + jq] = acc[t][e];
                        }
                    }
                }
                if (g.stamps && lane == 0 && li == 1) g.stamps[(size_t)blockIdx.x * 128 + 100 + wave] = clock64();
                // (b) this wave's contiguous run of (r, bi) row-block jobs: one linear stream of packed blocks
                const int nblocks = s2_nblocks;
                if (nblocks > 0) {
                    int r = s2_r0, bi = s2_bi0;
                    gptr4 P = s2_P;
                    f32x4 acc[NS];
                    float ssq[NS];
#pragma unroll
                    for (int t = 0; t < NS; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; ssq[t] = 0.f; }
                    const f32x4* Bbase = at + (size_t)gq * NSAMP + jq;   // B tile of block-row b: Bbase + b * 4 * NSAMP
                    int brow = bi;                                // block-row of the B tile in use (wave-uniform)
                    int c = nbk - bi;                             // chunks left in the current job
                    // B tiles are read one block ahead into the other of two register sets (static alternation: the loop
                    // is unrolled by 4), so the LDS latency hides behind this block's MFMAs.  The body of a block is one
                    // basic block (the next tile's row is a scalar select, not a branch) so that its operand fetches can be
                    // scheduled BETWEEN its MFMAs: issued in front of them they would wait for the VALU port behind the
                    // MFMA burst of the wave sharing this SIMD.
                    f32x4 bA[NS], bB[NS];
#pragma unroll
                    for (int t = 0; t < NS; ++t) bA[t] = Bbase[(size_t)brow * (4 * NSAMP) + 16 * t];
                    for (int q0 = 0; q0 < nblocks; q0 += 4) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int q = q0 + u;
                            if (q < nblocks) {
                                ring[(u + 3) & 3] = P[(size_t)(q + 3 < nblocks ? q + 3 : nblocks - 1) * 64];
                                const f32x4 a_cur = ring[u];
                                // where the next block's B tile lives: next chunk of this job, else the first chunk of the next job
                                const int brow_n = (c > 1) ? brow + 1 : (bi + 1 == nbk ? 0 : bi + 1);
                                const f32x4* Bn = Bbase + (size_t)brow_n * (4 * NSAMP);
                                if ((u & 1) == 0) {
#pragma unroll
                                    for (int t = 0; t < NS; ++t) bB[t] = Bn[16 * t];
#pragma unroll
                                    for (int s = 0; s < 4; ++s) {
#pragma unroll
                                        for (int t = 0; t < NS; ++t)
                                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s], bA[t][s], acc[t], 0, 0, 0);
                                    }
                                } else {
#pragma unroll
                                    for (int t = 0; t < NS; ++t) bA[t] = Bn[16 * t];
#pragma unroll
                                    for (int s = 0; s < 4; ++s) {
#pragma unroll
                                        for (int t = 0; t < NS; ++t)
                                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s], bB[t][s], acc[t], 0, 0, 0);
                                    }
                                }
                                // issue order of the block: MFMA, the ring load, then an LDS read after every other MFMA
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
                                for (int t = 0; t < NS; ++t) {
                                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                                }
                                __builtin_amdgcn_sched_group_barrier(0x008, 4 * NS - 1 - 2 * NS, 0);
                                brow = brow_n;
                                if (--c == 0) {
                                    // row-block (r, bi) complete: add its squares, start the next one
                                    if (o_u) {
#pragma unroll
                                        for (int t = 0; t < NS; ++t) {
                                            const int j = 16 * t + jq;
                                            if (j < nvalid)
                                                *((gout4)(o_u + ((size_t)r * g.T + (t0 + j)) * G.Mp + 16 * bi + 4 * gq)) = acc[t];
                                        }
                                    }
#pragma unroll
                                    for (int t = 0; t < NS; ++t) { ssq[t] += colsumsq4(acc[t]); acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                                    ++bi;
                                    if (bi == nbk || q == nblocks - 1) {
                                        // this wave's share of |u_r|^2 -> its own slot [wave][r]
#pragma unroll
                                        for (int t = 0; t < NS; ++t) {
                                            const float sq = xgroup_sum_mfma(ssq[t]);
                                            if (gq == 0) usq[(wave * R + r) * NSAMP + 16 * t + jq] = sq;
                                            ssq[t] = 0.f;
                                        }
                                        if (bi == nbk) { bi = 0; ++r; }
                                    }
                                    c = nbk - bi;
                                }
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            if (g.stamps && lane == 0 && li == 1) g.stamps[(size_t)blockIdx.x * 128 + 110 + wave] = clock64();

            __syncthreads();
            FW_STAMP(2 + li * 6 + 3);
            FW_REBASE();

            // ---- epilogue (i): per (sample, latent GP): variance, sample (temp_workaround.py:59,85,89-91) ----
            DBG_WSTAMP(40);
            for (int idx = tid; idx < NSAMP * R; idx += FW_THREADS) {
                const int r = idx / NSAMP, j = idx - r * NSAMP;
                float u2 = 0.f;
#pragma unroll
                for (int w = 0; w < FW_WAVES; ++w) u2 += usq[(w * R + r) * NSAMP + j];   // fixed order: bit-reproducible
                const float mu = meanp[r * NSAMP + j];
                // (float64 route: asq[j] already holds sigma^2 - |a|^2, differenced in float64)
                float a2;
                if ((SHP && INV8) || inv8_layer(nbk)) {              // stage1_inv8: one share of |a|^2 per wave, added in a fixed order
                    a2 = 0.f;
#pragma unroll
                    for (int w = 0; w < FW_WAVES; ++w) a2 += asq[(2 + w) * NSAMP + j];
                } else a2 = asq[j] + asq[NSAMP + j];
                const float v = (F64 && f64_l) ? fmaxf(asq[j] + u2, 0.f) : fmaxf(g_variance - a2 + u2, 0.f);
                const float z = (j < nvalid) ? zl[r * NSAMP + j] : 0.f;
                if (o_noise && j < nvalid) o_noise[(t0 + j) * R + r] = z;
                const float gs = fmaf(z, sqrtf(v), mu);
                gbuf[(0 * R + r) * NSAMP + j] = mu;
                gbuf[(1 * R + r) * NSAMP + j] = v;
                gbuf[(2 * R + r) * NSAMP + j] = gs;
                if (o_gmv && j < nvalid) {                       // what the adjoint's heads need per latent GP (csrc/backward.hip)
                    const gout1 o = o_gmv + (size_t)(t0 + j) * 3 * R;
                    o[r] = gs; o[R + r] = mu; o[2 * R + r] = v;
                }
            }
            DBG_WSTAMP(41);
            __syncthreads();
            DBG_WSTAMP(42);
            FW_STAMP(2 + li * 6 + 4);
            FW_REBASE();
            // ---- epilogue (ii): mixing (:142-145) + linear mean function (layers.py:46-48); ahead of another GP layer
            //      as ONE small MFMA product out[p][j] = sum_k A[p][k] B[k][j],  A = [W | mfA^T],  B = [f_r(j) ; x_d(j)]:
            //      wave t owns sample sub-tile t; the result lands as 4 outputs p = 4gq .. 4gq+3 of sample 16t + jq
            //      per lane, from which the next GP layer's x~ row is formed on the spot.
            const bool last = (li == g.n_layers - 1);
            const bool need_mv_any = last || o_mean || o_var;     // inner layers only hand their sample on
            if (!nx_gp) {
                const bool need_mv = need_mv_any;
                // nothing downstream needs an x~ row (last layer, or an LV layer next): a handful of outputs per
                // sample, one thread per (output, sample), dot products straight from LDS
                for (int idx = tid; idx < NSAMP * P; idx += FW_THREADS) {
                    const int p = idx / NSAMP, j = idx - p * NSAMP;
                    const long long t = t0 + j;
                    float o_s, o_m, o_v;
                    if (G.flags & FWF_HASW) {
                        o_s = o_m = o_v = 0.f;
                        if (need_mv) {
                            for (int r0 = 0; r0 < R; r0 += 4) {
                                float w[4], gm[4], gv[4], gs[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const int r = (r0 + e < R) ? r0 + e : R - 1;
                                    w[e] = (r0 + e < R) ? Wm[p * R + r] : 0.f;
                                    gm[e] = gbuf[(0 * R + r) * NSAMP + j]; gv[e] = gbuf[(1 * R + r) * NSAMP + j]; gs[e] = gbuf[(2 * R + r) * NSAMP + j];
                                }
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    o_m = fmaf(w[e], gm[e], o_m); o_v = fmaf(w[e] * w[e], gv[e], o_v); o_s = fmaf(w[e], gs[e], o_s);
                                }
                            }
                        } else o_s = dot_lds(Wm + p * R, 1, gbuf + 2 * R * NSAMP + j, NSAMP, R);
                    } else {
                        o_m = gbuf[(0 * R + p) * NSAMP + j]; o_v = gbuf[(1 * R + p) * NSAMP + j]; o_s = gbuf[(2 * R + p) * NSAMP + j];
                    }
                    float mf = 0.f;
                    if (G.mf_type == IWVI_MF_IDENTITY) mf = xin[j * XSTR + p];
                    else if (G.mf_type == IWVI_MF_LINEAR) {
                        mf = dot_lds(xin + j * XSTR, 1, mfA + p, P, D);
                        if (G.flags & FWF_HAS_MFB) mf += mfb[p];
                    }
                    xout[j * XSTR + p] = o_s + mf;
                    if (last) { obuf[p * NSAMP + j] = o_m + mf; obuf[(P + p) * NSAMP + j] = o_v; }
                    if (j < nvalid) {
                        if (o_sample) o_sample[t * P + p] = o_s + mf;
                        if (o_mean) o_mean[t * P + p] = o_m + mf;
                        if (o_var) o_var[t * P + p] = o_v;
                    }
                }
            } else if (wave < NS) {
                int j_ = 16 * wave + jq;                          // opaque copy: the row addresses below are formed here, per layer, instead of
                asm volatile("" : "+v"(j_));                      // living (spilled, in the narrow variants) across the whole layer loop
                const int j = j_;
                const long long t = t0 + j;
                const int Dm = (G.mf_type == IWVI_MF_LINEAR) ? D : 0;
                // (LEAN: the next layer's D = P <= 10 and nothing but the sample is handed on -- compile-time, or every step below is a handful of
                //  wave-uniform branches)
                const int npb = SHP ? 1 : (P + 15) >> 4;         // 16-row blocks of outputs: 1 or 2
                const bool need_mv = LEAN ? false : need_mv_any;
                const bool hasW = (G.flags & FWF_HASW) != 0;
                const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                f32x4 acc_s[2] = {zero4, zero4}, acc_m[2] = {zero4, zero4}, acc_v[2] = {zero4, zero4};
                const float* gm = gbuf; const float* gv = gbuf + R * NSAMP; const float* gs = gbuf + 2 * R * NSAMP;
                // K is laid out [R latent GPs padded to a multiple of 4 | D inputs]: every MFMA step is wholly one kind,
                // so the two loops below carry no lane-dependent choice of source (loads from clamped, valid addresses;
                // the A entry of a padded k or p is zero)
                const int p_lo = jq, p_hi = 16 + jq;              // A operand rows of the two output blocks (lane i = jq)
                const int pc_lo = p_lo < P ? p_lo : P - 1, pc_hi = p_hi < P ? p_hi : P - 1;
                // (K steps in groups of four, a group's operands requested together: one LDS round trip per group instead of one per step --
                //  the steps accumulate in the same order)
                for (int k0 = 0; k0 < R; k0 += 16) {
                    float bs[4], al[4], ah[4], bm[4], bv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = k0 + 4 * u + gq, kc = k < R ? k : R - 1;
                        bs[u] = gs[kc * NSAMP + j];
                        float wv = Wm[pc_lo * R + kc];                // (unconditional -- the block is reserved with or without a mixing matrix -- and
                        asm volatile("" : "+v"(wv));                  //  kept so: moved under `hasW` it is a branch and an LDS wait per step)
                        const float a = hasW ? wv : (pc_lo == kc ? 1.f : 0.f);
                        al[u] = (k < R && p_lo < P) ? a : 0.f;
                        ah[u] = 0.f;
                        if (npb > 1) {
                            const float a2 = hasW ? Wm[pc_hi * R + kc] : (pc_hi == kc ? 1.f : 0.f);
                            ah[u] = (k < R && p_hi < P) ? a2 : 0.f;
                        }
                        bm[u] = bv[u] = 0.f;
                        if (need_mv) { bm[u] = gm[kc * NSAMP + j]; bv[u] = gv[kc * NSAMP + j]; }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        {                                             // (a step beyond R has an A of zeros: no guard, no branch)
                            acc_s[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(al[u], bs[u], acc_s[0], 0, 0, 0);
                            if (npb > 1) acc_s[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[u], bs[u], acc_s[1], 0, 0, 0);
                            if (need_mv) {
                                acc_m[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(al[u], bm[u], acc_m[0], 0, 0, 0);
                                acc_v[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(al[u] * al[u], bv[u], acc_v[0], 0, 0, 0);
                                if (npb > 1) {
                                    acc_m[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[u], bm[u], acc_m[1], 0, 0, 0);
                                    acc_v[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[u] * ah[u], bv[u], acc_v[1], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
                DBG_WSTAMP(43);
                for (int d0 = 0; d0 < Dm; d0 += 16) {            // linear mean function rows: sample and mean alike
                    float bx[4], al[4], ah[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int d = d0 + 4 * u + gq, dc = d < Dm ? d : Dm - 1;
                        bx[u] = xin[j * XSTR + dc];
                        float a = mfA[dc * P + pc_lo];
                        asm volatile("" : "+v"(a));
                        al[u] = (d < Dm && p_lo < P) ? a : 0.f;
                        ah[u] = 0.f;
                        if (npb > 1) { const float a2 = mfA[dc * P + pc_hi]; ah[u] = (d < Dm && p_hi < P) ? a2 : 0.f; }
                    }
#pragma unroll
                    for (int u = 0; u < (SHP ? 3 : 4); ++u) {         // (compiled-in shapes: D <= 10)
                        {
                            acc_s[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(al[u], bx[u], acc_s[0], 0, 0, 0);
                            if (need_mv) acc_m[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(al[u], bx[u], acc_m[0], 0, 0, 0);
                            if (npb > 1) {
                                acc_s[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[u], bx[u], acc_s[1], 0, 0, 0);
                                if (need_mv) acc_m[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[u], bx[u], acc_m[1], 0, 0, 0);
                            }
                        }
                    }
                }
                DBG_WSTAMP(44);
                // what the product does not already carry: identity mean function, or the linear one's bias
                const bool mf_id = G.mf_type == IWVI_MF_IDENTITY, mf_b = G.mf_type == IWVI_MF_LINEAR && (G.flags & FWF_HAS_MFB);
                const float* mfp = mf_id ? xin + j * XSTR : (mf_b ? mfb : cst);
                float n2 = 0.f;
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    if (pb < npb) {
                        const int p0 = 16 * pb + 4 * gq;
                        float os[4], mfv[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int pc = (p0 + e < P) ? p0 + e : P - 1;
                            mfv[e] = mfp[pc];                      // unconditional load, selected after
                            mfv[e] = (mf_id || mf_b) ? mfv[e] : 0.f;
                            os[e] = acc_s[pb][e] + mfv[e];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (p0 + e < P) xout[j * XSTR + p0 + e] = os[e];
                        if (nx_gp) {                              // next layer's x~ entries p0 .. p0+3 and their squares
                            const f32x4 il = *reinterpret_cast<const f32x4*>(nx_cst + p0), zz = *reinterpret_cast<const f32x4*>(nx_cst + 32 + p0);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float v = fmaf(os[e], il[e], -zz[e]);
                                if (p0 + e < P) { xt[j * XSTR + p0 + e] = v; n2 = fmaf(v, v, n2); }
                            }
                        }
                        if (!LEAN && last) {                       // (not reached for the last layer -- nothing follows it --: kept for o_mean / o_var)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (p0 + e < P) { obuf[(p0 + e) * NSAMP + j] = acc_m[pb][e] + mfv[e]; obuf[(P + p0 + e) * NSAMP + j] = acc_v[pb][e]; }
                        }
                        if ((o_sample || o_mean || o_var) && j < nvalid) {     // requested outputs (not on the training path)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (p0 + e < P) {
                                    if (o_sample) o_sample[t * P + p0 + e] = os[e];
                                    if (o_mean) o_mean[t * P + p0 + e] = acc_m[pb][e] + mfv[e];
                                    if (o_var) o_var[t * P + p0 + e] = acc_v[pb][e];
                                }
                            }
                        }
                    }
                }
                DBG_WSTAMP(45);
                if (nx_gp) {
                    n2 = xgroup_sum_mfma(n2);
                    if (gq == 0) {
                        xt[j * XSTR + P] = nx_rbf ? -0.5f * n2 : n2;
                        xt[j * XSTR + P + 1] = 1.f;
                        for (int d = P + 2; d < 4 * nx_nsteps; ++d) xt[j * XSTR + d] = 0.f;
                    }
                }
            }
            if (nx_gp) xt_for = li + 1;
            DBG_WSTAMP(46);
            __syncthreads();
            DBG_WSTAMP(47);
            FW_STAMP(2 + li * 6 + 5);
        }
        float* tmp = xin; xin = xout; xout = tmp;
    }

    // ---- per-sample log-weight: Gaussian variational expectation (models.py:134,138) minus the local
    //      regularisers (:140-142) ----------------------------------------------------------------------
    // chunk-local reduction: with the IW tiling and K | NSAMP every data point's K samples sit in one chunk, so
    // logsumexp_k happens here from LDS and only one partial sum per workgroup crosses workgroups
    // (what this tail reads of the kernel arguments, requested together: read where they are used, each field was its own scalar-cache round
    //  trip on the one path every workgroup ends with -- fourteen of them, one after the other)
    FW_REBASE();
    const FwElboHot Eh = opaque_block(static_cast<const FwElboHot&>(g.e));
    struct TailHot { float* out_logw; unsigned long long* rng; int Dy, yrows, cnt, nchunks; };
    const TailHot th = opaque_block(TailHot{g.out_logw, g.rng_state, g.Dy, g.lds.yrows, g.lds.cnt, g.nchunks});
    const bool local_lse = LEAN || (Eh.enabled && Eh.ws && !Eh.mode_vi && Eh.stride_k == 1 && Eh.stride_b == Eh.K &&
                                    (NSAMP % Eh.K) == 0);
    if constexpr (LEAN) {
        // LEAN (5 <= K <= 32, K | 80): one half-wave per data point -- lane e < K of half-wave p holds sample p K + e, its log-weight goes
        // straight into the point's logsumexp by cross-lane steps (DPP within 16 lanes, one swizzle across): no LDS round trip per term, one
        // barrier instead of two.  (Before: 80 threads wrote log-weights to LDS, a barrier, ONE thread per point read its K terms twice:
        // 2.4 us of every workgroup in front of its arrival.)  The sum over k is a tree here: fixed order, not the serial one.
        const FwElboHot& E = Eh;
        const int K = E.K, npl = NSAMP / K;
        const int p = tid >> 5, e = tid & 31;
        if (p < npl) {                                             // (whole waves: two points per wave)
            const bool act = e < K;
            const int j = p * K + (act ? e : 0);
            const int Dy = th.Dy;
            const float likv = sm[th.cnt + 8];                     // (prologue)
            const float c0 = -0.5f * 1.8378770664093453f - 0.5f * logf(likv);
            const float inv2s = 0.5f / likv;
            const float* yrows = sm + th.yrows;
            float acc = 0.f;
            for (int d = 0; d < Dy; ++d) {
                const float df = yrows[d * NSAMP + j] - obuf[d * NSAMP + j];
                acc += c0 - (df * df + obuf[(Dy + d) * NSAMP + j]) * inv2s;
            }
            const float lwv = acc - lw[j];
            if (act) __hip_atomic_store(th.out_logw + t0 + j, lwv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            FW_STAMP(61);
            const float m = halfwave_all_max(act ? lwv : -INFINITY);
            const float ssum = halfwave_all_sum(act ? __expf(lwv - m) : 0.f);
            if (e == 0) {
                const float lp = m + logf(ssum) - logf((float)E.K_total);                  // models.py:148
                const long long b = t0 / K + p;
                if (E.ms) { E.ms[2 * b] = m; E.ms[2 * b + 1] = ssum; }
                if (E.logp) E.logp[b] = lp;
                xt[p] = lp;
            }
        }
        __syncthreads();
        double part = 0.0;
        if (tid == 0) {
            for (int p0 = 0; p0 < npl; p0 += 8) {                                      // fixed order; eight terms requested together
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = xt[p0 + q < npl ? p0 + q : npl - 1];
#pragma unroll
                for (int q = 0; q < 8; ++q) part += (p0 + q < npl) ? (double)v[q] : 0.0;
            }
        }
        FW_STAMP(62);
        fw_arrive_fast<NS>(gk, Eh, th.rng, th.nchunks, tid, chunk_id, part, step);
        FW_STAMP(63);
        return;
    }
    if (th.out_logw && tid < nvalid) {
        const int Dy = th.Dy;
        const float likv = sm[th.cnt + 8];                         // (prologue)
        const float c0 = -0.5f * 1.8378770664093453f - 0.5f * logf(likv);
        const float inv2s = 0.5f / likv;
        const float* yrows = sm + th.yrows;
        float acc = 0.f;
        for (int d = 0; d < Dy; ++d) {
            const float df = yrows[d * NSAMP + tid] - obuf[d * NSAMP + tid];
            acc += c0 - (df * df + obuf[(Dy + d) * NSAMP + tid]) * inv2s;
        }
        const float lwv = acc - lw[tid];
        // write-through (sc1) store: the last workgroup may read every log-weight without an acquire fence
        __hip_atomic_store(th.out_logw + t0 + tid, lwv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lw[tid] = lwv;
    }
    FW_STAMP(61);
    if (local_lse) {
        const FwElboHot& E = Eh;
        const int K = E.K, npl = nvalid / K;                      // complete points of this chunk
        __syncthreads();
        float lp = 0.f;
        if (tid < npl) {
            // (32 log-weights at a time, requested together with clamped indices: a loop of run-time length pays one LDS round trip per
            //  term, twice -- 2 us of every workgroup at K = 20, in front of its arrival.  Same order of the sums.)
            float m = -INFINITY, ssum = 0.f;
            const float* lwp = lw + tid * K;
            if (K <= 32) {
                float v[32];
#pragma unroll
                for (int e = 0; e < 32; ++e) v[e] = lwp[e < K ? e : K - 1];
#pragma unroll
                for (int e = 0; e < 32; ++e) m = fmaxf(m, v[e]);                       // (the clamped repeats change no maximum)
#pragma unroll
                for (int e = 0; e < 32; ++e) { const float t = __expf(v[e] - m); ssum += (e < K) ? t : 0.f; }
            } else {
                for (int k = 0; k < K; ++k) m = fmaxf(m, lwp[k]);
                for (int k = 0; k < K; ++k) ssum += __expf(lwp[k] - m);
            }
            lp = m + logf(ssum) - logf((float)E.K_total);                              // models.py:148
            const long long b = t0 / K + tid;
            if (E.ms) { E.ms[2 * b] = m; E.ms[2 * b + 1] = ssum; }
            if (E.logp) E.logp[b] = lp;
            xt[tid] = lp;
            xt[NSAMP + tid] = m; xt[2 * NSAMP + tid] = (float)(E.scale / (double)ssum);   // (adjoint heads below)
        }
        __syncthreads();
        if constexpr (!LEAN) {
            const FwElbo& Ef = g.e;
            float* const aw = ufirst(Ef.adj_w);
            if (aw) {
                // heads of the bound's adjoint (csrc/backward.hip: k_elbo_bwd, models.py:134-148): w = scale * softmax_k(L_nk); d / d final
                // mean = w (y - m) / s, d / d final variance = -w / (2 s); this chunk's share of d / d lik_variance in float64
                float* const adm = ufirst(Ef.adj_dmean); float* const adv = ufirst(Ef.adj_dvar);
                double ds = 0.0;
                if (tid < npl * K) {
                    const int pnt = tid / K;
                    const float wt = xt[2 * NSAMP + pnt] * __expf(lw[tid] - xt[NSAMP + pnt]);
                    const int Dy = th.Dy;
                    const float likv = sm[th.cnt + 8];
                    const float* yrows = sm + th.yrows;
                    aw[t0 + tid] = wt;
                    for (int d = 0; d < Dy; ++d) {
                        const float e = yrows[d * NSAMP + tid] - obuf[d * NSAMP + tid], v = obuf[(Dy + d) * NSAMP + tid];
                        adm[(t0 + tid) * Dy + d] = wt * e / likv;
                        adv[(t0 + tid) * Dy + d] = -0.5f * wt / likv;
                        ds += (double)wt * (-0.5 / (double)likv + 0.5 * ((double)e * e + (double)v) / ((double)likv * likv));
                    }
                }
                if (tid < 128) {                                   // the two waves that hold the chunk's samples (NSAMP <= 80): a fixed tree
                    for (int o = 32; o > 0; o >>= 1) ds += __shfl_xor(ds, o);
                    double* dsl = reinterpret_cast<double*>(xt + 3 * NSAMP + (NSAMP & 1));      // (8-byte aligned: xt is 16-byte aligned)
                    if (lane == 0) dsl[wave] = ds;
                }
                __syncthreads();
                if (tid == 0) {
                    const double* dsl = reinterpret_cast<const double*>(xt + 3 * NSAMP + (NSAMP & 1));
                    __hip_atomic_store(E.ws + th.nchunks + chunk_id, dsl[0] + dsl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        double part = 0.0;
        if (tid == 0) {
            for (int p0 = 0; p0 < npl; p0 += 8) {                                      // fixed order; eight terms requested together
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = xt[p0 + e < npl ? p0 + e : npl - 1];
#pragma unroll
                for (int e = 0; e < 8; ++e) part += (p0 + e < npl) ? (double)v[e] : 0.0;
            }
            if (!LEAN && !E.fast) __hip_atomic_store(E.ws + chunk_id, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        FW_STAMP(62);
        if (LEAN || E.fast) { fw_arrive_fast<NS>(gk, Eh, th.rng, th.nchunks, tid, chunk_id, part, step); FW_STAMP(63); return; }   // (uniform)
    }
    if constexpr (!LEAN) { fw_arrive<NS>(gk, sm, tid, chunk_id); FW_STAMP(63); }
}

// the packed arrival (fw_arrive_fast) applies when every point's K samples sit in one chunk (the kernel's local_lse), the launch finishes the
// ELBO itself, its arrivals fit the 9-bit count and the scratch holds a tagged slot per chunk for the exact (overflow) path
static void fw_decide_fast(FwArgs& a, unsigned grid, int nsamp, int64_t T) {
    const FwElbo& E = a.h.e;
    const int64_t ws_len = (T + 15) / 16;
    a.h.e.fast = (E.enabled && E.ws && E.elbo && !E.mode_vi && E.stride_k == 1 && E.stride_b == E.K && E.K > 0 && (nsamp % E.K) == 0 &&
                  a.h.rng_state && grid <= 511u && 2 * (int64_t)a.h.nchunks <= ws_len && E.kl_total <= 64 && !dbg_opt("IWVI_FW_SLOW_TAIL")) ? 1 : 0;
}

template <int NS, bool S16, bool BIG, int LEAN_MODE = 0, bool F64 = false>
static int launch_forward(const FwArgs& a, unsigned grid, size_t lds_bytes, hipStream_t stream) {
    static size_t attr_set = 0;
    if (lds_bytes > attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_dgp_forward<NS, S16, BIG, LEAN_MODE, F64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(k_dgp_forward, %zu B): %s", lds_bytes, hipGetErrorString(e)); return IWVI_ERR_LAUNCH; }
        attr_set = lds_bytes;
    }
    hipLaunchKernelGGL((k_dgp_forward<NS, S16, BIG, LEAN_MODE, F64>), dim3(grid), dim3(FW_THREADS), lds_bytes, stream, a);
    return check_launch("k_dgp_forward");
}

// Static stage-2 schedule of a GP layer: the R*nbk row-block jobs (r-major, bi ascending; job (r, bi) costs
// nbk - bi blocks) are cut into FW_WAVES contiguous runs with the smallest possible maximum cost; the q_mu^T
// row-blocks (cost nbk each) go to the last waves and count against them.
static void plan_stage2(FwGp& G) {
    // cost model: a job (r, bi) streams nbk - bi packed blocks, a q_mu^T row-block nbk; the two waves w and
    // w + FW_WAVES/2 share a SIMD (and its MFMA pipe), so the quantity to level is the load per SIMD pair
    // split-f16: a job is a PAIR of row-blocks (r, 2p), (r, 2p+1) -- they share their B vectors -- and costs are counted in slabs
    const int nbk = G.nbk, R = G.R, npr = G.s16 ? nbk / 2 : nbk, njobs = R * npr, W = FW_WAVES, nm = G.nrb < 2 ? G.nrb : 2;
    const int mean_cost = G.s16 ? nbk / 2 : nbk;
    std::vector<int> pref(njobs + 1, 0);
    for (int j = 0; j < njobs; ++j) pref[j + 1] = pref[j] + (G.s16 ? 2 * (npr - j % npr) : nbk - (j % nbk));
    // (split-f16: the loop is no longer bound by the SIMD's MFMA pipe but by each wave's own LDS reads: level the waves first)
    const bool per_wave = G.s16 != 0;
    struct Eval { int maxpair, sq, maxseg; bool pw; bool operator<(const Eval& o) const {
        if (pw && maxseg != o.maxseg) return maxseg < o.maxseg;
        return maxpair != o.maxpair ? maxpair < o.maxpair : (sq != o.sq ? sq < o.sq : maxseg < o.maxseg); } };
    int load[FW_WAVES], order[FW_WAVES];
    auto loads = [&](const int* b) {
        for (int k = 0; k < W; ++k) { load[k] = pref[b[k + 1]] - pref[b[k]]; order[k] = k; }
        std::sort(order, order + W, [&](int x, int y) { return load[x] != load[y] ? load[x] < load[y] : x < y; });
        for (int i = 0; i < nm; ++i) load[order[i]] += mean_cost;      // q_mu^T row-blocks ride on the lightest runs
        std::sort(order, order + W, [&](int x, int y) { return load[x] != load[y] ? load[x] > load[y] : x < y; });
    };
    auto evaluate = [&](const int* b) {
        loads(b);
        Eval e{0, 0, load[order[0]], per_wave};
        for (int i = 0; i < W / 2; ++i) { const int p = load[order[i]] + load[order[W - 1 - i]]; e.maxpair = p > e.maxpair ? p : e.maxpair; e.sq += p * p; }
        return e;
    };
    int b[FW_WAVES + 1];
    b[0] = 0; b[W] = njobs;
    for (int k = 1; k < W; ++k) {                                      // start: boundaries nearest the equal split
        const double target = (double)pref[njobs] * k / W;
        int best = b[k - 1];
        for (int j = b[k - 1]; j <= njobs; ++j) if (std::fabs(pref[j] - target) < std::fabs(pref[best] - target)) best = j;
        b[k] = best;
    }
    Eval best = evaluate(b);
    for (bool improved = true; improved;) {                            // boundary moves while the pair maximum drops
        improved = false;
        for (int i = 1; i < W; ++i) {
            for (int d : {-1, 1, -2, 2}) {
                const int old = b[i], nb = old + d;
                if (nb < b[i - 1] || nb > b[i + 1]) continue;
                b[i] = nb;
                const Eval e = evaluate(b);
                if (e < best) { best = e; improved = true; } else b[i] = old;
            }
        }
    }
    // runs sorted by load: the i-th heaviest and the i-th lightest share SIMD i
    loads(b);
    G.mean_wave[0] = G.mean_wave[1] = -1;
    int run_wave[FW_WAVES];
    for (int i = 0; i < W / 2; ++i) { run_wave[order[i]] = i; run_wave[order[W - 1 - i]] = i + W / 2; }
    {   // the lightest runs (which carry the q_mu^T row-blocks) as ranked before the row-blocks were added
        int seg[FW_WAVES], ord2[FW_WAVES];
        for (int k = 0; k < W; ++k) { seg[k] = pref[b[k + 1]] - pref[b[k]]; ord2[k] = k; }
        std::sort(ord2, ord2 + W, [&](int x, int y) { return seg[x] != seg[y] ? seg[x] < seg[y] : x < y; });
        for (int i = 0; i < nm; ++i) G.mean_wave[i] = (signed char)run_wave[ord2[i]];
    }
    // the first q_mu^T row-block on a wave above W/2: at the headline chunk (5 sub-tiles) those waves carry no solve in stage 1 and request
    // its slabs there.  Exchanging the SIMD pairs (0, W/2) and (1, W/2 + 1) changes no pair's load.
    if (G.mean_wave[0] == W / 2) {
        for (int k = 0; k < W; ++k) {
            const int w = run_wave[k];
            run_wave[k] = w == 0 ? 1 : w == 1 ? 0 : w == W / 2 ? W / 2 + 1 : w == W / 2 + 1 ? W / 2 : w;
        }
        for (int i = 0; i < nm; ++i) {
            const int w = G.mean_wave[i];
            G.mean_wave[i] = (signed char)(w == 0 ? 1 : w == 1 ? 0 : w == W / 2 ? W / 2 + 1 : w == W / 2 + 1 ? W / 2 : w);
        }
    }
    for (int k = 0; k < W; ++k) {
        G.jr[run_wave[k]] = (unsigned char)(b[k] / npr);
        G.jbi[run_wave[k]] = (unsigned char)(G.s16 ? 2 * (b[k] % npr) : b[k] % npr);
        G.nblk[run_wave[k]] = (unsigned short)(pref[b[k + 1]] - pref[b[k]]);
    }
    for (int w = 0; w < W; ++w) {
        const unsigned long long off = G.s16 ? ((unsigned long long)G.jr[w] * s16_slabs_total(nbk) + s16_slab_off(nbk, G.jbi[w])) * 128
                                             : ((unsigned long long)G.jr[w] * tri_blocks(nbk) + tri_upper_off(nbk, G.jbi[w])) * 64;
        const unsigned hi = (unsigned)G.nblk[w] | (unsigned)G.jr[w] << 16 | (unsigned)(G.jbi[w] & 63) << 24
                          | (G.mean_wave[0] == w ? 1u << 30 : 0u) | (G.mean_wave[1] == w ? 1u << 31 : 0u);
        G.s2w[w] = (unsigned long long)hi << 32 | (off & 0xffffffffull);
    }
}

// LDS image for a chunk of nsamp samples; fills the per-layer offsets of `a`.  stage_zt: keep every GP layer's
// Gram operand Z~ in LDS for the whole launch.
static size_t fw_plan_lds(FwArgs& a, int nsamp, int maxR, int maxP, bool stage_zt, bool stage_ls) {
    int scratch = 0, zdims = 0, o = 0, ls_max = 0;
    FwLds& l = a.h.lds;
    l.ltab = o; o += up4((int)((sizeof(FwArgs) - offsetof(FwArgs, L)) / 4));
    l.xa = o; o += up4(nsamp * a.h.xstr);
    l.xb = o; o += up4(nsamp * a.h.xstr);
    l.xt = o; o += up4(nsamp * a.h.xstr);
    l.lw = o; o += nsamp;
    l.rowi = o; o += nsamp;
    l.pidx = o; o += nsamp;
    {
        bool big = false;                                   // a layer with M > 128: one |a|^2 slot per wave besides the two
        for (int i = 0; i < a.h.n_layers; ++i) if (a.L[i].type == IWVI_LAYER_GP && (a.L[i].gp.nbk >= FW_SB_MIN_NBK || inv8_layer(a.L[i].gp.nbk))) big = true;
        l.asq = o; o += (big ? 2 + FW_WAVES : 2) * nsamp;
    }
    l.meanp = o; o += maxR * nsamp;
    l.gbuf = o; o += 3 * maxR * nsamp;
    l.obuf = o; o += 2 * maxP * nsamp;
    l.xyrows = o; o += a.h.XY ? nsamp * up4(a.h.XYdim) : 0;
    l.yrows = o; o += a.h.Y ? up4(nsamp * a.h.Dy) : 0;   // the chunk's targets, gathered in the prologue (the tail then waits for no global load)
    for (int i = 0; i < a.h.n_layers; ++i) {
        FwLayer& L = a.L[i];
        L.c_off = o;
        if (L.type == IWVI_LAYER_GP) {
            FwGp& G = L.gp;
            o += gpc_size(L.D, G.P, G.R);
            G.zt_off = -1;
            if (stage_zt) { G.zt_off = o; o += G.nbk * G.nsteps * 64; }
            G.ls_off = -1;
            if (stage_ls && G.nbk <= 8 && tri_blocks(G.nbk) * BLK16 > ls_max) ls_max = tri_blocks(G.nbk) * BLK16;
            zdims += G.R;
            const int need = gp_scratch_floats(G.Mp, G.nbk, G.R, nsamp, G.f64 != 0);
            if (need > scratch) scratch = need;
        } else {
            o += up4(L.lv.enc_out ? nsamp * 2 * L.lv.Lw : L.lv.wtotal);
            zdims += L.lv.Lw;
            const int need = lv_scratch_floats(L.lv.maxdim, nsamp);
            if (need > scratch) scratch = need;
        }
    }
    for (int i = 0; i < a.h.n_layers; ++i) {            // who hands whom a ready Gram operand
        FwLayer& L = a.L[i];
        const bool nx = i + 1 < a.h.n_layers && a.L[i + 1].type == IWVI_LAYER_GP;
        L.nx_gp = nx ? 1 : 0;
        L.nx_c_off = nx ? a.L[i + 1].c_off : L.c_off;
        L.nx_nsteps = nx ? a.L[i + 1].gp.nsteps : 0;
        L.nx_rbf = (nx && a.L[i + 1].gp.kern_type == IWVI_KERN_RBF) ? 1 : 0;
        L.nx_ls_off = -1; L.nx_ls_n = 0; L.nx_ls = nullptr;
    }
    int ls_first = -1;
    if (ls_max > 0) {                                   // one staging buffer, reused layer after layer
        for (int i = 0; i < a.h.n_layers; ++i) {
            if (a.L[i].type != IWVI_LAYER_GP) continue;
            if (a.L[i].gp.nbk <= 8) { a.L[i].gp.ls_off = o; if (ls_first < 0) ls_first = i; }
        }
        o += ls_max;
    }
    l.znoise = o;
    int z = 0;
    for (int i = 0; i < a.h.n_layers; ++i) {
        FwLayer& L = a.L[i];
        L.z_off = z;
        z += ((L.type == IWVI_LAYER_GP) ? L.gp.R : L.lv.Lw) * nsamp;
    }
    o += up4(zdims * nsamp);
    // work lists of the prologue
    a.h.ncopy = 0;
    auto add_copy = [&](const float* src, int n, int dst, bool vec) {
        if (!src || n <= 0) return;
        FwCopy& c = a.C[a.h.ncopy++];
        const bool v16 = vec && (n % 4 == 0) && (dst % 4 == 0) && (((uintptr_t)src) % 16 == 0);
        c.src = src; c.n = v16 ? -n : n; c.dst = dst;
    };
    // the first staged GP layer's forward-substitution stream rides in the prologue; every later one is fetched by
    // the GP layer before it, once that layer's solve has released the buffer
    a.h.ls_first = ls_first;
    for (int i = 0; i < a.h.n_layers; ++i) {
        if (a.L[i].type != IWVI_LAYER_GP) continue;
        for (int j = i + 1; j < a.h.n_layers; ++j) {
            if (a.L[j].type != IWVI_LAYER_GP) continue;
            if (a.L[j].gp.ls_off >= 0) {
                a.L[i].nx_ls_off = a.L[j].gp.ls_off; a.L[i].nx_ls_n = tri_blocks(a.L[j].gp.nbk) * BLK16;
                a.L[i].nx_ls = reinterpret_cast<const float*>(a.L[j].gp.LsP);
            }
            break;
        }
    }
    // who copies the big operands: the waves that draw no noise, when there are at least two of them (kernel: n_early)
    a.h.noise_drawn = 0; a.h.noise_any_src = 0;
    for (int i = 0; i < a.h.n_layers; ++i) {
        const FwLayer& L = a.L[i];
        const int dims = (L.type == IWVI_LAYER_GP) ? L.gp.R : L.lv.Lw;
        if (L.noise) a.h.noise_any_src = 1;
        else a.h.noise_drawn += ((dims + 3) >> 2) * nsamp;
    }
    a.h.nz_zero_mask = 0;
    for (int i = 0; i < IWVI_MAX_STACK; ++i) {
        const bool in = i < a.h.n_layers;
        const FwLayer& L = a.L[in ? i : 0];
        const int dims = (L.type == IWVI_LAYER_GP) ? L.gp.R : L.lv.Lw;
        a.h.nz_cnt[i] = in ? ((dims + 3) >> 2) * nsamp : 0; a.h.nz_dims[i] = in ? dims : 0; a.h.nz_zoff[i] = in ? L.z_off : 0;
        if (in && L.zero_noise) a.h.nz_zero_mask |= 1u << i;
    }
    {
        const int draw_waves = a.h.noise_any_src ? FW_WAVES : std::min(FW_WAVES, (a.h.noise_drawn + 63) / 64);
        a.h.n_early = (FW_WAVES - draw_waves >= 2) ? std::min(FW_WAVES - draw_waves, FW_WAVES / 2) : 0;   // (at least half of the waves fetch the table and the rows)
    }
    a.h.zt_mask = 0;
    // the copy list: mixing matrices, mean functions, encoder weights, then the GP layers' constant blocks and Gram operands
    for (int i = 0; i < a.h.n_layers; ++i) {
        FwLayer& L = a.L[i];
        FwNoise& nz = a.N[i];
        nz.src = L.noise; nz.zero = L.zero_noise; nz.z_off = L.z_off; nz.layer = i;
        if (L.type == IWVI_LAYER_GP) {
            const FwGp& G = L.gp;
            nz.dims = G.R;
            add_copy(G.W, G.P * G.R, L.c_off + gpc_W(), true);
            if (G.mf_type == IWVI_MF_LINEAR) {
                add_copy(G.mfA, L.D * G.P, L.c_off + gpc_A(G.P, G.R), true);
                add_copy(G.mfb, G.P, L.c_off + gpc_b(L.D, G.P, G.R), true);
            }
        } else {
            const FwLv& V = L.lv;
            nz.dims = V.Lw;
            int off = 0;
            for (int k = 0; k < V.n_enc; ++k) {
                const int nW = V.dims[k] * V.dims[k + 1], nb = V.dims[k + 1];
                add_copy(V.W[k], nW, L.c_off + off, true);
                if (V.b[k]) add_copy(V.b[k], nb, L.c_off + off + nW, true);
                else { FwCopy& c = a.C[a.h.ncopy++]; c.src = nullptr; c.n = nb; c.dst = L.c_off + off + nW; }
                off += nW + nb;
            }
        }
    }
    for (int i = 0; i < a.h.n_layers; ++i) {
        FwLayer& L = a.L[i];
        if (L.type != IWVI_LAYER_GP) continue;
        const FwGp& G = L.gp;
        add_copy(G.cst, IWVI_CST_FLOATS, L.c_off, true);                               // invls[32] | zc[32] | zmax2 | scales
        if (G.zt_off >= 0) {
            if (a.h.n_early > 0 && (((uintptr_t)G.ZtP) % 16 == 0)) a.h.zt_mask |= 1u << i;
            else add_copy(G.ZtP, G.nbk * G.nsteps * 64, G.zt_off, true);
        }
    }
    if (a.h.n_early > 0 && a.h.ncopy > a.h.n_early * (FW_MAX_COPY / FW_WAVES)) {          // (the small copies would not fit those waves' registers:
        for (int i = 0; i < a.h.n_layers; ++i)                                            //  every wave copies)
            if (a.h.zt_mask >> i & 1) add_copy(a.L[i].gp.ZtP, a.L[i].gp.nbk * a.L[i].gp.nsteps * 64, a.L[i].gp.zt_off, true);
        a.h.n_early = 0; a.h.zt_mask = 0;
    }
    l.cnt = o; o += 24;                                  // 12 counters / scalars, then one kernel variance per layer (device scalars, fetched in the prologue)
    l.scratch = o; o += up4(scratch);
    l.total = o;
    return (size_t)o * sizeof(float);
}

int dgp_forward_impl(const iwvi_layer_desc* layers, int n_layers, const float* X, int Dx, const float* XY, int XYdim,
                          const float* Y, int Dy, int64_t T, int64_t row_div, int64_t row_mod, float lik_variance,
                          uint64_t seed, uint64_t* rng_state, float* out_logw, const iwvi_elbo_desc* elbo, hipStream_t stream) {
    if (T <= 0) return IWVI_OK;                         // empty batch: nothing to do
    if (!layers || n_layers <= 0 || n_layers > IWVI_MAX_STACK) { set_error("iwvi_dgp_forward: %d layers (1..%d supported)", n_layers, IWVI_MAX_STACK); return IWVI_ERR_ARG; }
    if (!X || Dx <= 0 || Dx > IWVI_MAX_D) { set_error("iwvi_dgp_forward: null X or Dx=%d out of range (1..%d)", Dx, IWVI_MAX_D); return IWVI_ERR_ARG; }
    if (row_div < 1 || row_mod < 1) { set_error("iwvi_dgp_forward: row_div=%lld / row_mod=%lld must be >= 1", (long long)row_div, (long long)row_mod); return IWVI_ERR_ARG; }
    if (row_mod > 0x7fffffffLL || row_div > 0x7fffffffLL || T > 0x7fffffffLL - 4096) { set_error("iwvi_dgp_forward: more than 2^31 rows / samples"); return IWVI_ERR_ARG; }
    if (out_logw && (!Y || Dy <= 0 || Dy > IWVI_MAX_P || !(lik_variance > 0.f))) { set_error("iwvi_dgp_forward: out_logw needs Y, 1 <= Dy <= %d and a positive likelihood variance", IWVI_MAX_P); return IWVI_ERR_ARG; }
    FwArgs a{};
    a.h.n_layers = n_layers; a.h.X = X; a.h.XY = nullptr; a.h.Y = Y; a.h.Dx = Dx; a.h.XYdim = XYdim; a.h.Dy = Dy;
    a.h.T = T; a.h.row_div = (unsigned)row_div; a.h.row_mod = (unsigned)row_mod; a.h.lik_variance = lik_variance;
    a.h.seed = seed; a.h.rng_state = (unsigned long long*)rng_state; a.h.out_logw = out_logw;
    int D = Dx, maxR = 1, maxP = 1;
    bool need_rng = false;
    // stage 2 on split-f16 operands (one kernel variant for the launch): every GP layer must have an even number of 16-row blocks
    bool s16_all = true;                                         // ... and none asks for the fp32 variant (IWVI_LAYER_F32_STAGE2: per call, not per process)
    for (int i = 0; i < n_layers; ++i) if (layers[i].type == IWVI_LAYER_GP && (layers[i].flags & IWVI_LAYER_F32_STAGE2)) s16_all = false;
    for (int i = 0; i < n_layers; ++i) if (layers[i].type == IWVI_LAYER_GP && ((round_up(layers[i].M, 16) / 16) & 1)) s16_all = false;
    // float64 stage-1 route (IWVI_LAYER_F64_STAGE1 on any GP layer): the F64 kernel variants -- fp32 stage 2, the flagged layers' Gram and solve in float64
    bool f64_any = false;
    for (int i = 0; i < n_layers; ++i) if (layers[i].type == IWVI_LAYER_GP && (layers[i].flags & IWVI_LAYER_F64_STAGE1)) f64_any = true;
    if (f64_any) s16_all = false;
    for (int i = 0; i < n_layers; ++i) {
        const iwvi_layer_desc& d = layers[i];
        FwLayer& L = a.L[i];
        L.type = d.type; L.D = D; L.zero_noise = d.zero_noise;
        L.noise = d.noise; L.noise_out = d.noise_out; L.sample = d.sample; L.mean = d.mean; L.var = d.var;
        L.gmv_out = (d.type == IWVI_LAYER_GP) ? d.gmv_out : nullptr;
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
            const StateLayout s = state_layout(d.M, d.R);
            const char* st = (const char*)d.state;
            FwGp& G = L.gp;
            G.LsP = (const f32x4*)(st + s.off_LsP); G.LrTP = (const f32x4*)(st + s.off_LrTP);
            G.QmuP = (const f32x4*)(st + s.off_QmuP); G.ZtP = (const float*)(st + s.off_ZtP);
            G.cst = (const float*)(st + s.off_cst);
            G.W = d.W; G.mfA = d.mf_A; G.mfb = d.mf_b; G.a_out = d.a_out; G.u_out = d.u_out;
            G.M = d.M; G.Mp = s.Mp; G.nbk = s.nbk; G.nrb = s.nrb; G.nsteps = round_up(D + 2, 4) / 4;
            G.R = d.R; G.P = d.P; G.kern_type = d.kern_type; G.mf_type = d.mf_type; G.variance = d.variance;
            a.h.var_dev[i] = d.variance_dev;
            if (d.variance_dev) a.h.var_dev_mask |= 1u << i;
            G.s16 = s16_all ? 1 : 0;
            G.f64 = (d.flags & IWVI_LAYER_F64_STAGE1) ? 1 : 0;
            if (G.s16) { G.LrTP = (const f32x4*)(st + s.off_LrT16); G.QmuP = (const f32x4*)(st + s.off_Qmu16); }
            plan_stage2(G);
            if (d.R > maxR) maxR = d.R;
            if (d.P > maxP) maxP = d.P;
            D = d.P;
        } else if (d.type == IWVI_LAYER_LV) {
            FwLv& V = L.lv;
            if (d.latent_dim <= 0 || D + d.latent_dim > IWVI_MAX_D) { set_error("iwvi_lv_layer_forward: bad D=%d (1..32) or latent_dim=%d", D, d.latent_dim); return IWVI_ERR_ARG; }
            V.Lw = d.latent_dim; V.sampled_kl = d.sampled_kl; V.kl_local = d.kl_local;
            V.n_enc = 0; V.wtotal = 0; V.maxdim = 2 * d.latent_dim; V.enc_out = d.enc_out;
            if (d.enc_W && !d.enc_out) {
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
                V.n_enc = d.n_enc; V.act = d.enc_act;
                if (d.enc_act < IWVI_ACT_TANH || d.enc_act > IWVI_ACT_IDENTITY) { set_error("iwvi_lv_layer_forward: unknown activation %d", d.enc_act); return IWVI_ERR_UNSUPPORTED; }
                a.h.XY = XY;
            }
            if (D + d.latent_dim > maxP) maxP = D + d.latent_dim;
            D += d.latent_dim;
        } else { set_error("iwvi_dgp_forward: unknown layer type %d", d.type); return IWVI_ERR_ARG; }
    }
    if (out_logw) {
        if (layers[n_layers - 1].type != IWVI_LAYER_GP || D != Dy) { set_error("iwvi_dgp_forward: the last layer must be a GP layer with P == Dy (%d vs %d)", D, Dy); return IWVI_ERR_ARG; }
    }
    if (need_rng && !rng_state) { set_error("iwvi_dgp_forward: a layer draws its own noise but rng_state is NULL"); return IWVI_ERR_ARG; }
    if (elbo) {
        if (!out_logw || !rng_state) { set_error("iwvi_dgp_forward: the fused ELBO needs out_logw and rng_state (its ticket word)"); return IWVI_ERR_ARG; }
        if (elbo->B <= 0 || elbo->K <= 0 || elbo->n_glob < 0 || elbo->n_glob > FW_MAX_GLOB) { set_error("iwvi_dgp_forward: bad ELBO descriptor (B=%lld K=%d n_glob=%d)", (long long)elbo->B, elbo->K, elbo->n_glob); return IWVI_ERR_ARG; }
        const int64_t last_row = (elbo->B - 1) * elbo->stride_b + (int64_t)(elbo->K - 1) * elbo->stride_k;
        if (elbo->stride_b < 0 || elbo->stride_k < 0 || last_row >= T) { set_error("iwvi_dgp_forward: ELBO strides address row %lld of %lld", (long long)last_row, (long long)T); return IWVI_ERR_ARG; }
        FwElbo& E = a.h.e;
        E.enabled = 1; E.K = elbo->K; E.K_total = elbo->K_total > 0 ? elbo->K_total : elbo->K; E.mode_vi = elbo->mode_vi;
        E.B = elbo->B; E.stride_b = elbo->stride_b; E.stride_k = elbo->stride_k; E.scale = elbo->scale;
        E.n_glob = elbo->n_glob;
        for (int i = 0; i < elbo->n_glob; ++i) {
            if (!elbo->kl_global || !elbo->kl_global[i]) { set_error("iwvi_dgp_forward: null global KL pointer %d", i); return IWVI_ERR_ARG; }
            E.klg[i] = elbo->kl_global[i];
            E.klg_n[i] = elbo->kl_global_counts ? elbo->kl_global_counts[i] : 1;
            if (E.klg_n[i] <= 0 || E.klg_n[i] > IWVI_MAX_R) { set_error("iwvi_dgp_forward: bad global KL count %d", E.klg_n[i]); return IWVI_ERR_ARG; }
            E.kl_total += E.klg_n[i];
        }
        E.ms = elbo->out_lse_ms; E.logp = elbo->out_logp; E.elbo = elbo->out_elbo; E.ws = elbo->ws;
        E.adj_w = elbo->adj_w; E.adj_dmean = elbo->adj_dmean; E.adj_dvar = elbo->adj_dvar; E.adj_sums = elbo->adj_sums;
        if (E.adj_w || E.adj_dmean || E.adj_dvar || E.adj_sums) {
            if (!(E.adj_w && E.adj_dmean && E.adj_dvar && E.adj_sums)) { set_error("iwvi_dgp_forward: the adjoint heads adj_w / adj_dmean / adj_dvar / adj_sums come together"); return IWVI_ERR_ARG; }
            if (E.mode_vi || !E.ws || E.stride_k != 1 || E.stride_b != E.K || E.K_total != E.K) {
                set_error("iwvi_dgp_forward: the fused adjoint heads need the importance-weighted bound with contiguous samples, ws, and no sharded exchange");
                return IWVI_ERR_UNSUPPORTED;
            }
        }
        a.h.lw_init = elbo->lw_init; a.h.layer_base = elbo->noise_layer_base; a.h.x_per_sample = elbo->x_per_sample;
        a.h.lik_var_dev = elbo->lik_variance_dev;
    }
    {   // activation row stride: room for the widest layer input + 2 (x~), padded to a multiple of 4, and the widest output; odd
        int wmax = Dx + 2, dcur = Dx;
        for (int i = 0; i < n_layers; ++i) {
            const FwLayer& L = a.L[i];
            dcur = (L.type == IWVI_LAYER_GP) ? L.gp.P : dcur + L.lv.Lw;
            if (dcur + 2 > wmax) wmax = dcur + 2;
        }
        a.h.xstr = round_up(wmax, 4) + 1;
        if (a.h.xstr > XSTR_MAX) a.h.xstr = XSTR_MAX;
    }
    // chunk size: 16*NS samples per workgroup, NS no larger than what gives every CU a workgroup, then the
    // largest that fits the LDS (with Z~ staged if that fits too)
    const size_t LDS_MAX = 160 * 1024;
    int ns = (int)((T + 16 * 256 - 1) / (16 * 256));
    if (ns > FW_MAXNS) ns = FW_MAXNS;
    { const int cap = dbg_opt("IWVI_FW_MAX_NS"); if (cap > 0 && ns > cap) ns = cap; }   // development: fewer samples per workgroup than would fit
    if (ns < 1) ns = 1;
    size_t lds_bytes = 0;
    for (; ns >= 1; --ns) {
        lds_bytes = fw_plan_lds(a, 16 * ns, maxR, maxP, true, true);
        if (lds_bytes <= LDS_MAX) break;
        lds_bytes = fw_plan_lds(a, 16 * ns, maxR, maxP, true, false);
        if (lds_bytes <= LDS_MAX) break;
        lds_bytes = fw_plan_lds(a, 16 * ns, maxR, maxP, false, false);
        if (lds_bytes <= LDS_MAX) break;
    }
    if (ns < 1) { set_error("iwvi_dgp_forward: the layer stack needs %zu B of LDS per 16 samples (> 160 KiB)", lds_bytes); return IWVI_ERR_UNSUPPORTED; }
    for (int i = 0; i < n_layers; ++i) {                 // the scalar-path copy of what the layer loop reads (after the LDS plan)
        const FwLayer& L = a.L[i];
        FwHot& H = a.H[i];
        H = FwHot{};
        H.type = L.type; H.D = L.D; H.c_off = L.c_off; H.z_off = L.z_off;
        H.nx_c_off = L.nx_c_off; H.nx_nsteps = L.nx_nsteps; H.nx_ls_off = L.nx_ls_off; H.nx_ls_n = L.nx_ls_n; H.nx_ls = L.nx_ls;
        int fl = (L.nx_gp ? FWF_NX_GP : 0) | (L.nx_rbf ? FWF_NX_RBF : 0);
        if (L.noise_out || L.sample || L.mean || L.var || L.gmv_out) fl |= FWF_ANY_OUT;
        if (L.type == IWVI_LAYER_GP) {
            const FwGp& G = L.gp;
            H.M = G.M; H.Mp = G.Mp; H.nbk = G.nbk; H.nrb = G.nrb; H.nsteps = G.nsteps; H.R = G.R; H.P = G.P;
            H.kern_type = G.kern_type; H.mf_type = G.mf_type; H.zt_off = G.zt_off; H.ls_off = G.ls_off; H.variance = G.variance;
            H.LrTP = G.LrTP; H.QmuP = G.QmuP; H.LsP = G.LsP; H.ZtP = G.ZtP;
            { const StateLayout sl = state_layout(G.M, G.R); H.ls16_off = (int)(((long long)sl.off_Ls16 - (long long)sl.off_LsP) / 16); }
            if (G.W) fl |= FWF_HASW;
            if (G.s16) fl |= FWF_S16;
            if (G.f64) fl |= FWF_F64;
            if (G.mfb) fl |= FWF_HAS_MFB;
            if (G.a_out || G.u_out) fl |= FWF_ANY_OUT;
        } else {
            const FwLv& V = L.lv;
            H.R = V.Lw; H.nbk = V.n_enc; H.Mp = V.maxdim;
            H.LrTP = reinterpret_cast<const f32x4*>(V.enc_out);
            if (V.enc_out) fl |= FWF_PRE_ENC;
            if (V.sampled_kl) fl |= FWF_SAMPLED_KL;
            if (V.kl_local) fl |= FWF_ANY_OUT;
        }
        H.flags = fl;
        if (fl & FWF_PRE_ENC) a.h.pre_enc_mask |= 1u << i;
    }
    const long long chunks = (T + 16 * ns - 1) / (16 * ns);
    if (chunks > 0x7fffffffLL) { set_error("iwvi_dgp_forward: T too large"); return IWVI_ERR_ARG; }
    a.h.nchunks = (int)chunks;
    a.h.stamps = (g_stamp_buf && chunks + IWVI_MAX_STACK <= g_stamp_wgs) ? g_stamp_buf : nullptr;
    a.h.dbg_exit = g_dbg_exit;
    fw_decide_fast(a, (unsigned)chunks, 16 * ns, T);
    if (a.h.e.adj_w) {                                   // every point's K samples inside one chunk, two doubles of ws per chunk; the sums travel through ws
        if ((16 * ns) % a.h.e.K != 0 || 2 * chunks > (T + 15) / 16) {
            set_error("iwvi_dgp_forward: the fused adjoint heads need K (%d) to divide the chunk of %d samples", a.h.e.K, 16 * ns);
            return IWVI_ERR_UNSUPPORTED;
        }
        a.h.e.fast = 0;
    }
    bool big = false;                                    // a layer with M > 128: the variants that carry the generic / super-block solves
    for (int i = 0; i < n_layers; ++i) if (a.L[i].type == IWVI_LAYER_GP && a.L[i].gp.nbk > 8) big = true;
#define FW_LAUNCH(NS_) (s16_all ? (big ? launch_forward<NS_, true, true>(a, (unsigned)chunks, lds_bytes, stream)      \
                                      : launch_forward<NS_, true, false>(a, (unsigned)chunks, lds_bytes, stream))     \
                                : (big ? launch_forward<NS_, false, true>(a, (unsigned)chunks, lds_bytes, stream)     \
                                      : launch_forward<NS_, false, false>(a, (unsigned)chunks, lds_bytes, stream)))
    bool fp32_shp = false;                               // s16_all is off only because a layer asked for the fp32-MFMA stage 2
    for (int i = 0; i < n_layers; ++i) if (layers[i].type == IWVI_LAYER_GP && (layers[i].flags & IWVI_LAYER_F32_STAGE2)) fp32_shp = true;
    {   // the headline stack at the headline chunk size: the variants with its shapes and sources compiled in (k_dgp_forward: SHP / LEAN)
        // EVERY assumption the SHP / LEAN code compiles in is a condition here (the kernel has no run-time guard of its own,
        // and tests/test_gpu_lean_variant.py lists a shape per condition that must NOT take these variants):
        //   nvalid = NSAMP, grid * 80 == T  <- T % 80 == 0 (whole chunks)        nbk = 8                  <- M == 128
        //   nsteps = 3 / nx_nsteps = 3      <- D + 2 <= 12 of every GP layer      mean-function loop u < 3 <- D <= 10 (the same)
        //   npb = 1 (one block of outputs)  <- P <= 16                            rbf, operands staged, encoders precomputed, device noise
        bool shp = ns == 5 && (s16_all || fp32_shp) && !f64_any && !big && !a.h.noise_any_src && !a.h.lw_init && !a.h.x_per_sample && !g_dbg_exit && T % (16 * 5) == 0
                   && chunks * (16 * 5) == T;
        for (int i = 0; i < n_layers && shp; ++i) {
            const FwLayer& L = a.L[i];
            if (L.type == IWVI_LAYER_GP) {
                const FwGp& G = L.gp;                              // M = 128 (8 blocks), D <= 10, P <= 16, operands staged in LDS
                if (G.kern_type != IWVI_KERN_RBF || G.M != 128 || G.nbk != 8 || G.nsteps != 3 || L.D > 10 || G.P > 16 || G.ls_off < 0 || G.zt_off < 0) shp = false;
                if (L.nx_gp && L.nx_nsteps != 3) shp = false;
            }
            else if (!L.lv.enc_out) shp = false;
        }
        // mode 1: the bound's own evaluation (no per-layer output, the packed arrival; its tail: one half-wave per data point, lane = sample,
        // so a chunk must hold whole data points: K | 80 with K <= 16 samples per half-wave ... or K = 20 (4 points per chunk) -- see the tail)
        bool lean = shp && a.h.e.fast && a.h.out_logw && a.h.e.K >= 5 && a.h.e.K <= 32 && (16 * 5) % a.h.e.K == 0;
        for (int i = 0; i < n_layers && lean; ++i) if (a.H[i].flags & FWF_ANY_OUT) lean = false;
        if (shp && !dbg_opt("IWVI_FW_NO_LEAN")) {
            if (!s16_all) {                              // the strict-fp32 route (IWVI_LAYER_F32_STAGE2 on a layer) with the same shapes compiled in
                if (lean) { g_last_variant = 5 | 1 << 10; return launch_forward<5, false, false, 1>(a, (unsigned)chunks, lds_bytes, stream); }
                g_last_variant = 5 | 1 << 11;
                return launch_forward<5, false, false, 2>(a, (unsigned)chunks, lds_bytes, stream);
            }
            if (lean) { g_last_variant = 5 | 1 << 8 | 1 << 10; return launch_forward<5, true, false, 1>(a, (unsigned)chunks, lds_bytes, stream); }
            // mode 2: the same stack with outputs and the general tail (the forward of a value + gradient evaluation, predictions, read-backs)
            g_last_variant = 5 | 1 << 8 | 1 << 11;
            return launch_forward<5, true, false, 2>(a, (unsigned)chunks, lds_bytes, stream);
        }
    }
    if (f64_any) {
        g_last_variant = ns | 1 << 9 | 1 << 12;
        switch (ns) {
            case 1: return launch_forward<1, false, true, 0, true>(a, (unsigned)chunks, lds_bytes, stream);
            case 2: return launch_forward<2, false, true, 0, true>(a, (unsigned)chunks, lds_bytes, stream);
            case 3: return launch_forward<3, false, true, 0, true>(a, (unsigned)chunks, lds_bytes, stream);
            case 4: return launch_forward<4, false, true, 0, true>(a, (unsigned)chunks, lds_bytes, stream);
            default: return launch_forward<5, false, true, 0, true>(a, (unsigned)chunks, lds_bytes, stream);
        }
    }
    g_last_variant = ns | (s16_all ? 1 << 8 : 0) | (big ? 1 << 9 : 0);
    switch (ns) {
#ifndef IWVI_DEV_ONLY5   /* development builds: only the 80-sample variants (make DEV5=1) */
        case 1: return FW_LAUNCH(1);
        case 2: return FW_LAUNCH(2);
        case 3: return FW_LAUNCH(3);
        case 4: return FW_LAUNCH(4);
#endif
        default: return FW_LAUNCH(5);
    }
#undef FW_LAUNCH
}

}  // namespace iwvi

using namespace iwvi;

extern "C" int iwvi_dgp_forward(const iwvi_layer_desc* layers, int n_layers, const float* X, int Dx,
                                const float* XY, int XYdim, const float* Y, int Dy, int64_t T, int64_t row_div,
                                int64_t row_mod, float lik_variance, uint64_t seed, uint64_t* rng_state,
                                float* out_logw, const iwvi_elbo_desc* elbo, void* stream) {
    return dgp_forward_impl(layers, n_layers, X, Dx, XY, XYdim, Y, Dy, T, row_div, row_mod, lik_variance, seed,
                            rng_state, out_logw, elbo, (hipStream_t)stream);
}

/* diagnostic (not part of the drop-in surface): register a device buffer of 128 * max_workgroups 64-bit words;
 * every fused-forward launch with at most max_workgroups workgroups then stamps its phase boundaries
 * ([k] 100 MHz wall clock, [64 + k] shader clock) into it.  NULL switches stamping off. */
/* diagnostic: the k_dgp_forward variant of the last launch -- sub-tiles per workgroup | S16 << 8 | BIG << 9 | LEAN << 10 | (shapes compiled in, outputs kept) << 11 (tests/test_gpu_lean_variant.py) */
extern "C" int iwvi_debug_last_forward_variant(void) { return iwvi::g_last_variant; }
extern "C" void iwvi_debug_set_exit(int phase) { iwvi::g_dbg_exit = phase; }   /* diagnostic: fused-forward launches return after phase N */
extern "C" void iwvi_debug_set_stamps(void* buf, int64_t max_workgroups) {
    iwvi::g_stamp_buf = (unsigned long long*)buf;
    iwvi::g_stamp_wgs = buf ? max_workgroups : 0;
}

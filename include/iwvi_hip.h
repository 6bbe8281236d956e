/*
 * iwvi_hip.h -- C-ABI of the MI355X (gfx950) importance-weighted-ELBO hot path.
 *
 * The reference (hughsalimbeni/DGPs_with_IWVI) has no FFI: its hot path is a
 * chain of TensorFlow-1/GPflow-1 ops behind Python class signatures.  Each entry
 * point below replaces the op group named in its comment (file:line relative to
 * the reference checkout); dgps_with_iwvi_amd/{temp_workaround,layers,models}.py
 * bind them with ctypes and keep the reference's Python signatures.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *     the caller (PyTorch-ROCm) owns every buffer; nothing is allocated here;
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered,
 *     there is no global mutable state except the thread-local error string,
 *     and no call synchronises the device (safe inside hipGraph capture);
 *   - return 0 on success, negative IWVI_ERR_* otherwise, text in iwvi_last_error();
 *   - dense row-major float32 tensors unless stated; "T" is the flattened sample
 *     batch (B*K rows of the reference's [B, K, D] tensors, or N rows of [N, D]).
 */
#ifndef IWVI_HIP_H
#define IWVI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IWVI_ABI_VERSION 17

enum {
    IWVI_OK = 0,
    IWVI_ERR_ARG = -1,      /* bad size / null pointer / unsupported combination */
    IWVI_ERR_LAUNCH = -2,   /* hip launch or runtime error (text has hipGetErrorString) */
    IWVI_ERR_UNSUPPORTED = -3
};

/* stationary kernels of gpflow.kernels used on the path (temp_workaround.py:39,44,45) */
enum { IWVI_KERN_RBF = 0, IWVI_KERN_MATERN52 = 1 };

/* Encoder(activation_func=...) (layers.py:109,119,143-144; the reference's default is tf.nn.tanh) */
enum { IWVI_ACT_TANH = 0, IWVI_ACT_RELU = 1, IWVI_ACT_SIGMOID = 2, IWVI_ACT_SOFTPLUS = 3, IWVI_ACT_IDENTITY = 4 };

/* gpflow.mean_functions used by GPLayer.propagate (layers.py:46-48) */
enum { IWVI_MF_ZERO = 0, IWVI_MF_IDENTITY = 1, IWVI_MF_LINEAR = 2 };

#define IWVI_MAX_LAYERS 8   /* GP layers batched into one precompute launch   */
#define IWVI_MAX_R 32       /* latent GPs per layer                            */
#define IWVI_MAX_P 32       /* mixed outputs per layer                         */
#define IWVI_MAX_D 32       /* layer input dimension                           */
#define IWVI_MAX_M 512      /* inducing points per layer                       */
#define IWVI_MAX_KL 4       /* local-regulariser arrays fed to the ELBO reduce */
#define IWVI_MAX_ENC 8      /* encoder MLP layers                              */

int iwvi_version(void);
const char* iwvi_last_error(void);

/* ------------------------------------------------------------------------
 * Per-step, per-GP-layer state ("the inducing-set factorisation").
 * Replaces: Kuu(feat, kern, jitter) + tf.cholesky  (temp_workaround.py:39,48),
 * the operand side of tf.matrix_triangular_solve (:51), tf.matrix_band_part of
 * q_sqrt (:78), the operand side of Kuf (:44) and gauss_kl (layers.py:44 ->
 * temp_workaround.py:186-188).
 *
 * Mp = M rounded up to 16, nbk = Mp/16, ntri = nbk(nbk+1)/2.  Packed operands are
 * 16x16 blocks in v_mfma_f32_16x16x4_f32 fragment order.
 * The state buffer holds, in this order,
 *   double  Lm   [Mp*Mp]   lower Cholesky factor of Kuu   (written only with IWVI_GP_WANT_DENSE)
 *   double  Linv [Mp*Mp]   Lm^-1                          (written only with IWVI_GP_WANT_DENSE)
 *   float   LsP  [ntri*256]        forward-substitution stream of matrix_triangular_solve (:51): row-block bi =
 *                                  [-Lm(bi,0) .. -Lm(bi,bi-1), Lm(bi,bi)^-1], packed.  Layers with an even nbk <= 8 hold
 *                                  the off-diagonal blocks as split-f16 halves (16 B per lane: h1 x 4 | h2 x 4 of
 *                                  2^est (-Lm(bi,bj))) and the inverse diagonal blocks times 2^(-2 est): the state is an
 *                                  opaque operand image of the fused kernels, not an interchange format
 *   float   LrTP [R*ntri*256]      tril(q_sqrt[r])^T, upper-triangular blocks, packed
 *   float   QmuP [ceil(R/16)*nbk*256]  q_mu^T, packed  (mean = A^T q_mu, :68)
 *   float   ZtP  [nbk*9*64]        K_uf operand: augmented, centred, scaled inducing inputs
 *   float   cst  [64]              1 / lengthscales [32] | centre of Z / lengthscales [32]  (0 beyond D)
 *   double  kl   [IWVI_MAX_R]      kl[r] = latent GP r's share of the whitened KL[q(u) || p(u)]
 *                                  (the layer's KL is the sum of the first R entries)
 *   double  ws   [...]             factorisation workspace (16x16 blocks of the lower triangle)
 * iwvi_gp_state_bytes() returns the size; offsets via iwvi_gp_state_offsets().
 * ---------------------------------------------------------------------- */
#define IWVI_GP_WANT_DENSE 1   /* iwvi_gp_desc.flags: also write the dense float64 Lm and Lm^-1 */
#define IWVI_GP_WANT_LM    2   /* iwvi_gp_desc.flags: also write the dense float64 Lm (Lm^-1 then by iwvi_gp_dense_inverse)  */
/* iwvi_gp_desc.flags: prepare the layer for the float64 stage-1 route of the forward (IWVI_LAYER_F64_STAGE1 below): the precompute call
 * also leaves the dense float64 Lm and Lm^-1 (as with IWVI_GP_WANT_LM + iwvi_gp_dense_inverse, launched behind it on the same stream)
 * and the plain float32 inducing inputs z~ the factorisation saw. */
#define IWVI_GP_F64_STAGE1 4
/* iwvi_gp_desc.flags (ABI 17), for callers that KNOW which inputs moved since the last full precompute on this state buffer -- a
 * training step whose first op moves only the final layer's q(u) (build_models.py:288-300):
 * IWVI_GP_REUSE_FACTOR: Z, lengthscales, variance, jitter are unchanged: the factorisation and everything derived from it stay as they
 *   are, only the q(u) images (tril(q_sqrt)^T, q_mu^T, their split-f16 slabs and scales, kl[]) are rewritten -- ~5 us instead of ~26 at
 *   M = 128.  A layer of which nothing moved is simply left out of the call.
 * IWVI_GP_FACTOR_ONLY: the opposite -- the factorisation and its images only; the q(u) images are not touched (q may be written by
 *   another stream meanwhile).  Both together: error. */
#define IWVI_GP_REUSE_FACTOR 8
#define IWVI_GP_FACTOR_ONLY 16

typedef struct iwvi_gp_desc {
    const float* Z;            /* [M, D]  inducing inputs                        */
    const float* lengthscales; /* [D]     ARD lengthscales                       */
    const float* q_mu;         /* [M, R]                                         */
    const float* q_sqrt;       /* [R, M, M]; only the lower triangle is read     */
    void* state;               /* iwvi_gp_state_bytes(M, R) bytes, 256-B aligned */
    float variance;            /* kernel variance sigma^2                        */
    double jitter;             /* gpflow settings.numerics.jitter_level          */
    int32_t M, D, R;
    int32_t kern_type;         /* IWVI_KERN_*                                    */
    int32_t flags;             /* IWVI_GP_* bits above, or 0                     */
    const float* variance_dev; /* optional DEVICE scalar: read instead of `variance` when the launch runs (a trained
                                * kernel variance that lives on the device keeps a captured hipGraph valid across steps) */
} iwvi_gp_desc;

size_t iwvi_gp_state_bytes(int M, int R);
/* offsets (bytes) of {Lm, Linv, LsP, LrTP, QmuP, ZtP, cst, kl} inside the state buffer */
int iwvi_gp_state_offsets(int M, int R, size_t out_host[8]);

/* factorise up to IWVI_MAX_LAYERS layers per launch: grid (layer, role) -- role 0 Gram + Cholesky +
 * triangular inverse + packing, roles 1..R tril(q_sqrt[r])^T packing + KL share */
int iwvi_gp_precompute(const iwvi_gp_desc* layers_host, int n_layers, void* stream);
/* Lm^-1 (dense float64) from the dense Lm of states precomputed with IWVI_GP_WANT_LM (or _DENSE): one workgroup per 16-column block of
 * the inverse, n_layers <= IWVI_MAX_STACK in one launch.  The adjoint's route to the dense factors: one factorisation serves the
 * forward and the backward, and the inversion does not sit on the factorising workgroup's CU. */
int iwvi_gp_dense_inverse(const iwvi_gp_desc* layers_host, int n_layers, void* stream);

/* The same launch can also evaluate the Encoder MLP of a LatentVariableLayer (layers.py:137-152) for every row of
 * the minibatch: it does not depend on the factorisation, so it runs on otherwise idle CUs, off the critical
 * path of iwvi_dgp_forward (which then takes iwvi_layer_desc.enc_out instead of the weights).
 *   XY [rows, dims[0]];  enc_W[i] [dims[i], dims[i+1]], enc_b[i] [dims[i+1]] (host arrays of device pointers);
 *   out [rows, 2*latent_dim] = (means | raw), q_sqrt = softplus(raw - 3). */
typedef struct iwvi_enc_desc {
    const float* XY; int64_t rows;
    const float* const* enc_W; const float* const* enc_b; const int32_t* dims; int32_t n_enc, latent_dim;
    float* out;
    /* optional sampling tail (sample_X == NULL: absent).  For the IW tiling -- K samples per data row, sample
     * t = row * K + k -- the whole LatentVariableLayer is evaluated here instead of in the layer kernel (it needs
     * nothing from the factorisation): W = q_mu + z softplus(raw - 3) (layers.py:83-87), with z drawn from the same
     * counter-based stream the layer kernel would use for layer `layer_index` of evaluation *rng_state. */
    const float* X; int32_t Dx;          /* data rows [rows, Dx] (the minibatch, untiled) */
    int32_t K, sampled_kl, layer_index;
    uint64_t seed; const uint64_t* rng_state;
    float* sample_X;                     /* [rows * K, Dx + latent_dim] = concat(X tiled, W)          (layers.py:89) */
    float* sample_kl;                    /* [rows * K] local regulariser summed over the latent dims  (:98-103)      */
    float* sample_z;                     /* optional [rows * K, latent_dim]: the draws                               */
    int32_t act;                         /* IWVI_ACT_* of the hidden layers (0 = tanh)                                */
} iwvi_enc_desc;
int iwvi_model_precompute(const iwvi_gp_desc* layers_host, int n_layers,
                          const iwvi_enc_desc* encs_host, int n_encs, void* stream);

/* K1: Kuu(feat, kern, jitter) in float64 (temp_workaround.py:39)  -> Kuu [M, M] double */
int iwvi_rbf_gram_sym(const float* Z, const float* lengthscales, float variance, double jitter,
                      int kern_type, int M, int D, double* Kuu, void* stream);
/* K2: tf.cholesky(Kmm) in float64 (temp_workaround.py:48): A [M, M] (lower triangle read) ->
 * Lout [M, M] lower factor (upper triangle zero); ws = iwvi_chol_ws_bytes(M) bytes of scratch */
size_t iwvi_chol_ws_bytes(int M);
int iwvi_chol_factor(const double* A, double* Lout, int M, void* ws, void* stream);

/* ------------------------------------------------------------------------
 * GPLayer forward on a flattened sample batch (diag / marginal variance).
 * Replaces: Kuf (:44), matrix_triangular_solve (:51), Kdiag - sum A^2 (:59),
 * A^T q_mu (:68), einsum('rMm,sMn->srmn') (:78), + sum LTA^2 (:85), the
 * marginal sample (:89-91), SharedMixedMok mixing (:142-145) and the mean
 * function add (layers.py:46-48) -- one fused launch, nothing spilled to HBM.
 *
 *   F      [T/bcast_K, D]  layer input; bcast_K >= 1: every row stands for bcast_K consecutive samples
 *                   (the tiling of models.py:113 done inside the kernel); 1 = F has T rows
 *   noise  [T, R]   N(0,1) draws (z of temp_workaround.py:89); may be NULL -> z = 0
 *   W      [P, R]   SharedMixedMok.W, or NULL (then P must equal R)
 *   mf_A   [D, P], mf_b [P]  for IWVI_MF_LINEAR (mf_b may be NULL)
 *   sample/mean/var [T, P]   any may be NULL (not written)
 * var is clamped at 0 (float32 cancellation can undershoot; the reference is fp64).
 * ---------------------------------------------------------------------- */
int iwvi_gp_layer_forward(const void* state, int M, int D, int R, int P,
                          int kern_type, float variance,
                          const float* F, const float* noise, const float* W,
                          int mf_type, const float* mf_A, const float* mf_b,
                          float* sample, float* mean, float* var,
                          int64_t T, int bcast_K, void* stream);
/* the same with iwvi_layer_desc.flags bits for the layer (IWVI_LAYER_F32_STAGE2, IWVI_LAYER_F64_STAGE1); the plain entry = flags 0 */
int iwvi_gp_layer_forward_ex(const void* state, int M, int D, int R, int P,
                             int kern_type, float variance,
                             const float* F, const float* noise, const float* W,
                             int mf_type, const float* mf_A, const float* mf_b,
                             float* sample, float* mean, float* var,
                             int64_t T, int bcast_K, int layer_flags, void* stream);

/* Full covariance over the second axis (temp_workaround.py:45,56,83 with full_cov=True):
 *   F [S, N, D] -> mean [S, N, R] (+ the layer's mean function, layers.py:46-48: mf_type / mf_A [D, R] / mf_b [R] as
 *   iwvi_gp_layer_forward), cov [S, R, N, N].
 * ws: iwvi_gp_fullcov_ws_bytes(S*N, M, R) bytes of scratch (A and LTA, as the reference
 * materialises them). Plain kernels only (the SharedMixedMok branch forces full_cov=False). */
size_t iwvi_gp_fullcov_ws_bytes(int64_t T, int M, int R);
int iwvi_gp_layer_fullcov(const void* state, int M, int D, int R, int kern_type, float variance,
                          const float* F, int64_t S, int64_t N,
                          int mf_type, const float* mf_A, const float* mf_b,
                          float* mean, float* cov, void* ws, void* stream);
/* the same with iwvi_layer_desc.flags bits.  IWVI_LAYER_F64_STAGE1: a = Lm^-1 k comes from the float64 route, and the covariance
 * k(x_i, x_j) - a_i . a_j + u_i . u_j is differenced and accumulated in float64 before it is rounded to the float32 output. */
int iwvi_gp_layer_fullcov_ex(const void* state, int M, int D, int R, int kern_type, float variance,
                             const float* F, int64_t S, int64_t N,
                             int mf_type, const float* mf_A, const float* mf_b,
                             float* mean, float* cov, void* ws, int layer_flags, void* stream);

/* The jointly Gaussian sample of the full-covariance branch (temp_workaround.py:93-96, with the intended
 * fmean_SRN1 of :95):  sample[s, n, r] = mean[s, n, r] + (chol(cov[s, r] + jitter I) z[s, r])[n].
 * mean, sample [S, N, R]; cov [S, R, N, N]; z [S, R, N].  N <= 192 (N is the importance-sample axis on the IW path)
 * factorises in LDS; larger N needs ws = iwvi_mvn_sample_ws_bytes(S, N, R) bytes of scratch (0 for N <= 192) and is
 * latency-bound.  Rounding-tolerant PSD factorisation: the block is a float32 difference k - a.a + u.u and (near-)singular
 * blocks (X tiled over K: rank 1) are indefinite by rounding, so a non-positive pivot zeroes its column instead of
 * failing like tf.cholesky, and entries are clamped to |L_ij| <= sqrt(C_ii); on a well-conditioned block this is chol(). */
size_t iwvi_mvn_sample_ws_bytes(int64_t S, int N, int R);
int iwvi_mvn_sample(const float* mean, const float* cov, const float* z, float* sample,
                    int64_t S, int N, int R, float jitter, void* ws, void* stream);

/* ------------------------------------------------------------------------
 * The whole layer stack of DGP_VI.propagate (models.py:31-46) for a flattened sample batch in ONE
 * launch, ending in the per-sample log-weight of models.py:134-142.  Every sample's path through the
 * layers is independent of the other samples, so a workgroup keeps its chunk of samples in LDS from
 * the tiled input (models.py:113-116) to the Gaussian variational expectation (:134); only the
 * optional per-layer outputs and the log-weights [T] touch HBM.
 *
 * Row t of the batch stands for data row (t / row_div) % row_mod of X / XY / Y:
 *   IW tiling [B, K, .] (models.py:113-116): row_div = K, row_mod = B;  VI tiling tile(X, [S, 1])
 *   (:50-53): row_div = 1, row_mod = N;  an explicit [T, .] input: row_div = 1, row_mod = T.
 * layers[i].type selects GPLayer (layers.py:35-50; state from iwvi_gp_precompute) or
 * LatentVariableLayer (layers.py:72-105, encoder :137-152).  Per layer:
 *   noise      [T, R] (GP) / [T, latent_dim] (LV) N(0,1) draws, or NULL: drawn in-kernel from the
 *              counter-based stream (seed, step, layer, t) documented in DESIGN.md
 *   noise_out  optional: the draws that were used
 *   sample / mean / var   optional [T, P] (GP) or [T, D + latent_dim] (LV) outputs
 *   kl_local   LV only, optional [T, latent_dim]: log q(W) - log p(W) (sampled_kl) or the analytic KL
 * rng_state: 2 device words {step counter, ticket}, zeroed once by the caller; the launch reads the step
 *   and its last workgroup advances it, so a replayed hipGraph draws fresh noise.  May be NULL when every
 *   layer has explicit noise.  The ticket word is the launch's arrival counter (every workgroup adds to it once,
 *   the last one leaves it at 0): two launches that may overlap in time need their own rng_state.
 * Y / out_logw may be NULL (propagate only).  out_logw [T] = sum_d var_exp - sum local regularisers.
 * ---------------------------------------------------------------------- */
enum { IWVI_LAYER_GP = 0, IWVI_LAYER_LV = 1 };
#define IWVI_MAX_STACK 8    /* layers (GP + LV) in one fused launch */

typedef struct iwvi_layer_desc {
    int32_t type;                   /* IWVI_LAYER_*                                          */
    /* GP layer */
    const void* state;              /* precomputed iwvi_gp_desc.state                        */
    int32_t M, D, R, P;             /* inducing points, input dim, latent GPs, outputs       */
    int32_t kern_type, mf_type;     /* IWVI_KERN_*, IWVI_MF_*                                */
    float variance;
    const float* W;                 /* [P, R] SharedMixedMok.W or NULL (then P == R)         */
    const float* mf_A;              /* [D, P] for IWVI_MF_LINEAR                             */
    const float* mf_b;              /* [P] or NULL                                           */
    /* LV layer (D = input dim as above) */
    const float* const* enc_W;      /* host array of n_enc device pointers, or NULL = prior  */
    const float* const* enc_b;
    const int32_t* enc_dims;        /* host: n_enc + 1 widths, [0] = XY width, [n] = 2*latent */
    int32_t n_enc, latent_dim, sampled_kl;
    const float* enc_out;           /* [data rows, 2*latent_dim] from iwvi_model_precompute: used instead of enc_W    */
    /* both */
    const float* noise;             /* explicit N(0,1) draws, or NULL                        */
    int32_t zero_noise;             /* noise == NULL: 1 -> z = 0 (sample == mean), 0 -> draw in-kernel */
    float* noise_out;
    float* sample; float* mean; float* var;
    float* kl_local;
    float* a_out; float* u_out;     /* GP, optional (what the adjoint needs): A [T, Mp], L_r^T A [R, T, Mp] */
    float* gmv_out;                 /* GP, optional: [T, 3R] = (sample | mean | variance) of the R latent GPs before mixing */
    const float* variance_dev;      /* GP, optional device scalar read instead of `variance` (see iwvi_gp_desc) */
    int32_t enc_act;                /* LV: IWVI_ACT_* of the encoder's hidden layers (0 = tanh) */
    int32_t flags;                  /* GP: IWVI_LAYER_* bits below (0 = the default arithmetic) */
} iwvi_layer_desc;
/* Arithmetic of the R * M^2 contraction u_r = tril(q_sqrt_r)^T a (temp_workaround.py:78) and of mean = q_mu^T a (:68), per CALL
 * (no process-wide mode: two threads may differ).  Default: split-f16 operands (x = h1 + h2, three v_mfma_f32_16x16x32_f16 per slab,
 * fp32 accumulate, per-matrix power-of-two scales; 22 operand mantissa bits -- DESIGN.md section 4) whenever every GP layer of the
 * launch has an even number of 16-row blocks.  IWVI_LAYER_F32_STAGE2 on ANY GP layer of a launch makes the whole launch take the
 * fp32-MFMA variant (v_mfma_f32_16x16x4_f32). */
#define IWVI_LAYER_F32_STAGE2 1
/* iwvi_layer_desc.flags, per GP layer: K_uf, a = Lm^-1 k and sigma^2 - |a|^2 of THIS layer in float64 (v_mfma_f64_16x16x4_f64 against the
 * dense float64 Lm^-1; temp_workaround.py:44,51,59 -- the reference computes all of it in float64), a rounded to float32 only behind the
 * solve.  For ill-conditioned K_uu (many inducing points in a 1-3-dimensional box: cond(Lm) ~ 1e4, where a float32 k alone costs 5e-4 of
 * the mean and the float32 solve 1e-2 .. 1e-1) it brings the per-layer mean / variance to ~1e-6 of the float64 reference; it costs a
 * float64 Gram and M^2 float64 MFMA FLOPs per sample (half the fp32-MFMA rate), and the launch then runs stage 2 on fp32 MFMAs.
 * The layer's state must have been precomputed with IWVI_GP_F64_STAGE1. */
#define IWVI_LAYER_F64_STAGE1 2

/* Optional tail of the same launch: the last workgroup to finish performs models.py:138-150 on out_logw
 * (logsumexp over K minus log K, or the mean over S of :84; sum over points * scale; minus the global KLs),
 * so that one IW-ELBO evaluation is two launches (iwvi_gp_precompute + iwvi_dgp_forward).  Fields as the
 * arguments of iwvi_logw_reduce; needs rng_state (its second word is the arrival ticket). */
typedef struct iwvi_elbo_desc {
    int64_t B; int32_t K;
    int64_t stride_b, stride_k;
    const double* const* kl_global; const int32_t* kl_global_counts; int32_t n_glob;
    double scale; int32_t K_total, mode_vi;
    float* out_lse_ms; float* out_logp; double* out_elbo;
    double* ws;                     /* optional scratch, ceil(T/16) doubles, used when every point's K samples sit in one chunk (NULL -> the
                                     * last workgroup reads all of out_logw): one partial sum per workgroup; with out_elbo given and at
                                     * most 511 workgroups the partial sums travel in the workgroups' tickets instead (a 64-bit atomic on
                                     * rng_state[1]) and ws only holds, tagged with the evaluation's number, the partial sums too large
                                     * for that fixed-point field.  Contents are scratch either way */
    /* when the leading LatentVariableLayer was evaluated by iwvi_model_precompute (iwvi_enc_desc.sample_X): its local
     * regulariser per sample [T] (subtracted from the log-weights like models.py:141-142), and the position of this
     * stack's first layer in the model, so that its noise streams do not collide with that layer's; x_per_sample:
     * X is that layer's output [T, Dx], one row per sample (Y and XY keep the row_div / row_mod mapping). */
    const float* lw_init;
    int32_t noise_layer_base;
    int32_t x_per_sample;
    const float* lik_variance_dev;  /* optional device scalar read instead of iwvi_dgp_forward's lik_variance argument */
    /* Optional: the heads of the bound's adjoint from the same launch (what iwvi_iw_elbo_backward computes from the final layer's moments --
     * two launches less in front of the first layer adjoint of a value + gradient evaluation): adj_w [T] = d ELBO / d L_nk (scale x the
     * softmax over the K samples of models.py:146-148), adj_dmean / adj_dvar [T, Dy] = d ELBO / d final mean / variance, adj_sums [3] =
     * (sum_n (lse_n - log K), d ELBO / d lik_variance, the bound), written by the last workgroup.  All four or none.  Importance-weighted
     * bound only (mode_vi = 0), every point's K samples contiguous (stride_k = 1, stride_b = K) and inside one chunk of the launch
     * (K divides 16 x the launch's sub-tile count), ws given with room for two doubles per chunk, no K-sharded exchange (the weights of a
     * sharded job need the job-wide logsumexp: iwvi_iw_elbo_backward's lse_global).  Otherwise the call returns IWVI_ERR_UNSUPPORTED
     * before anything is launched -- fall back to iwvi_iw_elbo_backward. */
    float* adj_w; float* adj_dmean; float* adj_dvar; double* adj_sums;
} iwvi_elbo_desc;

int iwvi_dgp_forward(const iwvi_layer_desc* layers_host, int n_layers,
                     const float* X, int Dx, const float* XY, int XYdim, const float* Y, int Dy,
                     int64_t T, int64_t row_div, int64_t row_mod, float lik_variance,
                     uint64_t seed, uint64_t* rng_state, float* out_logw,
                     const iwvi_elbo_desc* elbo /* or NULL */, void* stream);

/* ----------------------------------------------------------------------
 * Backward pass (the reference gets these from TensorFlow's autodiff of the graph of
 * models.py:112-150, experiments/build_models.py:284-304).  Layer by layer; RBF and Matern-5/2 kernels.
 *
 * iwvi_gp_layer_backward: adjoint of iwvi_gp_layer_forward for T samples.
 *   state                  as precomputed WITH IWVI_GP_WANT_DENSE (the dense float64 Lm and Lm^-1 are read)
 *   F [T, D]               the layer's input rows (per sample)
 *   noise [T, R]           the draws the forward used (needed when d_sample is given)
 *   A [T, Mp], U [R, T, Mp]  a = Lm^-1 k and u_r = L_r^T a as the forward wrote them (a_out / u_out).  U may be NULL when
 *                          iwvi_gp_layer_backward_needs_u says 0 (M a multiple of 16 up to 512, T a multiple of 16 -- of 32 beyond
 *                          M = 256 --, the tiles within the LDS) and GMV is given: that shape takes the streaming
 *                          chain, which works from a alone (sum_r 2dv_r L_r u_r = sum_r 2dv_r (L_r L_r^T) a, dL_r = tril(G_r L_r))
 *   GMV [T, 3R]            optional, the forward's gmv_out
 *   d_sample/d_mean/d_var [T, P]  upstream gradients, any may be NULL (= 0)
 *   kl_weight              the objective contains -kl_weight * KL[q(u)||p(u)] of this layer (1 for the ELBO)
 *   outputs (any may be NULL): dF [T, D], dZ [M, D], dls [D], dvariance [1], dq_mu [M, R],
 *                          dq_sqrt [R, M, M] (lower triangle; zeros above), dW [P, R], dmf_A [D, P]
 *   ws                     iwvi_gp_layer_backward_ws_bytes(T, M, D, R) bytes
 * ---------------------------------------------------------------------- */
typedef struct iwvi_gp_bwd_desc {
    const void* state;
    const float* Z; const float* lengthscales; const float* q_mu; const float* q_sqrt;
    float variance;
    int32_t M, D, R, P, kern_type;
    const float* W; int32_t mf_type; const float* mf_A;
    const float* F; const float* noise; const float* A; const float* U;
    const float* GMV;               /* optional: the forward's gmv_out [T, 3R]; spares the adjoint a pass over A and U */
    const float* d_sample; const float* d_mean; const float* d_var;
    double kl_weight;
    float* dF; float* dZ; float* dls; float* dvariance; float* dq_mu; float* dq_sqrt;
    float* dW; float* dmf_A;        /* optional: [P, R] (SharedMixedMok.W), [D, P] (Linear mean function A) */
    void* side_stream;              /* optional hipStream_t: once dF is queued on `stream`, the parameter gradients of this
                                     * layer are queued there (after an event), so that they overlap the adjoint of the layer
                                     * below; the caller joins the two streams before reading the parameter gradients */
    void* side_stream2;             /* optional second side stream: the parameter branch then runs as two concurrent chains
                                     * (Cholesky adjoint | the other sums over samples); join both */
    int32_t prepared;               /* nonzero: iwvi_gp_layer_backward_prepare has already run on this (desc, ws) */
    const float* variance_dev;      /* optional device scalar read instead of `variance` (see iwvi_gp_desc) */
    int32_t phase;                  /* 0: the whole adjoint.  1: only the per-sample chain (dF and the partial sums are queued);
                                     * 2: only the parameter branch of a call made with phase 1 on the same descriptor and
                                     * workspace.  The caller orders 2 after 1 and may queue other work in between (so that, in a
                                     * captured graph, the next layer's chain follows this one's on the same hardware queue).
                                     * Shapes off the streaming chain do everything in phase 1; phase 2 is then a no-op. */
    int32_t flags;                  /* IWVI_BW_* bits below (0 = the default arithmetic) */
} iwvi_gp_bwd_desc;
/* IWVI_BW_F32_CHAIN: the adjoint chain's S_r products on fp32 MFMAs instead of split-f16 operands (per call, like IWVI_LAYER_F32_STAGE2). */
#define IWVI_BW_F32_CHAIN 1
/* IWVI_BW_OWN_QSCALE (ABI 17; read by the two prepare entries): the q(u) scales in the state's constant block may be stale -- the state
 * was last precomputed with IWVI_GP_FACTOR_ONLY, or q(u) has moved since --: take max |L_r| from q_sqrt itself (same value, same
 * rounding; +2 us of the launch).  The preparation then depends on the state's factorisation only, not on a precompute of the current q(u). */
#define IWVI_BW_OWN_QSCALE 4
/* (both sizing entries answer for either arithmetic mode a later call may select through desc.flags: the larger workspace; u is needed
 * when either mode's adjoint reads it) */
size_t iwvi_gp_layer_backward_ws_bytes(int64_t T, int M, int D, int R);
/* 1 if the adjoint of this layer shape takes the GEMM path (in either arithmetic mode) and therefore needs the forward's u_out, 0 if the
 * streaming chain (which works from a_out alone) will run */
int iwvi_gp_layer_backward_needs_u(int64_t T, int M, int D, int R, int P);
/* the parameter-only part of the adjoint (scaled inducing inputs, float32 Lm^-1, the streaming chain's packed operands
 * S_r = L_r L_r^T and Lm^-T): reads state (dense factors), Z, lengthscales, q_sqrt, M / D / R of the descriptor only, so it can
 * be queued on another stream beside the forward; then set desc.prepared.  Called implicitly otherwise. */
int iwvi_gp_layer_backward_prepare(const iwvi_gp_bwd_desc* desc, int64_t T, void* ws, void* stream);
/* The same for n layers (n <= IWVI_MAX_STACK) in ONE launch: descs[i] with its workspace ws[i]; every descriptor must ask for the same
 * arithmetic mode (flags & IWVI_BW_F32_CHAIN), else IWVI_ERR_ARG.  (The prepare steps of a model are queued beside the layer kernel,
 * which holds every CU; as separate launches they run one after the other once it retires.) */
int iwvi_gp_layers_backward_prepare(const iwvi_gp_bwd_desc* descs, int n, int64_t T, void* const* ws, void* stream);
/* Outputs left NULL are not formed.  With ONLY dq_mu / dq_sqrt given (what the natural-gradient op of build_models.py:288-295 reads)
 * the call reduces to the heads and the two sums over samples behind those gradients (on either path: the streaming chain, or the
 * GEMMs over a_out / u_out): no prepare step, no dense factors in `state`, no kernel adjoint, no adjoint of the factorisation; the
 * values are bit-identical to the full call's. */
int iwvi_gp_layer_backward(const iwvi_gp_bwd_desc* desc, int64_t T, void* ws, void* stream);

/* Adjoint of the ELBO tail (models.py:134-150) for the IW tiling (sample t = b*K + k):
 *   L_nk = sum_dy varexp(Y; fmean, fvar) - sum_i sum_q kl_local[i][t, q];  ELBO = scale * sum_n (lse_k L_nk - log K) - KL.
 * out_w [T] = d ELBO / d L_nk (scale * softmax over k), d_mean / d_var [T, Dy] = heads of the final layer,
 * out_sums[0] = sum_n (lse - log K), out_sums[1] = d ELBO / d lik_variance, out_sums[2] = the bound itself
 * (scale * out_sums[0] - the sum of the kl_global arrays, as iwvi_elbo_desc);  ws: 2*B doubles. Outputs may be NULL
 * except out_sums.  lse_global [B] (or NULL): K-sharded training -- the logsumexp of every point over ALL the job's
 * K_total samples (after the exchange of iwvi_lse_merge); the weights are then exp(L - lse_global), this rank's share of the
 * softmax, and out_sums[0] / [2] are the job's.  mode_vi != 0: the bound of DGP_VI (models.py:84: mean over the samples instead of the logsumexp). */
int iwvi_iw_elbo_backward(const float* fmean, const float* fvar, const float* Y, int Dy,
                          const float* const* kl_local, const int32_t* kl_dims, int n_local,
                          int64_t B, int K, float lik_variance, double scale, int mode_vi,
                          float* out_w, float* d_mean, float* d_var,
                          const double* const* kl_global, const int32_t* kl_global_counts, int n_glob,
                          const float* lse_global, int K_total,
                          double* out_sums, double* ws, void* stream);
/* the same with the likelihood variance read from device memory when the launch runs (lik_variance_dev != NULL) */
int iwvi_iw_elbo_backward_dev(const float* fmean, const float* fvar, const float* Y, int Dy,
                              const float* const* kl_local, const int32_t* kl_dims, int n_local,
                              int64_t B, int K, float lik_variance, const float* lik_variance_dev, double scale, int mode_vi,
                              float* out_w, float* d_mean, float* d_var,
                              const double* const* kl_global, const int32_t* kl_global_counts, int n_glob,
                              const float* lse_global, int K_total,
                              double* out_sums, double* ws, void* stream);

/* Adjoint of the LatentVariableLayer (layers.py:83-103): mu, sigma [B, latent_dim] (the encoder's outputs per data
 * row), noise [T, latent_dim] (the draws), dF_next [T, ld_next] = gradient w.r.t. the layer's output rows (columns
 * col0 .. col0+latent_dim-1 are W's; may be NULL), w [T] = d ELBO / d L_nk (the regulariser enters with -1; may be
 * NULL).  d_enc_out [B, 2*latent_dim] = gradient w.r.t. the encoder's (means | raw) output.  mu / sigma rows have stride
 * ld_enc; sigma_is_raw != 0: `sigma` holds the encoder's raw output (sigma = softplus(raw - 3)), so that the [B, 2*latent_dim]
 * block iwvi_model_precompute leaves can be passed as (out, out + latent_dim, 2*latent_dim, 1). */
int iwvi_lv_layer_backward(const float* mu, const float* sigma, int ld_enc, int sigma_is_raw, const float* noise,
                           const float* dF_next, int ld_next, int col0, const float* w,
                           int latent_dim, int64_t B, int K, int sampled_kl, float* d_enc_out, void* stream);

/* Adjoint of the Encoder MLP (layers.py:137-152): d_out [rows, dims[n_enc]] -> dW[i] [dims[i], dims[i+1]], db[i]. */
size_t iwvi_encoder_backward_ws_bytes(int64_t rows, const int32_t* dims, int n_enc);
int iwvi_encoder_backward(const float* XY, int64_t rows, const float* const* enc_W, const float* const* enc_b,
                          const int32_t* dims, int n_enc, const float* d_out,
                          float* const* dW, float* const* db, void* ws, void* stream);
/* the same for an encoder whose hidden layers use activation IWVI_ACT_* */
int iwvi_encoder_backward_act(const float* XY, int64_t rows, const float* const* enc_W, const float* const* enc_b,
                              const int32_t* dims, int n_enc, int act, const float* d_out,
                              float* const* dW, float* const* db, void* ws, void* stream);

/* iwvi_lv_layer_backward followed by iwvi_encoder_backward_act, as one launch plus the reduction (the training step's form: the
 * workgroup that back-propagates eight data rows through the encoder forms their d(means | raw) itself; same arithmetic, same sums).
 * dims[n_enc] must be 2 * latent_dim; XY [B, dims[0]]. */
int iwvi_lv_encoder_backward(const float* mu, const float* sigma, int ld_enc, int sigma_is_raw, const float* noise,
                             const float* dF_next, int ld_next, int col0, const float* w,
                             int latent_dim, int64_t B, int K, int sampled_kl,
                             const float* XY, const float* const* enc_W, const float* const* enc_b,
                             const int32_t* dims, int n_enc, int act,
                             float* const* dW, float* const* db, void* ws, void* stream);

/* Test log-likelihood of experiments/run_conditional_density_estimation.py:148-169, batched over the test points:
 * samples: S predictive draws per point (element (s, n) at samples[s*sample_stride + n*point_stride]); y [N].
 * Gaussian KDE with Silverman's bandwidth 1.06 std S^(-1/5) (:158-162) -> out_logp [N]; squared error of the
 * sample mean (:165) -> out_sqerr [N]; optional out_mean_std [N, 2].  Outputs may be NULL. */
int iwvi_kde_loglik(const float* samples, int64_t sample_stride, int64_t point_stride, const float* y,
                    int64_t N, int S, float* out_logp, float* out_sqerr, float* out_mean_std, void* stream);

/* Optimiser steps of experiments/build_models.py:284-304.
 * iwvi_natgrad_step: GPflow NatGradOptimizer (natural parameterisation) on a whitened (q_mu [M, R], q_sqrt [R, M, M]),
 * in place, float64 inside; dq_mu / dq_sqrt = gradients of the ELBO (the objective that is maximised).
 * iwvi_adam_step: TensorFlow AdamOptimizer on GPflow's unconstrained variables; transform 0 = identity,
 * 1 = positive (param = softplus(x) + 1e-6), | IWVI_ADAM_GRAD_F64: `grad` points to doubles (the likelihood-variance
 * gradient leaves iwvi_iw_elbo_backward in float64); x/m/v are the optimiser's state (same length as param);
 * init != 0 fills them from the current parameter values instead of stepping; t = 1-based step count. */
size_t iwvi_natgrad_ws_bytes(int M);
int iwvi_natgrad_step(float* q_mu, float* q_sqrt, const float* dq_mu, const float* dq_sqrt,
                      int M, int R, double gamma, void* ws, void* stream);
/* ABI 16: the same with the workspace sized for R latent GPs.  M <= 128: the step runs as five launches spread over the chip (their block
 * images take ~340 KB per latent GP at M = 128) when `ws_bytes` holds them for all R, else -- as iwvi_natgrad_step does beyond the R its
 * R-independent size happens to cover -- as one workgroup per latent GP: the same result, ~1.7x the time.  iwvi_natgrad_ws_bytes_ex(M, R)
 * is the size that always takes the spread route; iwvi_debug_last_natgrad_route(): 0 = multi-launch (M > 128), 1 = one workgroup per
 * latent GP, 2 = spread. */
size_t iwvi_natgrad_ws_bytes_ex(int M, int R);
int iwvi_natgrad_step_ex(float* q_mu, float* q_sqrt, const float* dq_mu, const float* dq_sqrt,
                         int M, int R, double gamma, void* ws, size_t ws_bytes, void* stream);
int iwvi_debug_last_natgrad_route(void);
#define IWVI_ADAM_GRAD_F64 16
typedef struct iwvi_adam_tensor {
    float* param; const float* grad; float* x; float* m; float* v; int64_t n; int32_t transform;
} iwvi_adam_tensor;
int iwvi_adam_step(const iwvi_adam_tensor* tensors_host, int n_tensors, double lr, double beta1, double beta2,
                   double eps, int64_t t, int maximise, int init, void* stream);
/* the same with the step count on the device: the launch uses *t_dev + 1 and then stores it back (one tiny extra launch),
 * so a captured hipGraph of a training step stays valid from step to step */
int iwvi_adam_step_dev(const iwvi_adam_tensor* tensors_host, int n_tensors, double lr, double beta1, double beta2,
                       double eps, int64_t* t_dev, int maximise, void* stream);

/* models.py:138-150 on precomputed log-weights: logw row of (point b, sample k) = b*stride_b + k*stride_k;
 * arguments as iwvi_iw_elbo_reduce. */
int iwvi_logw_reduce(const float* logw, int64_t B, int K, int64_t stride_b, int64_t stride_k,
                     const double* const* kl_global_host, const int32_t* kl_global_counts_host, int n_glob,
                     double scale, int K_total, int mode_vi,
                     float* out_lse_ms, float* out_logp, double* out_elbo, uint64_t* ticket, void* stream);

/* ------------------------------------------------------------------------
 * LatentVariableLayer forward alone (layers.py:72-105) with its Encoder MLP (:137-152).
 *   F   [T, D];  XY [T, XYdim] or NULL (prior mode, :73-81);  noise [T, Lw] or NULL (z = 0)
 *   enc_W[i] [dims[i], dims[i+1]], enc_b[i] [dims[i+1]], dims_host[n_enc+1],
 *   dims[0] = XYdim, dims[n_enc] = 2*Lw; tanh on all but the last layer, skip
 *   connection where dims[i] == dims[i+1]; q_sqrt = softplus(raw - 3).
 *   sample/mean/cov [T, D+Lw] (any may be NULL), kl [T, Lw]:
 *   sampled_kl != 0 -> log q(W) - log p(W) (:98-100) else analytic KL (:101-103).
 * (The IW path runs this layer inside iwvi_dgp_forward, once per data point.)
 * ---------------------------------------------------------------------- */
int iwvi_lv_layer_forward(const float* F, const float* XY, const float* noise,
                          const float* const* enc_W_host, const float* const* enc_b_host,
                          const int32_t* dims_host, int n_enc,
                          int D, int Lw, int sampled_kl,
                          float* sample, float* mean, float* cov, float* kl,
                          int64_t T, void* stream);
/* the same for an encoder whose hidden layers use activation IWVI_ACT_* (Encoder(activation_func=...), layers.py:109) */
int iwvi_lv_layer_forward_act(const float* F, const float* XY, const float* noise,
                              const float* const* enc_W_host, const float* const* enc_b_host,
                              const int32_t* dims_host, int n_enc, int act,
                              int D, int Lw, int sampled_kl,
                              float* sample, float* mean, float* cov, float* kl,
                              int64_t T, void* stream);

/* ------------------------------------------------------------------------
 * The IW-ELBO reduction (models.py:133-150): Gaussian variational expectations
 * (:134), sum over Dy (:138), minus local regularisers (:140-142), logsumexp over
 * K minus log K (:148), sum over points * scale minus global KLs (:150).
 *   fmean, fvar: Dy-wide rows, row of (point b, sample k) = b*stride_b + k*stride_k
 *                (IW tiling [B,K,Dy]: stride_b = K, stride_k = 1; VI tiling [S*N,Dy]: 1, N);
 *                fvar = diagonal variances;  Y [B, Dy]
 *   kl_local[i] rows of kl_dims[i] floats, same row indexing, for i < n_kl
 *   kl_global[i]: pointer to kl_global_counts[i] doubles (state.kl of GP layer i: its R KL shares;
 *                 counts NULL -> 1 each), n_glob of them
 *   out_lse_ms [B, 2] = (max_k L, sum_k exp(L - max)) per point, for K-sharded merging
 *   out_logp   [B]     logsumexp - log(K_total)   (K_total = K when not sharded)
 *   out_elbo   [1] double = sum(logp) * scale - sum(kl_global)
 *   ticket     [1] device word, zeroed ONCE by the caller (never per call): the last workgroup to finish
 *              performs the final sum and re-zeroes it, so the whole reduction is one launch; needed when
 *              out_elbo != NULL
 * Any out pointer may be NULL.  mode_vi != 0 -> reduce_mean over K instead (models.py:84).
 * ---------------------------------------------------------------------- */
int iwvi_iw_elbo_reduce(const float* fmean, const float* fvar, const float* Y,
                        float lik_variance, int64_t B, int K, int Dy,
                        int64_t stride_b, int64_t stride_k,
                        const float* const* kl_local_host, const int32_t* kl_dims_host, int n_kl,
                        const double* const* kl_global_host, const int32_t* kl_global_counts_host, int n_glob,
                        double scale, int K_total, int mode_vi,
                        float* out_lse_ms, float* out_logp, double* out_elbo, uint64_t* ticket, void* stream);

/* ABI 16: the same with the likelihood variance read from a 1-element device tensor when `lik_variance_dev` is not NULL (a TRAINED
 * likelihood variance lives on the device: a captured graph of a K-sharded training step stays valid while it changes, and no
 * device-to-host copy sits in the step); `lik_variance` is then only validated (> 0). */
int iwvi_iw_elbo_reduce_dev(const float* fmean, const float* fvar, const float* Y,
                            float lik_variance, const float* lik_variance_dev, int64_t B, int K, int Dy,
                            int64_t stride_b, int64_t stride_k,
                            const float* const* kl_local_host, const int32_t* kl_dims_host, int n_kl,
                            const double* const* kl_global_host, const int32_t* kl_global_counts_host, int n_glob,
                            double scale, int K_total, int mode_vi,
                            float* out_lse_ms, float* out_logp, double* out_elbo, uint64_t* ticket, void* stream);

/* Merge K-sharded partials after the RCCL exchange (not in the reference):
 *   ms_all [G, B, 2] gathered (max, sumexp) pairs -> logp [B], elbo [1] as above. */
int iwvi_lse_merge(const float* ms_all, int G, int64_t B, int K_total,
                   const double* const* kl_global_host, const int32_t* kl_global_counts_host, int n_glob,
                   double scale, float* out_logp, double* out_elbo, void* stream);

/* The same for n_steps independent evaluations exchanged in ONE all-gather (forward-only evaluation loops amortise
 * the latency-bound collective this way): ms_all [G, n_steps, B, 2] -> logp [n_steps, B] (or NULL), elbo [n_steps]. */
int iwvi_lse_merge_steps(const float* ms_all, int G, int n_steps, int64_t B, int K_total,
                         const double* const* kl_global_host, const int32_t* kl_global_counts_host, int n_glob,
                         double scale, float* out_logp, double* out_elbo, void* stream);

/* whitened gauss_kl alone (temp_workaround.py:186-188), K14: -> kl [1] double */
int iwvi_gauss_kl(const float* q_mu, const float* q_sqrt, int M, int R, double* kl, void* stream);

/* gpflow Gaussian.variational_expectations as a callable (the reference calls it on explicit moments,
 * models.py:66,134):  out[t, d] = -1/2 log 2pi - 1/2 log s2 - 1/2 ((Y[row(t), d] - Fmu[t, d])^2 + Fvar[t, d]) / s2,
 * row(t) = (t / row_div) % row_mod (the tilings of iwvi_dgp_forward; Y already tiled: row_div = 1, row_mod = T).
 * Fmu, Fvar, out [T, Dy].  (The hot path never calls this: the expectation is fused into iwvi_dgp_forward's tail.) */
int iwvi_gaussian_var_exp(const float* Fmu, const float* Fvar, const float* Y, float lik_variance,
                          int64_t T, int Dy, int64_t row_div, int64_t row_mod, float* out, void* stream);

/* white=False (temp_workaround.py:63-65: "another backsubstitution in the unwhitened case").  The unwhitened
 * q(u) = N(f, q_sqrt q_sqrt^T) gives the same conditional as the whitened one with f_w = Lm^-1 f and
 * q_sqrt_w[r] = Lm^-1 tril(q_sqrt[r]) (lower triangular again), so the back-substitution is applied ONCE to the
 * operands (float64 accumulate) instead of per sample; the caller then runs iwvi_gp_precompute on (f_w, q_sqrt_w).
 * state: precomputed WITH IWVI_GP_WANT_DENSE for the same (M, R) (the dense Lm^-1 is read);
 * f [M, R], q_sqrt [R, M, M] or NULL -> f_w [M, R], q_sqrt_w [R, M, M]. */
int iwvi_unwhiten(const void* state, int M, int R, const float* f, const float* q_sqrt,
                  float* f_w, float* q_sqrt_w, void* stream);

/* counter-based N(0,1) fill (Philox4x32-10 + Box-Muller); stream documented in DESIGN.md */
int iwvi_fill_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);
/* same stream, but the counter lives on the device: state[0] = counter (read by the launch, then advanced
 * by ceil(n/4) by its last block), state[1] = internal ticket; zero both once. Lets a captured hipGraph
 * draw fresh noise on every replay. */
int iwvi_fill_normal_dev(float* out, int64_t n, uint64_t seed, uint64_t* state, void* stream);

/* Diagnostic / development route switches (NOT part of the drop-in surface, like iwvi_debug_set_stamps): the library itself never reads
 * the environment.  name = one of the IWVI_* route names listed in csrc/abi.hip (e.g. "IWVI_BW_FUSED", "IWVI_NATGRAD_UNFUSED"),
 * value 0 = default route.  Returns 0, or IWVI_ERR_ARG for an unknown name.  Process-wide; not for concurrent use with launches. */
int iwvi_debug_set_option(const char* name, int value);
/* Which instantiation of the fused forward kernel the LAST iwvi_dgp_forward call of this process launched (tests assert that a shape
 * takes / does not take the compiled-in-shapes variants): bits 0-7 sub-tiles per workgroup (NS), bit 8 split-f16 stage 2, bit 9 the
 * large-M (M > 128) build, bits 10-11 LEAN mode (0 general, 1 bound-only, 2 with per-layer outputs), bit 12 a layer ran the float64
 * stage-1 route.  0 before the first launch. */
int iwvi_debug_last_forward_variant(void);
/* In-kernel time stamps of the fused forward / of the precompute launch (128 64-bit words per workgroup; NULL switches them off) and an
 * early exit of the fused forward after phase N -- timing scripts only (scripts/stamp_*.py, bench.py's gemm_phase_mfma_util). */
void iwvi_debug_set_stamps(void* buf, int64_t max_workgroups);
void iwvi_debug_set_pre_stamps(void* buf);
void iwvi_debug_set_exit(int phase);

#ifdef __cplusplus
}
#endif
#endif /* IWVI_HIP_H */

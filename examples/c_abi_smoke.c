/* The C-ABI used from plain C (no Python, no torch): one GPLayer (temp_workaround.py:12-98 + SharedMixedMok mixing,
 * :142-145, + linear mean function, layers.py:46-48) on T samples -- iwvi_gp_precompute then iwvi_gp_layer_forward --
 * with inputs given by closed formulas so that tests/test_gpu_c_abi.py can rebuild them for the oracle.
 * Build:  hipcc examples/c_abi_smoke.c -Iinclude -Ldgps_with_iwvi_amd/csrc -liwvi_hip -o examples/c_abi_smoke
 * Output: "mean" and "var" rows of the first 4 samples and float64 checksums, one number per line. */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "iwvi_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define IW(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s: %s (%d)\n", #x, iwvi_last_error(), r_); return 3; } } while (0)

static float* upload(const float* h, size_t n) {
    float* d = NULL;
    if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) return NULL;
    if (hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

int main(void) {
    enum { M = 48, D = 4, R = 3, P = 4, T = 100 };
    static float Z[M * D], ls[D], q_mu[M * R], q_sqrt[R * M * M], F[T * D], noise[T * R], W[P * R], A[D * P];
    for (int i = 0; i < M * D; ++i) Z[i] = (float)sin(0.37 * i + 0.1);
    for (int d = 0; d < D; ++d) ls[d] = 1.0f + 0.25f * d;
    for (int i = 0; i < M * R; ++i) q_mu[i] = (float)cos(0.11 * i);
    for (int r = 0; r < R; ++r)
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < M; ++j)
                q_sqrt[(r * M + i) * M + j] = j > i ? 0.f : (i == j ? 0.5f + 0.01f * r : 0.02f * (float)sin(0.3 * (i + 2 * j + r)));
    for (int i = 0; i < T * D; ++i) F[i] = (float)sin(0.05 * i) * 1.5f;
    for (int i = 0; i < T * R; ++i) noise[i] = (float)cos(0.7 * i);
    for (int i = 0; i < P * R; ++i) W[i] = 0.3f * (float)sin(1.0 + i);
    for (int d = 0; d < D; ++d) for (int p = 0; p < P; ++p) A[d * P + p] = d == p ? 1.f : 0.f;

    if (iwvi_version() != IWVI_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    float *dZ = upload(Z, M * D), *dls = upload(ls, D), *dqm = upload(q_mu, M * R), *dqs = upload(q_sqrt, R * M * M);
    float *dF = upload(F, T * D), *dn = upload(noise, T * R), *dW = upload(W, P * R), *dA = upload(A, D * P);
    float *ds, *dm, *dv;
    void* state;
    CK(hipMalloc((void**)&ds, T * P * sizeof(float))); CK(hipMalloc((void**)&dm, T * P * sizeof(float))); CK(hipMalloc((void**)&dv, T * P * sizeof(float)));
    CK(hipMalloc(&state, iwvi_gp_state_bytes(M, R)));
    if (!dZ || !dls || !dqm || !dqs || !dF || !dn || !dW || !dA) { fprintf(stderr, "upload failed\n"); return 2; }

    iwvi_gp_desc g;
    memset(&g, 0, sizeof(g));                              /* optional fields (variance_dev) = NULL */
    g.Z = dZ; g.lengthscales = dls; g.q_mu = dqm; g.q_sqrt = dqs; g.state = state;
    g.variance = 1.3f; g.jitter = 1e-6; g.M = M; g.D = D; g.R = R; g.kern_type = IWVI_KERN_RBF; g.flags = 0;
    IW(iwvi_gp_precompute(&g, 1, NULL));
    IW(iwvi_gp_layer_forward(state, M, D, R, P, IWVI_KERN_RBF, 1.3f, dF, dn, dW, IWVI_MF_LINEAR, dA, NULL, ds, dm, dv, T, 1, NULL));
    CK(hipDeviceSynchronize());

    static float hs[T * P], hm[T * P], hv[T * P];
    CK(hipMemcpy(hs, ds, sizeof hs, hipMemcpyDeviceToHost)); CK(hipMemcpy(hm, dm, sizeof hm, hipMemcpyDeviceToHost)); CK(hipMemcpy(hv, dv, sizeof hv, hipMemcpyDeviceToHost));
    double cs = 0, cm = 0, cv = 0;
    for (int i = 0; i < T * P; ++i) { cs += hs[i]; cm += hm[i]; cv += hv[i]; }
    for (int i = 0; i < 4 * P; ++i) printf("%.9g\n", hm[i]);
    for (int i = 0; i < 4 * P; ++i) printf("%.9g\n", hv[i]);
    printf("%.12g\n%.12g\n%.12g\n", cs, cm, cv);
    return 0;
}

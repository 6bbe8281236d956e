#!/usr/bin/env python3
"""Error of the two arithmetic modes of the R * M^2 contraction (split-f16 operands on v_mfma_f32_16x16x32_f16 -- the default -- and
fp32 MFMAs, iwvi_layer_desc.flags & IWVI_LAYER_F32_STAGE2) against the float64 oracle, at the stacks of BASELINE.json configs[2]/[3]/[4]
with the batch cut to what the oracle evaluates in seconds (the arithmetic per sample does not depend on B).
    python scripts/split16_error.py > profiles/r03_split16_error.txt"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgps_with_iwvi_amd import settings, synthetic
from oracle.from_spec import build_oracle, oracle_noise

dev = torch.device("cuda:0")
CASES = [("configs[2] L=2 M=128 K=20 +LV", dict(L=2, M=128, K=20, B=48, with_lv=True)),
         ("configs[3] L=3 M=256 K=50", dict(L=3, M=256, K=50, B=8, with_lv=False)),
         ("configs[4] L=5 M=512 K=100", dict(L=5, M=512, K=100, B=2, with_lv=False))]
print("%-32s %-10s %12s %12s %14s" % ("stack (B cut for the oracle)", "stage 2", "max|d mean|", "max|d var|", "|d ELBO|/|ELBO|"))
for name, cfg in CASES:
    spec = synthetic.make_spec(seed=0, parity=True, n_data=4096, **cfg)
    zs = synthetic.make_noise(spec, seed=1)
    zd = [torch.as_tensor(z, dtype=torch.float32, device=dev) for z in zs]
    om = build_oracle(spec)
    ref = om.build_likelihood(oracle_noise(spec, zs))
    _, _, means_o, covs_o, _ = om.log_weights(oracle_noise(spec, zs))
    for mode, f32 in (("split-f16", False), ("fp32", True)):
        settings.fw_f32_stage2 = f32
        m = synthetic.build_model(spec, dev)
        elbo = float(m.compute_log_likelihood(zd))
        fmean, fvar, _, _, _, means, covs = m._forward_iw(zd)
        dm = max(float(np.abs(mm.double().cpu().numpy() - mo).max()) for mm, mo in zip(means[:-1], means_o[:-1]))
        dm = max(dm, float(np.abs(fmean.double().cpu().numpy() - means_o[-1]).max()))
        vo = np.diagonal(covs_o[-1], axis1=-2, axis2=-1).transpose(0, 2, 1)
        dv = float(np.abs(fvar.double().cpu().numpy() - vo).max())
        print("%-32s %-10s %12.3e %12.3e %14.3e" % (name, mode, dm, dv, abs(elbo - ref) / abs(ref)))
settings.fw_f32_stage2 = False

# the ill-conditioned family: M = 160 .. 256 inducing points in a ONE-dimensional box -- K_uu numerically rank-deficient,
# cond(Lm) >> 1e3; the super-block solve multiplies by explicit 128 x 128 inverses from M = 256 (nbk >= 16), the column-at-a-time solve runs
# below.  Both arithmetic modes against float64; the suite's tolerance for this family is 5e-3 on the ELBO (tests/test_gpu_random_sweep.py).
print()
print("%-32s %-10s %12s %12s %14s" % ("1-D inputs, L=2, K=10, B=16", "stage 2", "max|d mean|", "max|d var|", "|d ELBO|/|ELBO|"))
for M in (160, 192, 224, 256):
    spec = synthetic.make_spec(seed=M, parity=True, n_data=4096, L=2, M=M, K=10, B=16, Dx=1, with_lv=False)
    zs = synthetic.make_noise(spec, seed=1)
    zd = [torch.as_tensor(z, dtype=torch.float32, device=dev) for z in zs]
    om = build_oracle(spec)
    ref = om.build_likelihood(oracle_noise(spec, zs))
    _, _, means_o, covs_o, _ = om.log_weights(oracle_noise(spec, zs))
    for mode, f32 in (("split-f16", False), ("fp32", True)):
        settings.fw_f32_stage2 = f32
        m = synthetic.build_model(spec, dev)
        elbo = float(m.compute_log_likelihood(zd))
        fmean, fvar, _, _, _, means, covs = m._forward_iw(zd)
        dm = max(float(np.abs(mm.double().cpu().numpy() - mo).max()) for mm, mo in zip(means[:-1], means_o[:-1]))
        dm = max(dm, float(np.abs(fmean.double().cpu().numpy() - means_o[-1]).max()))
        vo = np.diagonal(covs_o[-1], axis1=-2, axis2=-1).transpose(0, 2, 1)
        dv = float(np.abs(fvar.double().cpu().numpy() - vo).max())
        print("%-32s %-10s %12.3e %12.3e %14.3e" % ("M = %d" % M, mode, dm, dv, abs(elbo - ref) / abs(ref)))
settings.fw_f32_stage2 = False

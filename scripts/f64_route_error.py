#!/usr/bin/env python3
"""The ill-conditioned 1-D family of scripts/split16_error.py (M = 100 .. 512 inducing points in a one-dimensional box, L = 2, K = 10, B = 16)
with the float64 stage-1 route off (float32 Gram + solve: split-f16 / fp32 stage 2) and on (IWVI_LAYER_F64_STAGE1: K_uf, Lm^-1 k and
sigma^2 - |a|^2 in float64, fp32 stage 2), against the float64 oracle.    python scripts/f64_route_error.py > profiles/r05_f64_route_error.txt"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgps_with_iwvi_amd import _abi, settings, synthetic
from oracle.from_spec import build_oracle, oracle_noise

dev = torch.device("cuda:0")
print("%-34s %-22s %12s %12s %14s %8s" % ("1-D inputs, L=2, K=10, B=16", "route", "max|d mean|", "max|d var|", "|d ELBO|/|ELBO|", "variant"))
for M, Dx in ((100, 1), (128, 1), (160, 1), (192, 1), (224, 1), (256, 1), (384, 1), (512, 1), (128, 2), (256, 2), (256, 3)):
    spec = synthetic.make_spec(seed=M, parity=True, n_data=4096, L=2, M=M, K=10, B=16, Dx=Dx, with_lv=False)
    zs = synthetic.make_noise(spec, seed=1)
    zd = [torch.as_tensor(z, dtype=torch.float32, device=dev) for z in zs]
    om = build_oracle(spec)
    ref = om.build_likelihood(oracle_noise(spec, zs))
    _, _, means_o, covs_o, _ = om.log_weights(oracle_noise(spec, zs))
    for mode, f64, f32 in (("float32, split-f16 stage 2", "off", False), ("float32, fp32 stage 2", "off", True), ("float64 stage 1", "on", False)):
        settings.fw_f32_stage2, settings.f64_stage1 = f32, f64
        m = synthetic.build_model(spec, dev)
        elbo = float(m.compute_log_likelihood(zd))
        var = _abi.lib().iwvi_debug_last_forward_variant()
        fmean, fvar, _, _, _, means, covs = m._forward_iw(zd)
        dm = max(float(np.abs(mm.double().cpu().numpy() - mo).max()) for mm, mo in zip(means[:-1], means_o[:-1]))
        dm = max(dm, float(np.abs(fmean.double().cpu().numpy() - means_o[-1]).max()))
        vo = np.diagonal(covs_o[-1], axis1=-2, axis2=-1).transpose(0, 2, 1)
        dv = float(np.abs(fvar.double().cpu().numpy() - vo).max())
        print("%-34s %-22s %12.3e %12.3e %14.3e %#8x" % ("M = %d, Dx = %d" % (M, Dx), mode, dm, dv, abs(elbo - ref) / abs(ref), var))
settings.fw_f32_stage2, settings.f64_stage1 = False, "auto"

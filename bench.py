#!/usr/bin/env python3
"""IW-ELBO samples/sec on MI355X (BASELINE.json metric) -- one JSON line on rank 0.

A "step" is one full forward IW-ELBO evaluation (``DGP_IWVI._build_likelihood`` equivalent: the
per-step K_uu Gram/Cholesky/inverse of every GP layer, on-device N(0,1) noise, LatentVariableLayer,
all GP layers, Gaussian variational expectations, log-sum-exp over K, scaled sum minus the KLs) on
one minibatch already resident in HBM.  Default workload = BASELINE.json configs[2] (the config the
metric is quoted on): 2-layer DGP + LatentVariableLayer, M=128, K=20, batch=1024, Dx=8, Dy=1, R=5.

  python bench.py [--gpus N --steps K --warmup W] [--config 1..4] [--shard k|n]
      --gpus N > 1 with no WORLD_SIZE in the environment: this process starts the N rank processes itself (launch_ranks:
      before anything touches the GPU), relays rank 0's JSON line and exits with the first non-zero rank's code
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...      (the same ranks from a launcher)

Multi-GPU (weak scaling, per-GPU work fixed):
  --shard k (default): every rank holds the same B points and draws its own K importance samples
      (global K = N*K); per step one RCCL all-gather of the per-point (max, sum-exp) pairs [B, 2]
      and a merge kernel give the global log-sum-exp and ELBO (mode i);
  --shard n: every rank owns B different points and all K samples; one scalar all-reduce per step.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this image needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails in the legacy mode); read when HIP initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import dgps_with_iwvi_amd  # noqa: F401,E402  (first: the package sets HIP_FORCE_DEV_KERNARG=1 before HIP initialises -- the product's own environment)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CONFIGS = {   # BASELINE.json configs[1..4]
    1: dict(L=2, M=128, K=5, B=1024, with_lv=False),
    2: dict(L=2, M=128, K=20, B=1024, with_lv=True),
    3: dict(L=3, M=256, K=50, B=4096, with_lv=False),
    4: dict(L=5, M=512, K=100, B=8192, with_lv=False),
}
PEAK_MFMA_F32 = 157.3e12     # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_MFMA_F16 = 2.5e15       # dense f16-input MFMA peak (never the 2:1-sparsity figure)
PEAK_HBM = 8.0e12            # bytes / s


def f_alg_layer(M, D, R, P):
    """Algorithmic FLOPs per sample of one GP layer: Gram + triangular
    Lm^-1 k + R triangular L_r^T a + mean + reductions + mixing/mean function."""
    return (2 * M * D + 3 * M) + M * M + R * M * M + 2 * M * R + (2 * M + 2 * R * M) + 6 * R * P + 2 * D * P


def f_alg_split(spec, f32_stage2=False):
    """The same algorithmic FLOPs per sample split by the matrix instruction that executes them: (fp32-MFMA part, split-f16 part).
    Split-f16 (x = h1 + h2; csrc/dgp_forward.hip): stage 2 (R M^2 + 2 M R) when EVERY GP layer has an even number of 16-row blocks,
    the off-diagonal updates of stage 1 (M^2 - 16 M) of a layer with an even block count <= 8, and (round 4) the dense part of the
    super-block solve of a layer with M > 240 in such a launch.  A split-f16 product costs three
    f16 MFMA FLOPs per algorithmic FLOP, so its ceiling is PEAK_MFMA_F16 / 3."""
    gps = [l for l in spec["layers"] if l["type"] == "gp"]
    nbks = [(l["Z"].shape[0] + 15) // 16 for l in gps]
    s16_all = (not f32_stage2) and all(n % 2 == 0 for n in nbks)
    tot, _ = f_alg_model(spec)
    f16 = 0.0
    for l, nbk in zip(gps, nbks):
        M, R = l["Z"].shape[0], l["q_mu"].shape[1]
        if s16_all:
            f16 += R * M * M + 2 * M * R
        if nbk % 2 == 0 and nbk <= 8:
            f16 += M * M - 16 * M
        if s16_all and nbk >= 16:                                # the dense part of the super-block solve: the blocks -L(I, <I), 512 FLOP per block and sample
            f16 += 512.0 * sum(min(8, nbk - 8 * I) * 8 * I for I in range((nbk + 7) // 8))
    return tot - f16, f16


def f_alg_model(spec):
    tot, per_layer = 0.0, []
    for l in spec["layers"]:
        if l["type"] == "lv":
            f = 2.0 * sum(a * b for a, b in zip(l["dims"][:-1], l["dims"][1:])) + 20
        else:
            M, D = l["Z"].shape
            R = l["q_mu"].shape[1]
            P = l["W"].shape[0] if l["W"] is not None else R
            f = f_alg_layer(M, D, R, P)
        per_layer.append(f)
        tot += f
    return tot + 10, per_layer


class Step:
    """One ELBO evaluation = 2 launches (precompute incl. the encoder; fused layer stack + ELBO tail), capturable into a hipGraph.
    Noise is drawn inside the forward kernel from its counter-based stream; the step counter lives on the
    device and is advanced by the launch itself, so every graph replay sees fresh noise."""

    def __init__(self, model, spec, dev, shard, world, exchange=None, K_total=None):
        self.model, self.dev, self.shard, self.world = model, dev, shard, world
        self.exchange = (world > 1) if exchange is None else exchange
        self.B, self.K = spec["B"], spec["K"]
        self.K_total = K_total if K_total is not None else self.K * self.world
        self.out = torch.zeros(1, dtype=torch.float64, device=dev)

    def run(self, out=None, zs=None):
        """``out``: the exchange staging buffer this evaluation's result goes to ([B, 2] pairs / 1-element ELBO);
        ``zs``: injected noise (the --check leg), None = drawn on the device."""
        m = self.model
        if self.shard == "k" and self.exchange:
            self.ms, self.glob = m.lse_partials(zs, K_total=self.K_total, out=out)
            return self.ms
        self.out = m._build_likelihood(zs, out=out)
        return self.out


def training_leg(model, samples, iters=20, flops_forward=None):
    """Informational, outside the timed region of the headline metric: the same workload through the backward pass
     -- one IW-ELBO value + gradient evaluation, and one training step of the reference
    (NatGrad op + Adam op = two gradient evaluations, experiments/build_models.py:297-300), eager launches."""
    try:
        from dgps_with_iwvi_amd import backward
        from dgps_with_iwvi_amd.training import Trainer

        def timed(fn):
            fn(); fn()
            torch.cuda.synchronize()
            reps = []
            for _ in range(5):                               # median of five batches: one host stall inside a 20-iteration batch
                t0 = time.perf_counter()                     # otherwise shows up as a 2x slower step
                for _ in range(iters):
                    fn()
                torch.cuda.synchronize()
                reps.append((time.perf_counter() - t0) / iters * 1e3)
            return float(np.median(reps))

        grad_ms = timed(lambda: backward.iw_elbo_and_gradients(model))
        # the same evaluation replayed from a hipGraph (what a training loop with use_graph=True runs)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            backward.iw_elbo_and_gradients(model)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                keep = backward.iw_elbo_and_gradients(model)
        torch.cuda.current_stream().wait_stream(side)
        grad_graph_ms = timed(g.replay)
        step_ms = timed(Trainer(model).step)
        step_graph_ms = timed(Trainer(model, use_graph=True).step)
        roof = None
        if flops_forward:
            # algorithmic FLOPs of one value + gradient evaluation: the forward's + two adjoint products per forward
            # product (d operand A, d operand B) = 3 x; priced against the fp32-MFMA peak (the adjoint's products over samples and its
            # Cholesky adjoint are fp32 / fp64 MFMA work; only phase 1 of the chain kernel runs split f16) -- fp32-equivalent, like
            # roofline.frac_fp32_equivalent of the forward
            f = 3.0 * flops_forward
            ach = f / (grad_graph_ms * 1e-3)
            roof = {"bound": "mfma", "kernel": "one value + gradient evaluation = one hipGraph replay (~30 launches; dominant: k_bw_chain per GP layer)",
                    "achieved": ach / 1e12, "peak": PEAK_MFMA_F32 / 1e12, "unit": "TFLOP/s", "frac": ach / PEAK_MFMA_F32,
                    "flops_per_evaluation": f, "traffic": None,
                    "limiter": "launch-bound chains: ~30 dependent launches and 3 cross-queue joins per evaluation (DESIGN.md section 6)"}
            try:
                bj = json.load(open(os.path.join(ROOT, "profiles", "backward_latest.json")))
                roof["kernels_of_the_committed_profile"] = bj["kernels"][:6]
                roof["pmc_profile_of_commit"] = bj.get("pmc_profile_of_commit")
                roof["traffic"] = sum((k.get("hbm_read_bytes") or 0.0) + (k.get("hbm_write_bytes") or 0.0) for k in bj["kernels"]
                                      if (k.get("calls") or 0) > 0 and k.get("hbm_read_bytes") is not None) or None
                roof["traffic_note"] = "sum over the profile's kernels of their per-dispatch HBM bytes (one dispatch of each per evaluation, two of the chain kernel's variants)"
            except Exception:
                pass
        return {"roofline": roof, "value_and_gradient_ms": grad_graph_ms, "value_and_gradient_eager_ms": grad_ms,
                "gradient_samples_per_s": samples / grad_graph_ms * 1e3,
                "train_step_ms": step_graph_ms, "train_step_eager_ms": step_ms,
                "note": "value + gradient: one hipGraph replay of backward.iw_elbo_and_gradients (fused forward that keeps a, streaming adjoint chain "
                        "per layer, Cholesky adjoint; DESIGN.md section 6); train step: training.Trainer(use_graph=True).step = NatGrad op + Adam op, "
                        "ONE graph replay (the synthetic data is not minibatched; two replays otherwise), trained scalars and Adam's step count "
                        "on the device; runs after the timed region and changes the model's parameters"}
    except Exception as e:                                   # the informational leg must not take the headline line down -- but it fails LOUDLY:
        import traceback                                     # the traceback goes to stderr and the JSON carries the error text
        traceback.print_exc()
        return {"error": "%s: %s" % (type(e).__name__, e)}


def sharded_training_leg(model, shard, world, K_total, dist, iters=10):
    """Informational, all ranks (world > 1): one training step of the sharded job -- NatGrad op + Adam op, ONE exchange per op's
    evaluation (what a training loop pays: the equivalent of --xch-every 1) -- eagerly and as hipGraph segments with the collectives
    between them (training.Trainer(use_graph=True): N-shard 3 segments around 2 gradient all-reduces, K-shard 5 around 2 all-gathers of
    the [B, 2] pairs + 2 all-reduces).  Max over ranks of the per-step wall time between barriers.  Changes the model's parameters: runs last."""
    try:
        from dgps_with_iwvi_amd.training import Trainer
        out = {}
        for name, ug in (("train_step_eager_ms", False), ("train_step_ms", True)):
            tr = Trainer(model, shard=shard, use_graph=ug, check_finite=False, K_total=K_total if shard == "k" else None)
            for _ in range(3):
                tr.step()
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                tr.step()
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t = torch.tensor([(time.perf_counter() - t0) / iters * 1e3], dtype=torch.float64, device=model.X.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            out[name] = float(t.item())
            if ug:
                g = tr._graphs["step"][1]
                out["graph_segments"], out["collectives_per_step"] = int(g.n_graphs), int(g.n_collectives)
        out["note"] = ("training.Trainer(shard=%r).step on %d ranks: every op's evaluation exchanges once (xch_every = 1); the graph form replays "
                       "hipGraph segments with the collectives issued between them" % (shard, world))
        return out
    except Exception as e:
        import traceback
        traceback.print_exc()
        return {"error": "%s: %s" % (type(e).__name__, e)}


def cpu_baseline(spec, seconds=12.0, dtype=torch.float64):
    """The reference-equivalent CPU path (oracle/ref_torch_cpu.py; float64 like the reference, and float32 = the
    precision the device path computes in) timed on
    this host's cores on the SAME workload; bounded to ~`seconds` of CPU work.  The thread count is the
    best of {all hardware threads, 1/2, 1/4 of them, 32, 16, 8, 4}: oversubscribing small
    batched matmuls with 256 threads is slower than fewer threads, and the baseline should be the host's best."""
    from dgps_with_iwvi_amd import synthetic
    from oracle.ref_torch_cpu import CpuDGP
    # a BOUNDED sample of the workload: the restatement materialises [S, R, M, N] intermediates (8 B x 3 R M per sample and layer) and
    # executes ~2 R M^2 FLOP per sample -- at configs[4] one full evaluation is 10 TFLOP and tens of GB.  The first B_cpu rows of the
    # minibatch (all K samples each), B_cpu from a ~50 GFLOP/s / 8 GB budget; configs[1] / [2] run whole (B_cpu = B)
    K_ = spec["K"]
    gps_ = [l for l in spec["layers"] if l["type"] == "gp"]
    f_est = sum(2.0 * l["q_mu"].shape[1] * l["Z"].shape[0] ** 2 + l["Z"].shape[0] ** 2 for l in gps_)
    mem_per_row = max(3.0 * l["q_mu"].shape[1] * l["Z"].shape[0] * 8 for l in gps_) * K_
    B_cpu = int(max(8, min(spec["B"], (seconds * 0.1) * 5e10 / (f_est * K_), 8e9 / mem_per_row)))
    if B_cpu < spec["B"]:
        spec = dict(spec, B=B_cpu, _cut=True)
    ncpu = os.cpu_count() or 1
    try:
        torch.set_num_interop_threads(1)                     # (one op at a time: the op sequence is a chain)
    except RuntimeError:
        pass
    zs = synthetic.make_noise(spec, seed=1)
    m = CpuDGP(spec, dtype)
    best = None
    for nt in sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4), min(32, ncpu), min(16, ncpu), min(8, ncpu), min(4, ncpu)}, reverse=True):
        torch.set_num_threads(nt)
        m.elbo(zs)                               # warm-up
        t0 = time.perf_counter()
        m.elbo(zs)
        one = time.perf_counter() - t0
        if best is None or one < best[1]:
            best = (nt, one)
    nt, one = best
    torch.set_num_threads(nt)
    iters = int(max(3, min(200, (seconds * 0.6) / max(one, 1e-3))))
    times = []
    for _ in range(iters):
        t0 = time.perf_counter()
        m.elbo(zs)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    # FLOPs the restatement EXECUTES per sample (dense einsum 2 R M^2 per inner layer, M^2 solve, Gram; the final layer's full K x K
    # covariance: SURVEY.md section 8d, "for information") and the bytes of the [S, R, M, N] intermediate it writes and re-reads
    K = spec["K"]
    f_exec, lta_bytes = 0.0, 0.0
    for l in spec["layers"]:
        if l["type"] != "gp":
            continue
        M, D = l["Z"].shape
        R = l["q_mu"].shape[1]
        f_exec += 2 * M * D + M * M + 2 * R * M * M + 2 * M * R + (4 * R * M * K if l["W"] is None else 2 * R * M)
        lta_bytes += 3.0 * R * M * (8 if dtype == torch.float64 else 4)
    gflops = f_exec * spec["B"] * K / med / 1e9
    return dict(value=spec["B"] * spec["K"] / med, unit="samples/s", cores=nt, kind="port", executed_gflops=gflops,
                executed_flop_per_sample=f_exec, lta_gbytes_per_s=lta_bytes * spec["B"] * K / med / 1e9,
                threads_note="best of {all, 1/2, 1/4, 32, 16, 8, 4} hardware threads.  Where the time goes (torch profiler, 8 threads, float64): "
                             "36 % the element-wise square of the permuted [S, R, M, N] intermediate (a strided pass over 13 M values), 27 % the "
                             "einsum GEMM, 8 % adds, 5 % copies of permuted views -- a chain of ~150 separately parallelised ops, so beyond a few "
                             "threads per op the fork / join and the strided passes, not the FLOPs (see executed_gflops), set the time",
                sample="%d IW-ELBO evaluations of the same workload%s (B=%d, K=%d), %s torch-CPU/MKL "
                       "restatement of the reference op sequence (materialised Kmn, A, LTA, full K x K final "
                       "covariance), %d threads of %d, median" % (iters, "" if not spec.get("_cut") else " cut to its first rows", spec["B"], spec["K"],
                                                                  "float64" if dtype == torch.float64 else "float32", nt, ncpu),
                ms_per_step=med * 1e3)


def roofline_object(achieved, flops, launch_ms, spec, traffic, traffic_src, pmc):
    from dgps_with_iwvi_amd import settings
    f32p, f16p = f_alg_split(spec, settings.fw_f32_stage2)
    mix_peak = (f32p + f16p) / (f32p / PEAK_MFMA_F32 + f16p / (PEAK_MFMA_F16 / 3.0))
    hbm_frac = None if traffic is None else traffic / (launch_ms * 1e-3) / PEAK_HBM
    busy = pmc.get("mfma_busy_frac")
    # `bound` names the roofline `achieved` / `peak` are priced against (the contract's "hbm" | "mfma": this kernel's arithmetic is matrix
    # products and its HBM traffic is ~1 % of the peak, so the matrix-core ceiling of its instruction mix); `limiter` says, from the
    # counters, what the kernel actually waits for -- which today is neither
    if busy is None:
        limiter, why = "unknown", "no counters for this workload: the kernel's arithmetic is matrix products"
    elif busy >= 0.5:
        limiter, why = "mfma", "matrix pipe busy %.2f of the kernel" % busy
    elif hbm_frac is not None and hbm_frac >= 0.5:
        limiter, why = "hbm", "HBM traffic at %.2f of peak" % hbm_frac
    else:
        limiter, why = "latency/issue", ("matrix pipe busy %.2f of the kernel, HBM at %.3f of peak: dependent chains (triangular solves), barriers and "
                                         "launch / drain, not a throughput limit" % (busy, hbm_frac or 0.0))
    return {"bound": "mfma", "limiter": limiter, "bound_from": why,
            "kernel": "k_dgp_forward (all layers fused, one launch per ELBO evaluation)",
            "achieved": achieved / 1e12, "peak": mix_peak / 1e12, "unit": "TFLOP/s", "frac": achieved / mix_peak,
            "peak_basis": "instruction mix: %.0f%% of the algorithmic FLOPs on fp32 MFMAs (157.3 TF), %.0f%% as split f16 (2.5 PF / 3)" % (
                100 * f32p / (f32p + f16p), 100 * f16p / (f32p + f16p)),
            "frac_fp32_equivalent": achieved / PEAK_MFMA_F32,
            "traffic": traffic, "traffic_source": traffic_src, "hbm_frac": hbm_frac, "launch_ms": launch_ms,
            "flops_per_launch": flops, **pmc}


def l2_operand_stream(spec, T, launch_ms, variant=None):
    """Bytes of packed operands one launch of the layer kernel pulls out of L2 (every workgroup streams every layer's images once for its
    16 * NS samples: the factor's solve stream, the R images of tril(q_sqrt)^T, q_mu^T, the Gram operand -- csrc/iwvi_common.h: state
    layout), over the kernel's duration.  Not a contract field: the guide's measured L2-served rate (MI355X_MICROARCH.md, 'rows shared by
    every workgroup') is 66-73 GB/s per CU = 16.8-18.8 TB/s, which is what bounds the M = 512 config where the matrix pipe does not."""
    if variant is None:                                        # (bits of iwvi_debug_last_forward_variant: samples per workgroup / 16, split-f16 route)
        from dgps_with_iwvi_amd import _abi
        variant = int(_abi.lib().iwvi_debug_last_forward_variant())
    v = int(variant)
    ns, s16 = v & 0xff, bool(v >> 8 & 1)
    if ns < 1:
        return None
    tri = lambda n: n * (n + 1) // 2
    def sb16_slabs(nbk):
        return sum(min(nbk - 8 * I, 8) * 4 * I for I in range(1, (nbk + 7) // 8)) if nbk >= 16 else 0
    def sb16_tri(nbk):
        return sum(min(nbk - 8 * I, 8) * (min(nbk - 8 * I, 8) - 1) // 2 for I in range((nbk + 7) // 8)) if nbk >= 16 else 0
    per_wg = 0
    for l in spec["layers"]:
        if l["type"] != "gp":
            continue
        M, D = l["Z"].shape
        R = l["q_mu"].shape[1]
        nbk, nrb, nsteps = (M + 15) // 16, (R + 15) // 16, (D + 2 + 3) // 4
        per_wg += nbk * nsteps * 256                                                       # Z~ (Gram operand)
        if nbk >= 16 and s16:
            per_wg += sb16_slabs(nbk) * 2048 + sb16_tri(nbk) * 1024 + nbk * 1024          # solve: split-f16 slabs, split-f16 inverse blocks, fp32 diagonal blocks
        else:
            per_wg += tri(nbk) * 1024                                                      # solve stream, fp32 blocks
        if s16:
            slabs = sum((nbk - (bi & ~1) + 1) // 2 for bi in range(nbk))
            per_wg += R * slabs * 2048 + nrb * ((nbk + 1) // 2) * 2048                     # tril(q_sqrt[r])^T and q_mu^T as split-f16 slabs
        else:
            per_wg += R * tri(nbk) * 1024 + nrb * nbk * 1024
    nwg = (T + 16 * ns - 1) // (16 * ns)
    total = per_wg * nwg
    return {"bytes_per_workgroup": per_wg, "workgroups": nwg, "samples_per_workgroup": 16 * ns, "bytes_per_launch": total,
            "TB_per_s": total / (launch_ms * 1e-3) / 1e12, "guide_measured_l2_rate_TB_per_s": [16.8, 18.8],
            "note": "packed operands streamed from L2 by every workgroup (algorithmic: the images' sizes x workgroups), over the kernel's "
                    "measured duration; informational -- `bound` above stays the contract's matrix-core ceiling"}


def gemm_phase_mfma_util(model, spec, B, K):
    """MFMA utilisation INSIDE the batched conditional GEMM phases of the layer kernel (north_star: ">= 40 % MFMA utilisation on the
    batched conditional GEMMs", temp_workaround.py:51 and :68,78): per GP layer, the matrix instructions a workgroup issues in stage 1
    (a = Lm^-1 k) and stage 2 (u_r = L_r^T a, mean = q_mu^T a) x their issue cycles on one SIMD, over 4 SIMDs x the phase's duration in
    shader clocks (in-kernel stamps, median over the workgroups of one stamped launch).  Issue cycles per instruction
    (MI355X_MICROARCH.md / scripts/ubench/mfma_rate.hip): v_mfma_f32_16x16x4_f32 32, v_mfma_f32_16x16x16_f16 16, v_mfma_f32_16x16x32_f16 16."""
    import ctypes
    from dgps_with_iwvi_amd import _abi, settings
    lib = _abi.lib()
    lib.iwvi_debug_set_stamps.restype = None
    lib.iwvi_debug_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    dev = model.X.device
    T = B * K
    gps = [(i, l) for i, l in enumerate(spec["layers"]) if l["type"] == "gp"]
    nbks = [(l["Z"].shape[0] + 15) // 16 for _, l in gps]
    s16 = (not settings.fw_f32_stage2) and all(n % 2 == 0 for n in nbks)
    ns = min(5, max(1, (T + 16 * 256 - 1) // (16 * 256)))
    nwg = (T + 16 * ns - 1) // (16 * ns)
    if nwg + 16 > 32768:
        return None
    buf = torch.zeros(32768 * 128, dtype=torch.int64, device=dev)
    el = dict(B=B, K=K, stride_b=K, stride_k=1, mode_vi=False)
    model.precompute(with_encoders=True)
    for _ in range(2):
        model._fused_forward(T, K, B, (B, K), elbo=el)
    torch.cuda.synchronize()
    lib.iwvi_debug_set_stamps(buf.data_ptr(), 32768)
    model._fused_forward(T, K, B, (B, K), elbo=el)
    torch.cuda.synchronize()
    lib.iwvi_debug_set_stamps(None, 0)
    full = buf.view(32768, 128).cpu().numpy()
    full = full[full[:, 0] > 0]
    if len(full) == 0:
        return None
    nsamp_wg = int(round(T / len(full) / 16.0))                   # sub-tiles per workgroup the launch actually used
    NS = max(1, min(5, nsamp_wg))
    cyc = full[:, 64:].astype(np.float64)
    per_layer, tot_issue, tot_win = [], 0.0, 0.0
    tri = lambda n: n * (n + 1) // 2
    for (li, l), nbk in zip(gps, nbks):
        R = l["q_mu"].shape[1]
        nrb = (R + 15) // 16
        # stage 1 (csrc/dgp_forward.hip): nbk <= 8: per sub-tile nbk diagonal solves (4 fp32 MFMAs) + the blocks below them (2 f16 MFMAs
        # of K = 32 when nbk is even, else 4 fp32); 8 < nbk < 16: every block 4 fp32; nbk >= 16: the super-block stream (inverse blocks fp32)
        if nbk <= 8:
            off = tri(nbk) - nbk
            i1 = NS * (nbk * 4 * 32 + off * (2 * 16 if nbk % 2 == 0 else 4 * 32))
        elif nbk < 16:
            i1 = NS * tri(nbk) * 4 * 32
        else:
            dense, inv = 0, 0                                      # blocks of -L(I, <I) / of the inverse diagonal super-blocks
            for I in range((nbk + 7) // 8):
                r0, nr = 8 * I, min(8, nbk - 8 * I)
                dense += nr * r0
                inv += nr * (nr + 1) // 2
            # S16 launches: the dense part as 16 x 32 slabs of split f16 (3 MFMAs of 16 clocks per pair of blocks)
            i1 = NS * (inv * 4 * 32 + (dense // 2 * 3 * 16 if s16 else dense * 4 * 32))
        if s16:                                                    # pairs of row-blocks: (nkc - p) steps of 6 MFMAs per sub-tile; q_mu^T: 3 per slab
            nkc = nbk // 2
            i2 = NS * (R * sum(nkc - p for p in range(nkc)) * 6 + nrb * nkc * 3) * 16
        else:
            i2 = NS * (R * tri(nbk) + nrb * nbk) * 4 * 32
        w1 = float(np.median(cyc[:, 2 + li * 6 + 2] - cyc[:, 2 + li * 6 + 1]))
        w2 = float(np.median(cyc[:, 2 + li * 6 + 3] - cyc[:, 2 + li * 6 + 2]))
        per_layer.append({"layer": li, "M": int(l["Z"].shape[0]), "R": int(R),
                          "stage1": {"mfma_issue_clk": i1, "window_clk": w1, "util": i1 / (4.0 * w1) if w1 > 0 else None},
                          "stage2": {"mfma_issue_clk": i2, "window_clk": w2, "util": i2 / (4.0 * w2) if w2 > 0 else None}})
        tot_issue += i1 + i2
        tot_win += w1 + w2
    return {"value": tot_issue / (4.0 * tot_win) if tot_win > 0 else None, "per_layer": per_layer, "subtiles_per_workgroup": NS,
            "split_f16_stage2": bool(s16),
            "how": "sum over the GP layers of (MFMA instructions of stage 1 + stage 2 per workgroup x issue cycles) / (4 SIMDs x the two phases' "
                   "median duration in shader clocks, from the in-kernel stamps of one launch); the phases include their operand fetches "
                   "and the VALU work between the MFMAs, not the barriers that end them"}


def no_dev_kernarg_leg(args):
    """The same timed loop in a FRESH process with HIP_FORCE_DEV_KERNARG=0 (kernel arguments in host-visible memory: the runtime's
    default when the package has not been imported before HIP initialises), started before this process touches the GPU."""
    env = dict(os.environ, HIP_FORCE_DEV_KERNARG="0", IWVI_BENCH_CHILD="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup), "--config", str(args.config),
           "--no-cpu-baseline", "--no-train-leg", "--median-iters", "0", "--no-kernarg-leg"]
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            return {"error": "child exited %d: %s" % (p.returncode, p.stderr[-300:])}
        r = json.loads(line[0])
        return {"ms_per_step": r["ms_per_step"], "value": r["value"], "forward_launch_ms": r["roofline"]["launch_ms"],
                "how": "same command in a fresh process with HIP_FORCE_DEV_KERNARG=0"}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def visible_gpu_count(root="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs this node shows, WITHOUT any HIP / torch.cuda call in the launcher parent (it must not initialise the GPU: its children
    do): the KFD topology in sysfs -- a node with ``simd_count`` > 0 is a GPU -- narrowed by ``HIP_VISIBLE_DEVICES`` /
    ``ROCR_VISIBLE_DEVICES`` / ``CUDA_VISIBLE_DEVICES`` when set.  Falls back to ``torch.cuda.device_count()`` (which does not
    initialise HIP on this image) only where the topology is not readable."""
    import glob
    nodes = glob.glob(os.path.join(root, "*", "properties"))
    n = None
    if nodes:
        n = 0
        for f in nodes:
            try:
                with open(f) as fh:
                    props = dict(l.split()[:2] for l in fh if len(l.split()) >= 2)
                n += int(props.get("simd_count", "0")) > 0
            except (OSError, ValueError):
                n = None
                break
    if n is None:
        return torch.cuda.device_count()
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(args, argv):
    """``--gpus N`` > 1 without a launcher: start the N rank processes from HERE -- a process that has not touched the GPU (no HIP
    call so far: ``import torch`` does not initialise it and the device count comes from sysfs, ``visible_gpu_count``; nothing is re-exec'd) -- with the
    environment contract of ``torch.distributed.run`` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), dmabuf IPC for RCCL
    and device-resident kernel arguments.  Rank 0's stdout (the one JSON line) is relayed; the other ranks' stdout goes to stderr.
    Returns the exit code: the first non-zero rank's (the others are then killed -- exactly the process groups started here), 124 on
    ``--launch-timeout``."""
    import signal
    import socket
    import threading
    n = args.gpus
    ndev = visible_gpu_count()
    if args.rendezvous_only:
        ndev = n
    if n > ndev and not args.oversubscribe:
        print("bench.py: --gpus %d but this node shows %d device(s); one rank per GPU.  (--oversubscribe --backend gloo puts several "
              "ranks on one device: a plumbing run, never a measurement)" % (n, ndev), file=sys.stderr)
        return 2
    if n > ndev and args.backend == "nccl":
        print("bench.py: RCCL cannot place two ranks on one device; --oversubscribe needs --backend gloo", file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                HSA_ENABLE_IPC_MODE_LEGACY="0", IWVI_BENCH_LAUNCHED="1", IWVI_BENCH_BACKEND=args.backend)
    base.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    procs, out0 = [], []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env, cwd=ROOT, text=True,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, start_new_session=True))
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()

    def kill_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)             # (start_new_session: the rank's pid is its process group)
                except ProcessLookupError:
                    pass
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass

    def on_signal(signum, frame):
        kill_all()
        sys.exit(128 + signum)
    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, on_signal)
    deadline = time.monotonic() + args.launch_timeout
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc = bad[0][1] if bad[0][1] > 0 else 1
            print("bench.py: rank %d exited with %d; stopping the other ranks" % bad[0], file=sys.stderr)
            kill_all()
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            print("bench.py: ranks still running after --launch-timeout %.0f s; stopping them" % args.launch_timeout, file=sys.stderr)
            kill_all()
            rc = 124
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    lines = [l for l in out0 if l.startswith("{")]
    for l in out0:
        (sys.stdout if l.startswith("{") else sys.stderr).write(l)
    sys.stdout.flush()
    if rc == 0 and len(lines) != 1:
        print("bench.py: rank 0 printed %d JSON lines (expected 1)" % len(lines), file=sys.stderr)
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--shard", choices=["k", "n"], default="k")
    ap.add_argument("--split-k", action="store_true",
                    help="strong scaling of the K-shard: the CONFIG's K importance samples are divided over the ranks "
                         "(sharding.split_samples: 50 over 8 -> 7,7,6,6,6,6,6,6 -- BASELINE.json configs[3]/[4] 'K-sharded across 8') "
                         "instead of every rank drawing K of its own (weak scaling, the default)")
    ap.add_argument("--batch", type=int, default=None, help="override the config's minibatch size (plumbing tests; not a BASELINE workload)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-train-leg", action="store_true", help="skip the (informational) training-step timing")
    ap.add_argument("--sharded-train-leg", action="store_true",
                    help="with more than one rank: also time the sharded training step (eager and as hipGraph segments; informational, after the "
                         "timed region).  Opt-in: its collectives have only ever met gloo ranks on one GPU, and a scaling run must not depend on them")
    ap.add_argument("--check", action="store_true",
                    help="after the timed region: one evaluation on INJECTED job-wide noise through the very same sharded path "
                         "(graph-free), and on rank 0 the unsharded evaluation of the whole job on the same noise; both go into the JSON")
    ap.add_argument("--median-iters", type=int, default=60, help="hipEvent-timed single-evaluation replays for ms_per_step_median")
    ap.add_argument("--no-kernarg-leg", action="store_true", help="skip the (informational) HIP_FORCE_DEV_KERNARG=0 child run")
    ap.add_argument("--xch-every", type=int, default=0,
                    help="multi-GPU: evaluations per exchange (default: one exchange per graph replay of up to 25 evaluations; 1 = one "
                         "exchange per evaluation, what a training loop pays)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=os.environ.get("IWVI_BENCH_BACKEND", "nccl"),
                    help='process-group backend: "nccl" IS RCCL on ROCm (the measured path); "gloo" only for plumbing runs')
    ap.add_argument("--oversubscribe", action="store_true",
                    help="plumbing only: allow more ranks than devices (rank r on device r %% device_count; needs --backend gloo)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="plumbing only: every rank joins the process group (gloo, CPU tensors), all-reduces its rank and rank 0 prints "
                         "{n_ranks_seen, rank_sum, env}; no GPU is touched -- checks the launcher and the rendezvous by themselves")
    ap.add_argument("--launch-timeout", type=float, default=1800.0, help="self-launched ranks are stopped after this many seconds")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))               # (this process never touches the GPU)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rendezvous_only:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([rank], dtype=torch.int64)
        dist.all_reduce(t)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "n_gpus": world, "n_ranks_seen": dist.get_world_size(), "rank_sum": int(t.item()),
                              "self_launched": os.environ.get("IWVI_BENCH_LAUNCHED") == "1",
                              "environment": {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "HIP_FORCE_DEV_KERNARG",
                                                                              "MASTER_ADDR", "LOCAL_WORLD_SIZE")}}))
        dist.destroy_process_group()
        return
    kernarg_leg = None
    if world == 1 and not args.no_kernarg_leg and not args.no_train_leg and os.environ.get("IWVI_BENCH_CHILD") != "1" \
            and os.environ.get("HIP_FORCE_DEV_KERNARG") == "1":
        kernarg_leg = no_dev_kernarg_leg(args)                   # (before this process's first GPU call)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the HIP path has no CPU fallback)")
    if world > torch.cuda.device_count() and os.environ.get("IWVI_BENCH_BACKEND", args.backend) == "nccl":
        raise SystemExit("bench.py: %d ranks on %d device(s): RCCL needs one device per rank" % (world, torch.cuda.device_count()))
    local %= torch.cuda.device_count()                           # (several ranks on one device only in plumbing runs: gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    force_xch = world == 1 and os.environ.get("IWVI_BENCH_FORCE_XCH") == "1"   # plumbing test: the exchange path on one rank
    if world > 1 or force_xch:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("IWVI_BENCH_BACKEND", args.backend)   # "nccl" is RCCL on ROCm; "gloo" only for plumbing tests
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world:
        # a launcher started a different number of ranks than --gpus names: the line would carry an n_gpus the caller did not ask for
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (the launcher's rank count and --gpus must agree)" % (args.gpus, world))

    from dgps_with_iwvi_amd import _abi, settings, synthetic
    _abi.lib()                                                   # fail loudly if the extension is missing
    if world > 1:
        settings.set_seed(settings.seed + 7919 * rank)           # every rank its own Philox key: the job's K_total samples are distinct
    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["B"] = int(args.batch)
    if args.split_k and args.shard != "k":
        raise SystemExit("--split-k divides the importance samples: it needs --shard k")
    from dgps_with_iwvi_amd.sharding import split_samples
    # importance samples of THIS rank / of the job: weak K-shard: K each, K * world in all; --split-k: the config's K divided
    K_parts = split_samples(cfg["K"], world) if args.split_k else [cfg["K"]] * world
    K_local = K_parts[rank]
    K_job = cfg["K"] if (args.split_k or args.shard != "k") else cfg["K"] * world
    # parity=True: random q_mu / dense lower-triangular q_sqrt (a trained-like state).  The reference's
    # initial values (q_mu = 0, q_sqrt = 1e-5 I) would feed the MFMAs mostly zeros and flatter the clock.
    spec = synthetic.make_spec(seed=0, parity=True, n_data=65536, **cfg)
    if args.shard == "n" and world > 1:                          # each rank owns different points
        lo = (rank * cfg["B"]) % (spec["n_data"] - cfg["B"] + 1)
        spec = dict(spec, X=spec["X"][lo:], Y=spec["Y"][lo:])
    model = synthetic.build_model(spec, dev, num_samples=K_local)
    model.lv_in_precompute = os.environ.get("IWVI_BENCH_LV_PRE", "0") == "1"   # leading LV layer inside the precompute launch
    step = Step(model, spec, dev, args.shard, world, exchange=(world > 1 or force_xch), K_total=K_job)
    B, K = cfg["B"], K_local
    # ---- capture -----------------------------------------------------------------------------
    # steps per graph replay: every step is the complete evaluation (fresh noise from the device counter); several
    # per replay only spares the host-side launch between them.  Multi-GPU runs exchange the replay's evaluations in
    # one collective (they are independent of each other; see sharding.OverlappedExchange).
    import math
    spg = 1
    if not args.no_graph:
        cap = int(os.environ.get("IWVI_BENCH_SPG", "25"))
        if args.steps >= 12:
            cap = min(cap, args.steps // 3)                      # never ONE replay as the whole timed region: >= 3 replays (at the driver's
                                                                 # --steps 20: 4 replays of 5 evaluations; a replay's launch is ~7 us)
        if args.xch_every > 0 and (world > 1 or force_xch):
            cap = args.xch_every
        spg = max(d for d in range(1, cap + 1) if args.steps % d == 0)      # the timed region is exactly --steps evaluations
        if spg < min(4, args.steps) and world == 1 and not force_xch:
            spg = min(cap, args.steps)                                        # awkward --steps: full replays + a remainder launched singly
    xch = None
    if world > 1 or force_xch:
        # multi-GPU: the evaluations of one graph replay are exchanged in one collective on a side stream
        from dgps_with_iwvi_amd.sharding import OverlappedExchange
        xch = OverlappedExchange(args.shard, world, B, K_job, float(spec["n_data"]) / B, dev, steps=spg, timed=True)
    graph = None
    step.run()
    torch.cuda.synchronize()

    if not args.no_graph:
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step.run()
            if xch is None:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=s, capture_error_mode="thread_local"):
                    for _ in range(spg):
                        step.run()
            else:
                # one graph per staging slot: spg complete evaluations, each writing its result into the slot
                graph = []
                for slot in range(xch.depth):
                    gph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gph, stream=s, capture_error_mode="thread_local"):
                        for view in xch.slot_views(slot):
                            step.run(out=view)
                    graph.append(gph)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()

    def one_step():
        """spg evaluations (one graph replay), plus their exchange when sharded."""
        if xch is None:
            if graph is not None:
                graph.replay()
            else:
                step.run()
            return
        slot = xch.before_step()
        if graph is not None:
            graph[slot].replay()
        else:
            for view in xch.slot_views(slot):
                step.run(out=view)
        xch.submit(global_kls=getattr(step, "glob", None))

    def fence():
        if xch is not None:
            xch.finish()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up: whole replays (at least one when anything is exchanged or replayed from a graph: the first collective creates
    # the RCCL communicator and the first replay uploads the graph -- neither may land in the timed region), then the rest
    # of --warmup one evaluation at a time
    n_warm = args.warmup // spg
    if xch is not None or graph is not None:
        n_warm = max(n_warm, 3)                                  # a graph's first replays are slower than its steady state (68.0, 66.4, 66.0,
                                                                 # then 65.6 us per evaluation: scripts/time_fixed_overhead.py)
    for _ in range(n_warm):
        one_step()
    for _ in range(max(0, args.warmup - n_warm * spg)):
        step.run()
    # ... and the device at its steady clocks: after an idle spell the GPU needs ~20 ms of work before an evaluation takes what it takes in
    # a long run (measured with --steps 20: 65.7 us per evaluation after 60 warm-up evaluations, 64.4 after 120, 63.7 after 240, 63.5 after
    # 500 and 1000; a 200-step run: 62.8).  A timed region of 20 evaluations is 1.3 ms long, so the warm-up goes on until the device has been
    # busy for 25 ms -- never fewer evaluations than --warmup asks for; the count is reported as `warmup_evaluations_run`
    warm_s = float(os.environ.get("IWVI_BENCH_WARM_SECONDS", "0.025"))
    if warm_s > 0:
        fence()
        t_w = time.perf_counter()
        one_step()
        fence()
        n_more = min(5000, max(0, int(math.ceil(warm_s / max(time.perf_counter() - t_w, 1e-6))) - 1))
        if dist is not None:                                     # (every rank submits the same number of exchanges)
            t_n = torch.tensor([n_more], dtype=torch.int64, device=dev)
            dist.all_reduce(t_n, op=dist.ReduceOp.MAX)
            n_more = int(t_n.item())
        for _ in range(n_more):
            one_step()
        n_warm += 1 + n_more
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps // spg):
        one_step()
    for _ in range(args.steps % spg):                            # (only when --steps has no useful divisor; single GPU)
        step.run()
    t_enqueued = time.perf_counter() - t0                        # host side done; the device may still be running
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    final_elbo = float((xch.finish()[-1] if xch is not None else step.out.reshape(1)).item())
    exchange_ms = xch.exchange_ms() if xch is not None else None

    # ---- --check: the sharded path against the unsharded job on injected noise (multi-rank correctness, no timing) ----------
    check = None
    if args.check:
        Kt = K_job
        Bt = B * (world if args.shard == "n" else 1)
        job_spec = synthetic.make_spec(seed=0, parity=True, n_data=65536, **dict(cfg, K=Kt, B=Bt))
        zjob = synthetic.make_noise(job_spec, seed=123, K=Kt, B=Bt)
        if args.shard == "k":
            k0 = sum(K_parts[:rank])
            zloc = [z[:, k0:k0 + K_local] for z in zjob]
        else:
            zloc = [z[rank * B:(rank + 1) * B] for z in zjob]
        zloc = [torch.as_tensor(np.ascontiguousarray(z), dtype=torch.float32, device=dev) for z in zloc]
        if xch is not None:
            slot = xch.before_step()
            views = xch.slot_views(slot)
            for view in views:                                   # every evaluation of the slot on the same noise
                step.run(out=view, zs=zloc)
            xch.submit(global_kls=getattr(step, "glob", None))
            got = [float(v) for v in xch.finish().tolist()]
        else:
            got = [float(step.run(zs=zloc).reshape(1).item())]
        fence()
        check = {"sharded_elbo": got[0], "all_steps_equal": all(g == got[0] for g in got), "K_total": Kt, "B_total": Bt}
        if rank == 0:
            full = synthetic.build_model(job_spec, dev)
            check["unsharded_elbo"] = float(full._build_likelihood(
                [torch.as_tensor(z, dtype=torch.float32, device=dev) for z in zjob]).item())
            check["rel_diff"] = abs(check["sharded_elbo"] - check["unsharded_elbo"]) / abs(check["unsharded_elbo"])

    # ---- SURVEY.md section 8(d) protocol: median of >= 50 hipEvent-timed iterations, one complete evaluation per iteration
    #      (own graph of 1 evaluation: precompute + fused forward; fresh device-drawn noise every replay) ------------------
    med = None
    if world == 1 and args.median_iters > 0:
        s3 = torch.cuda.Stream(device=dev)
        s3.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s3):
            model._build_likelihood()
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, stream=s3, capture_error_mode="thread_local"):
                keep1 = model._build_likelihood()
        torch.cuda.current_stream().wait_stream(s3)
        torch.cuda.synchronize()
        for _ in range(10):
            g1.replay()
        torch.cuda.synchronize()
        evs1 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.median_iters)]
        for a, b in evs1:
            a.record()
            g1.replay()
            b.record()
        torch.cuda.synchronize()
        ts = np.array([a.elapsed_time(b) for a, b in evs1])
        med = {"ms_per_step_median": float(np.median(ts)), "p10": float(np.percentile(ts, 10)), "p90": float(np.percentile(ts, 90)),
               "iters": int(args.median_iters), "warmup": 10,
               "how": "hipEvents around single-evaluation hipGraph replays (one precompute + one fused forward each), back to back"}
        if graph is not None and xch is None:
            # ... and of the timed region's own unit: hipEvents around whole replays of the timed graph (spg evaluations each), per evaluation
            evs_r = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
            for a, b in evs_r:
                a.record()
                graph.replay()
                b.record()
            torch.cuda.synchronize()
            tr = np.array([a.elapsed_time(b) for a, b in evs_r]) / spg
            med.update({"ms_per_step_median_of_replays": float(np.median(tr)), "replays_p10": float(np.percentile(tr, 10)),
                        "replays_p90": float(np.percentile(tr, 90)), "evaluations_per_replay": int(spg)})

    # ---- dominant kernel: the fused forward (all layers), HIP events around a graph of back-to-back launches
    tot_flops, _ = f_alg_model(spec)
    dom_flops = tot_flops * B * K
    model.precompute(with_encoders=True)
    NREP = 20
    fwd = lambda: model._fused_forward(B * K, K, B, (B, K), elbo=dict(B=B, K=K, stride_b=K, stride_k=1, mode_vi=False))
    fwd()
    torch.cuda.synchronize()
    s2 = torch.cuda.Stream(device=dev)
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        fwd()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=s2, capture_error_mode="thread_local"):
            for _ in range(NREP):
                keep_logw = fwd()
    torch.cuda.current_stream().wait_stream(s2)
    torch.cuda.synchronize()
    for _ in range(3):
        g2.replay()
    torch.cuda.synchronize()
    reps = max(5, args.steps // NREP)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        g2.replay()
        b.record()
    torch.cuda.synchronize()
    dom_ms = float(np.median([a.elapsed_time(b) for a, b in evs])) / NREP
    achieved = dom_flops / (dom_ms * 1e-3)
    try:                                                     # (read now: the variant the timed launches took)
        l2_stream = l2_operand_stream(spec, B * K, dom_ms)
    except Exception as e:                                   # informational
        l2_stream = {"error": "%s: %s" % (type(e).__name__, e)}
    gemm_util = None
    if world == 1:
        try:
            gemm_util = gemm_phase_mfma_util(model, spec, B, K)
        except Exception as e:                                   # (diagnostic stamps: must not take the headline line down)
            gemm_util = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- the same evaluation with the fp32-MFMA stage 2 (iwvi_layer_desc.flags & IWVI_LAYER_F32_STAGE2: a per-call flag), so that the
    #      pure-fp32 figure is observed in the same run, next to the split-f16 default
    fp32_path = None
    if world == 1 and not settings.fw_f32_stage2:
        settings.fw_f32_stage2 = True
        try:
            s4 = torch.cuda.Stream(device=dev)
            s4.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s4):
                model._build_likelihood()
                g4 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g4, stream=s4, capture_error_mode="thread_local"):
                    for _ in range(10):
                        keep4 = model._build_likelihood()
            torch.cuda.current_stream().wait_stream(s4)
            torch.cuda.synchronize()
            g4.replay()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            for _ in range(max(2, args.steps // 10)):
                g4.replay()
            torch.cuda.synchronize()
            ms4 = (time.perf_counter() - t4) / (max(2, args.steps // 10) * 10) * 1e3
            fp32_path = {"ms_per_step": ms4, "samples_per_s": B * K / ms4 * 1e3,
                         "model_frac_of_fp32_mfma_peak": B * K / ms4 * 1e3 * tot_flops / PEAK_MFMA_F32,
                         "how": "IWVI_LAYER_F32_STAGE2 on every GP layer (v_mfma_f32_16x16x4_f32 in stage 2), hipGraph of 10 evaluations"}
        finally:
            settings.fw_f32_stage2 = False

    # informational, never `value`: two INDEPENDENT evaluations in flight -- two models with the same parameters and their own state, noise
    # streams and results, each a captured chain [factorisation launch -> layer launch] x 10 on its own stream.  One model's 25 us
    # factorisation (ten workgroups) then runs beside the other's layer launch instead of in front of its own: what a serving loop over
    # independent minibatches gets, and the head-room an overlapped single evaluation is after.  A training
    # step cannot do this (its next parameters depend on this step's gradient), which is why the headline stays one evaluation at a time.
    two_in_flight = None
    if world == 1 and os.environ.get("IWVI_BENCH_TWO_IN_FLIGHT", "1") != "0" and not args.no_train_leg:
        try:
            models2 = [model, synthetic.build_model(spec, dev, num_samples=K_local)]
            streams2 = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
            graphs2, keep2 = [], []
            for m2, s2 in zip(models2, streams2):
                s2.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s2):
                    m2._build_likelihood()
                    g2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g2, stream=s2, capture_error_mode="thread_local"):
                        for _ in range(10):
                            keep2.append(m2._build_likelihood())
                    graphs2.append(g2)
            for s2 in streams2:
                torch.cuda.current_stream().wait_stream(s2)
            torch.cuda.synchronize()

            def both():
                for g2, s2 in zip(graphs2, streams2):
                    with torch.cuda.stream(s2):
                        g2.replay()
            for _ in range(3):
                both()
            torch.cuda.synchronize()
            reps2 = max(4, args.steps // 10)
            t2 = time.perf_counter()
            for _ in range(reps2):
                both()
            torch.cuda.synchronize()
            ms2 = (time.perf_counter() - t2) / (reps2 * 20) * 1e3
            two_in_flight = {"ms_per_evaluation": ms2, "samples_per_s": B * K / ms2 * 1e3,
                             "finite": bool(all(torch.isfinite(k).all().item() for k in keep2[-2:])),
                             "how": "two models (same parameters, own state / noise stream / result), each a hipGraph of 10 [k_precompute -> "
                                    "k_dgp_forward] evaluations replayed on its own stream, both streams busy; informational -- independent "
                                    "evaluations overlap, one evaluation's latency is `ms_per_step`"}
        except Exception as e:                                   # (an informational leg must not take the headline line down)
            import traceback
            traceback.print_exc()
            two_in_flight = {"error": "%s: %s" % (type(e).__name__, e)}

    # HBM bytes per launch of the dominant kernel from the committed PMC profile of this very workload (the counters
    # need their own rocprofv3 passes and cannot be read live); null for any other workload
    traffic, traffic_src, pmc = None, None, {}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
        if args.config == 2 and world == 1:
            traffic, traffic_src = tj["hbm_bytes"], tj["source"]
            pmc = {k: tj[k] for k in ("mfma_busy_frac", "mfma_issued_f32_tflops", "mfma_issued_f16_tflops", "mfma_pipe_frac",
                                      "kernel_avg_us_rocprof", "pmc_profile_of_commit") if k in tj}
            # are the quoted counters OF the kernels this run timed?  (source hash recorded with the profile vs the tree's)
            from dgps_with_iwvi_amd.kernel_resources import csrc_hash
            pmc["pmc_profile_is_of_these_sources"] = (tj.get("csrc_sha256") == csrc_hash()) if tj.get("csrc_sha256") else None
    except Exception:
        pass
    sharded_train = None
    if world > 1 and args.sharded_train_leg and not args.rendezvous_only:
        sharded_train = sharded_training_leg(model, args.shard, world, K_job, dist)      # (every rank takes part in its collectives)
    if rank == 0:
        total = float(B) * (K_job if args.shard == "k" else K * world) * args.steps      # the job's samples per step x steps
        res = {
            "metric": "IW-ELBO samples/sec (KxN) at L=2, M=128, K=20",
            "value": total / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "warmup_evaluations_run": max(args.warmup, n_warm * spg),   # (whole graph replays: never fewer than asked)
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.split_k else "weak", "vs_baseline": None, "dtype": "f32" if settings.fw_f32_stage2 else "f32 (split-f16 operands in stage 2)", "data": "synthetic",
            # what "f32" means on this path: float32 data and accumulation, float64 factorisation (K_uu, Cholesky); the operands of
            # stage 2 (u_r = L_r^T a, mean = q_mu^T a) enter the matrix cores as x = h1 + h2, two f16 planes = 22 mantissa bits, with
            # power-of-two scales -- as accurate as the fp32 MFMA it replaces (tests/test_gpu_split16.py; IWVI_FW_F32_STAGE2=1 switches back)
            "dtype_note": ("fp32 data and accumulate, fp64 factorisation; stage-2 matrix operands as split f16 (h1 + h2, 22 mantissa bits) "
                           "unless IWVI_LAYER_F32_STAGE2 is set on a layer (see fp32_path)" if not settings.fw_f32_stage2 else "fp32 MFMA stage 2 (IWVI_LAYER_F32_STAGE2)"),
            "config": {"workload": "BASELINE.json configs[%d]: %s; Dx=8, Dy=1, inner layers G5 (R=5, P=8), RBF-ARD, "
                                   "per-step Gram+Cholesky included, noise drawn on device%s" % (
                                       args.config, spec["name"], "" if not args.batch else " -- batch overridden (--batch %d): a plumbing run, not the BASELINE workload" % args.batch),
                       "global_batch": B * (world if args.shard == "n" else 1),
                       "K_total": K_job, "K_per_rank": K_parts,
                       "sharding": ("none" if world == 1 else args.shard + "-shard" + (" (the config's K split over the ranks)" if args.split_k else "")),
                       "launch": "eager" if graph is None else ("hipGraph replay, %d steps per replay" % spg) + ("" if xch is None else ", one exchange per replay")},
            "elbo": final_elbo,
            "n_ranks_seen": (dist.get_world_size() if dist is not None else 1),
            # side-stream time of one exchange (collective + merge kernel, from events; overlapped with the next replay) and how many
            # evaluations it carries
            "exchange_ms": exchange_ms, "evaluations_per_exchange": (spg if xch is not None else None),
            "host_enqueue_ms_per_step": t_enqueued / args.steps * 1e3,
            # achieved = algorithmic FLOP / s of the dominant kernel.  peak = the ceiling of THIS instruction mix: the fp32-MFMA part of the
            # algorithm at the fp32-MFMA peak, the split-f16 part (three f16 MFMA FLOPs per algorithmic FLOP) at a third of the f16 peak --
            # so frac <= 1 whatever the shape.  frac_fp32_equivalent (algorithmic / fp32-MFMA peak) is last round's number; it passes 1
            # where stage 2 dominates (M >= 256) and is kept only for comparison.  bound: from the counters of the committed profile.
            "roofline": roofline_object(achieved, dom_flops, dom_ms, spec, traffic, traffic_src, pmc),
        }
        if gemm_util is not None:
            res["roofline"]["gemm_phase_mfma_util"] = gemm_util
        res["roofline"]["l2_operand_stream"] = l2_stream
        f32p, f16p = f_alg_split(spec, settings.fw_f32_stage2)
        mix_peak = (f32p + f16p) / (f32p / PEAK_MFMA_F32 + f16p / (PEAK_MFMA_F16 / 3.0))
        res["model_frac_of_mfma_peak"] = res["value"] / world * tot_flops / mix_peak          # the WHOLE step (both launches) against the mix ceiling
        res["model_frac_of_fp32_mfma_peak"] = res["value"] / world * tot_flops / PEAK_MFMA_F32
        if fp32_path is not None:
            res["fp32_path"] = fp32_path
        if two_in_flight is not None:
            res["two_in_flight"] = two_in_flight
        if kernarg_leg is not None:
            res["no_dev_kernarg"] = kernarg_leg
        res["self_launched"] = os.environ.get("IWVI_BENCH_LAUNCHED") == "1"     # bench.py --gpus N started its own ranks (no torchrun)
        res["environment"] = {"HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG"),
                              "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                              "set_by": "dgps_with_iwvi_amd/__init__.py (setdefault at import, before HIP initialises)"}
        if med is not None:
            res.update({"ms_per_step_median": med["ms_per_step_median"], "median_protocol": med})
        if check is not None:
            res["check"] = check
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(spec, args.cpu_seconds)
            try:                                                 # float32 on the CPU: the precision the device path computes in
                res["cpu_baseline_fp32"] = cpu_baseline(spec, args.cpu_seconds * 0.5, torch.float32)
            except Exception as e:                               # (a float32 Cholesky of K_uu + 1e-6 I may fail on the host)
                res["cpu_baseline_fp32"] = {"error": "%s: %s" % (type(e).__name__, e)}
            res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
        if world == 1 and not args.no_train_leg:
            res["training_step"] = training_leg(model, B * K, flops_forward=tot_flops * B * K)
        if sharded_train is not None:
            res["training_step"] = sharded_train
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

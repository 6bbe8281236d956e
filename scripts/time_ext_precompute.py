#!/usr/bin/env python3
"""Experiment (development aid): ONE evaluation as two CONCURRENT launches -- the 1024-thread factorisation launch
(iwvi_gp_precompute_pub, publishing through the merged launch's counters) on a second, high-priority stream, and the layer launch in
its merged form (IWVI_FZ_EXT: every workgroup a chunk workgroup that waits for those counters) on the main stream -- against the
default two launches one behind the other.  Checks that both give the same bits, then times captured graphs of 20 evaluations."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS
from dgps_with_iwvi_amd import _abi, settings, synthetic
from dgps_with_iwvi_amd.layers import GPLayer

dev = torch.device("cuda:0")
cfg = CONFIGS[2]
B, K = cfg["B"], cfg["K"]
spec = synthetic.make_spec(seed=0, parity=True, n_data=65536, **cfg)
lib = _abi.lib()
side = torch.cuda.Stream(device=dev, priority=-1)


def build():
    settings.set_seed(0)
    return synthetic.build_model(spec, dev)


def eval_default(m):
    return m._build_likelihood()


def eval_ext(m):
    main = torch.cuda.current_stream()
    gps = [l.state_desc() for l in m.layers if isinstance(l, GPLayer)]
    arr = (_abi.GpDesc * len(gps))(*gps)
    side.wait_stream(main)                                        # the factorisation overwrites what the previous layer launch reads
    with torch.cuda.stream(side):
        _abi.check(lib.iwvi_gp_precompute_pub(arr, len(gps), _abi.ptr(m._fz_ws()), _abi.stream_ptr()))
    # ... and the layer launch sits behind a one-workgroup gate that leaves when the factorisation's workgroups have their CUs
    _abi.check(lib.iwvi_fz_gate(arr, len(gps), _abi.ptr(m._fz_ws()), _abi.stream_ptr()))
    settings.merged_launch = True
    try:
        out = m._build_likelihood()
    finally:
        settings.merged_launch = False
    main.wait_stream(side)
    return out


# ---- same bits?
ma, mb = build(), build()
va = [float(eval_default(ma)) for _ in range(3)]
print('default evaluations done', va, flush=True)
EXT = int(os.environ.get("EXT", "1"))
_abi.set_debug_option("IWVI_FZ_EXT", EXT)
vb = []
for _ in range(3):
    vb.append(float(eval_ext(mb))); print("ext evaluation:", vb[-1], flush=True)
_abi.set_debug_option("IWVI_FZ_EXT", 0)
torch.cuda.synchronize()
print("default:", va, flush=True)
print("ext    :", vb, "equal bits:", va == vb)
words = mb._fz_ws()[:24].view(torch.int32).cpu().numpy()
print("sync words gen/role/early/pack/done/timeout:", words[:6].tolist())

def eval_ext_serial(m):
    """the same two launches, one behind the other on ONE stream (what the counters and the merged-form layer launch cost without any overlap)"""
    gps = [l.state_desc() for l in m.layers if isinstance(l, GPLayer)]
    arr = (_abi.GpDesc * len(gps))(*gps)
    _abi.check(lib.iwvi_gp_precompute_pub(arr, len(gps), _abi.ptr(m._fz_ws()), _abi.stream_ptr()))
    settings.merged_launch = True
    try:
        return m._build_likelihood()
    finally:
        settings.merged_launch = False


# ---- timing, eager: two real streams
for name, m, fn, ext in (("two launches, one stream (eager)", ma, eval_default, 0), ("factorisation launch beside the layer launch (eager)", mb, eval_ext, EXT),
                         ("the same two launches one behind the other (eager)", mb, eval_ext_serial, EXT)):
    _abi.set_debug_option("IWVI_FZ_EXT", ext)
    for _ in range(50): fn(m)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(400): fn(m)
    torch.cuda.synchronize()
    print("%-56s %.2f us per evaluation (400 eager evaluations)" % (name, (time.perf_counter() - t0) * 1e6 / 400), flush=True)
_abi.set_debug_option("IWVI_FZ_EXT", 0)

# ---- timing: captured graphs of 20 evaluations
graphs = {}
for name, m, fn, ext in (("two launches, one stream", ma, eval_default, 0), ("factorisation launch beside the layer launch", mb, eval_ext, EXT)):
    _abi.set_debug_option("IWVI_FZ_EXT", ext)
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(m)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            for _ in range(20):
                keep = fn(m)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graphs[name] = (g, keep)
_abi.set_debug_option("IWVI_FZ_EXT", 0)
for name, (g, keep) in graphs.items():
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
    print("first replay of '%s': %.1f us per evaluation, ELBO %.6f" % (name, (time.perf_counter() - t0) * 1e6 / 20, float(keep)), flush=True)
for _ in range(20):
    for g, _k in graphs.values(): g.replay()
torch.cuda.synchronize()
for rnd in range(3):
    for name, (g, keep) in graphs.items():
        ts = []
        for _ in range(10):
            torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6 / 20)
        print("%-46s %.2f us per evaluation (median of 10 replays of 20), last ELBO %.6f" % (name, np.median(ts), float(keep)), flush=True)
words = mb._fz_ws()[:96].view(torch.int32).cpu().numpy()
print("sync words gen/role/early/pack/done/timeout:", words[:6].tolist(), "started:", int(words[22]))

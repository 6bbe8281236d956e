"""The C-ABI consumed from plain C (examples/c_abi_smoke.c, compiled with gcc as C99 against include/iwvi_hip.h and the
in-tree libiwvi_hip.so -- no Python, no torch in that process) against the fp64 oracle on the same formula-defined inputs."""
import os
import subprocess

import numpy as np
import pytest

from oracle import iwvi_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "c_abi_smoke.c")
EXE = os.path.join(ROOT, "examples", "c_abi_smoke")


def _build():
    lib = os.path.join(ROOT, "dgps_with_iwvi_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c99", "-O2", SRC, "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                           "-D__HIP_PLATFORM_AMD__", "-L" + lib, "-liwvi_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", EXE])


def test_header_is_valid_c99_and_links():
    """(no GPU needed) include/iwvi_hip.h compiles as C and every call the example makes resolves in libiwvi_hip.so."""
    if not os.path.exists(os.path.join(ROOT, "dgps_with_iwvi_amd", "csrc", "libiwvi_hip.so")):
        pytest.skip("libiwvi_hip.so not built (run __graft_entry__.build())")
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_plain_c_caller_matches_oracle(gpu_device):
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    vals = np.array([float(x) for x in out.stdout.split()])
    M, D, R, P, T = 48, 4, 3, 4, 100
    f32 = np.float32
    Z = np.sin(0.37 * np.arange(M * D) + 0.1).astype(f32).reshape(M, D)
    ls = (1.0 + 0.25 * np.arange(D)).astype(f32)
    q_mu = np.cos(0.11 * np.arange(M * R)).astype(f32).reshape(M, R)
    r, i, j = np.meshgrid(np.arange(R), np.arange(M), np.arange(M), indexing="ij")
    q_sqrt = np.where(j > i, 0.0, np.where(i == j, (f32(0.5) + f32(0.01) * r.astype(f32)).astype(np.float64),
                                           (f32(0.02) * np.sin(0.3 * (i + 2 * j + r)).astype(f32)).astype(np.float64))).astype(f32)
    F = (np.sin(0.05 * np.arange(T * D)).astype(f32) * f32(1.5)).reshape(1, T, D)
    z = np.cos(0.7 * np.arange(T * R)).astype(f32).reshape(1, T, R)
    W = (f32(0.3) * np.sin(1.0 + np.arange(P * R)).astype(f32)).reshape(P, R)
    A = np.eye(D, P).astype(f32)
    layer = O.GPLayer(O.SharedMixedMok(O.RBF(D, variance=float(f32(1.3)), lengthscales=ls), W), Z, R, O.Linear(A, np.zeros(P)))
    layer.q_mu, layer.q_sqrt = q_mu, q_sqrt
    s, m, v, _ = layer.propagate(F, z=z)
    m, v, s = m[0], v[0], s[0]
    np.testing.assert_allclose(vals[:4 * P], m[:4].reshape(-1), rtol=2e-3, atol=1e-3)
    np.testing.assert_allclose(vals[4 * P:8 * P], v[:4].reshape(-1), rtol=2e-3, atol=1e-4)
    cs, cm, cv = vals[8 * P:8 * P + 3]
    assert abs(cm - m.sum()) <= 2e-3 * np.abs(m).sum() and abs(cv - v.sum()) <= 2e-3 * np.abs(v).sum()
    assert abs(cs - s.sum()) <= 2e-3 * np.abs(s).sum()

"""Golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py from the fp64 oracle).

CPU side: the oracle must keep reproducing them (drift guard).  The GPU side of the same fixtures is
tests/test_gpu_parity.py::test_golden_fixture."""
import glob
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import CASES, arrays_to_spec, oracle_outputs   # noqa: E402

FIXTURES = sorted(p for p in glob.glob(os.path.join(HERE, "golden", "*.npz")) if not os.path.basename(p).startswith("grad_"))


def load(path):
    with np.load(path, allow_pickle=False) as f:
        a = {k: f[k] for k in f.files}
    spec = arrays_to_spec(a)
    zs = [a["z%d" % i] for i in range(len(spec["layers"]))]
    out = {k[4:]: v for k, v in a.items() if k.startswith("out_")}
    return spec, zs, out


def test_every_case_has_a_fixture():
    assert sorted(os.path.basename(p)[:-4] for p in FIXTURES) == sorted(CASES)


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_reproduces_golden(path):
    spec, zs, out = load(path)
    now = oracle_outputs(spec, zs)
    assert sorted(now) == sorted(out)
    for k, v in out.items():
        np.testing.assert_allclose(now[k], v, rtol=1e-10, atol=1e-12, err_msg=k)

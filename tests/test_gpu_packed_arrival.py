"""The end of the fused forward launch (csrc/dgp_forward.hip: fw_arrive_fast): every workgroup's partial sum of the per-point log p
(models.py:148) travels inside its ticket -- one 64-bit atomic add, 46 bits of fixed point in units of 2^-20 -- and the last arriver
finishes models.py:150 from the value the add returns.  Checked against the route it replaced (partials stored, drained, re-read by
the last arriver; IWVI_FW_SLOW_TAIL): the two agree to the fixed point's resolution; a partial too large for its share of the field
takes the exact, tagged path (all chunks, or only the chunks that hold an outlier); integer adds commute, so the value is
bit-identical run after run."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)


def _both_tails(spec, zs, dev):
    from dgps_with_iwvi_amd import _abi, synthetic
    out = []
    for slow in (0, 1):
        _abi.set_debug_option("IWVI_FW_SLOW_TAIL", slow)
        try:
            model = synthetic.build_model(spec, dev)
            zd = [_t(z, dev) for z in zs]
            vals = [float(model.compute_log_likelihood(zd)) for _ in range(3)]
        finally:
            _abi.set_debug_option("IWVI_FW_SLOW_TAIL", 0)
        assert vals[0] == vals[1] == vals[2], vals                 # the same draws -> the same bits, whatever the order of arrival
        out.append(vals[0])
    return out


@pytest.mark.parametrize("L,M,K,B,lv", [(2, 128, 20, 1024, True), (2, 64, 5, 333, False), (1, 32, 10, 64, True)])
def test_packed_arrival_matches_the_stored_partials(gpu_device, L, M, K, B, lv):
    from dgps_with_iwvi_amd import synthetic
    spec = synthetic.make_spec(L=L, M=M, B=B, K=K, with_lv=lv, seed=3 * L + K, n_data=4 * B)
    zs = synthetic.make_noise(spec, seed=8)
    fast, slow = _both_tails(spec, zs, gpu_device)
    chunks = (B * K + 79) // 80 + 8                               # (80-sample chunks at these shapes; a few more for smaller ones)
    scale = spec["n_data"] / B
    assert abs(fast - slow) <= chunks * 2.0 ** -21 * scale + 1e-12 * abs(slow), (fast, slow)


@pytest.mark.parametrize("which", ["every chunk", "two chunks"])
def test_partials_beyond_the_fixed_point_range_take_the_exact_path(gpu_device, which):
    from dgps_with_iwvi_amd import synthetic
    spec = synthetic.make_spec(L=2, M=64, B=256, K=20, with_lv=True, seed=11, n_data=256)
    zs = synthetic.make_noise(spec, seed=12)
    if which == "every chunk":
        spec["lik_var"] = 1e-9                                    # residual^2 / (2 variance) ~ 1e8 per point: beyond 2^17 in every workgroup
    else:
        Y = np.array(spec["Y"], copy=True)
        Y[5] += 3.0e3; Y[200] -= 4.0e3                            # two outliers: (3e3)^2 / (2 * 0.01) = 4.5e8
        spec["Y"] = Y
    fast, slow = _both_tails(spec, zs, gpu_device)
    assert np.isfinite(fast) and abs(fast) > 2.0 ** 17
    assert abs(fast - slow) <= 1e-12 * abs(slow) + 64 * 2.0 ** -21, (fast, slow)


def test_many_large_partials_of_one_sign_do_not_overflow_the_field(gpu_device):
    """510 arrivals whose partial sums share a sign and mostly lie between 2^16 and 2^17 (oracle, float64: 315 of them; their sum is
    1.1 * 2^25, beyond the signed 46-bit field's +-2^25): a workgroup's share of the field is 2^16, so those go the exact way and the
    packed total cannot wrap."""
    from dgps_with_iwvi_amd import synthetic
    spec = synthetic.make_spec(L=1, M=32, B=2040, K=20, with_lv=False, seed=21, n_data=2040)
    spec["lik_var"] = 6e-5
    zs = synthetic.make_noise(spec, seed=22)
    fast, slow = _both_tails(spec, zs, gpu_device)
    assert np.isfinite(fast) and abs(fast) > 2.0 ** 25
    assert abs(fast - slow) <= 1e-12 * abs(slow) + 512 * 2.0 ** -21, (fast, slow)


def test_two_models_on_two_streams_do_not_interfere(gpu_device):
    """Two evaluations in flight (bench.py: ``two_in_flight``): two models with the same parameters, each with its own operand state, noise
    stream and arrival word, evaluated concurrently on two streams give the values they give one after the other -- nothing in the
    library is shared between launches except what the caller passes in."""
    from dgps_with_iwvi_amd import settings, synthetic
    spec = synthetic.make_spec(L=2, M=128, B=512, K=20, with_lv=True, seed=4, n_data=4096)

    def build():
        settings.set_seed(0)
        return synthetic.build_model(spec, gpu_device)

    ref = build()
    serial = [float(ref._build_likelihood()) for _ in range(6)]          # device-drawn noise: evaluation i of a fresh model, for every model
    a, b = build(), build()
    sa, sb = torch.cuda.Stream(device=gpu_device), torch.cuda.Stream(device=gpu_device)
    outs = {"a": [], "b": []}
    torch.cuda.synchronize()
    for _ in range(6):
        with torch.cuda.stream(sa):
            outs["a"].append(a._build_likelihood())
        with torch.cuda.stream(sb):
            outs["b"].append(b._build_likelihood())
    torch.cuda.synchronize()
    assert [float(v) for v in outs["a"]] == serial
    assert [float(v) for v in outs["b"]] == serial

"""bench.py's `roofline.l2_operand_stream`: the packed-operand bytes a workgroup of the layer kernel streams from L2, from the image sizes of
csrc/iwvi_common.h (state layout) -- pinned against the sizes written out by hand for the three bench configurations."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _stream(cfg_index, variant, cut_B=None):
    import bench
    from dgps_with_iwvi_amd import synthetic
    cfg = dict(bench.CONFIGS[cfg_index])
    spec = synthetic.make_spec(seed=0, parity=False, n_data=4096, **dict(cfg, B=cut_B or cfg["B"]))
    T = cfg["B"] * cfg["K"]
    return bench.l2_operand_stream(spec, T, 1.0, variant=variant), cfg


def test_headline_stack_bytes_per_workgroup():
    # configs[2]: LV layer + two GP layers, M = 128 (8 blocks), 80 samples per workgroup, split-f16 stage 2
    l2, cfg = _stream(2, 5 | 1 << 8, cut_B=64)
    tri, slabs = 36, sum((8 - (bi & ~1) + 1) // 2 for bi in range(8))        # fp32 solve blocks; split-f16 slabs of one tril(q_sqrt)^T image: 20
    assert slabs == 20
    inner = 8 * 3 * 256 + tri * 1024 + 5 * slabs * 2048 + 1 * 4 * 2048           # D = 9 -> 3 k-steps of Z~; R = 5 images; q_mu^T: 4 slabs
    final = 8 * 3 * 256 + tri * 1024 + 1 * slabs * 2048 + 1 * 4 * 2048           # D = 8, R = 1
    assert l2["bytes_per_workgroup"] == inner + final == 348160
    assert l2["samples_per_workgroup"] == 80 and l2["workgroups"] == cfg["B"] * cfg["K"] // 80 == 256
    assert abs(l2["TB_per_s"] - l2["bytes_per_launch"] / 1e-3 / 1e12) < 1e-12


def test_wide_stacks_bytes_per_workgroup():
    l3, _ = _stream(3, 5 | 1 << 8 | 1 << 9, cut_B=64)
    l4, cfg4 = _stream(4, 3 | 1 << 8 | 1 << 9, cut_B=64)
    assert l3["bytes_per_workgroup"] == 2125824 and l4["bytes_per_workgroup"] == 14688256      # (the numbers of profiles/r05i_cfg3/4_bench.json)
    assert l4["samples_per_workgroup"] == 48 and l4["workgroups"] == (cfg4["B"] * cfg4["K"] + 47) // 48 == 17067
    # the fp32 route streams fp32 blocks instead of the split-f16 slabs: the same bytes per element, one image less per slab pair
    f4, _ = _stream(4, 3 | 1 << 9, cut_B=64)
    assert 0.9 < f4["bytes_per_workgroup"] / l4["bytes_per_workgroup"] < 1.1

"""``settings.f64_stage1``: which GP layers take the float64 stage-1 route (host logic; the route itself: tests/test_gpu_f64_route.py)."""
import pytest


def test_auto_rule_is_by_input_dimension_and_overrides_win():
    from dgps_with_iwvi_amd import settings
    assert settings.f64_stage1 == "auto" and settings.f64_auto_max_dim == 3
    assert [settings.use_f64_stage1(d) for d in (1, 2, 3, 4, 8, 9)] == [True, True, True, False, False, False]
    assert settings.use_f64_stage1(8, True) is True and settings.use_f64_stage1(1, False) is False      # a layer's own choice
    with settings.temp_settings(f64_stage1="on"):
        assert settings.use_f64_stage1(8) and not settings.use_f64_stage1(8, False)
    with settings.temp_settings(f64_stage1="off"):
        assert not settings.use_f64_stage1(1) and settings.use_f64_stage1(1, True)
    assert settings.f64_stage1 == "auto"
    with settings.temp_settings(f64_stage1="sometimes"):
        with pytest.raises(ValueError):
            settings.use_f64_stage1(1)


def test_flags_of_the_header_and_the_bindings_agree():
    import os
    import re
    from dgps_with_iwvi_amd import _abi
    h = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "iwvi_hip.h")).read()
    val = lambda name: int(re.search(r"#define\s+%s\s+(\d+)" % name, h).group(1))
    assert val("IWVI_LAYER_F64_STAGE1") == _abi.LAYER_F64_STAGE1 and val("IWVI_GP_F64_STAGE1") == _abi.GP_F64_STAGE1
    assert val("IWVI_LAYER_F32_STAGE2") == _abi.LAYER_F32_STAGE2 and val("IWVI_GP_WANT_DENSE") == _abi.GP_WANT_DENSE and val("IWVI_GP_WANT_LM") == _abi.GP_WANT_LM


def test_split16_variance_rule():
    """M > 240 and variance / jitter >= 2^30: the layer asks for the fp32-MFMA variant (the split-f16 image of the super-block inverses
    is bounded by sigma / sqrt(jitter), which leaves the f16 range there)."""
    from dgps_with_iwvi_amd import settings
    with settings.temp_settings(jitter=1e-6):
        assert settings.split16_variance_ok(512, 1.0) and settings.split16_variance_ok(256, 1000.0)
        assert not settings.split16_variance_ok(256, 1100.0) and not settings.split16_variance_ok(512, 1e6)
        assert settings.split16_variance_ok(128, 1e9)            # M <= 240: no explicit inverses in the solve
    with settings.temp_settings(jitter=1e-9):
        assert not settings.split16_variance_ok(256, 2.0) and settings.split16_variance_ok(256, 0.5)

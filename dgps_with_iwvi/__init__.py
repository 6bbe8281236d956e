"""Drop-in import name of the reference package (``from dgps_with_iwvi.layers import GPLayer`` as in the reference's
tests/test_gp_layer.py:11-12 and experiments/build_models.py): every name resolves to the MI355X implementation in
``dgps_with_iwvi_amd``.  The GPflow-1.x pieces the reference imports from ``gpflow`` (kernels, features, likelihoods,
mean_functions, settings) are re-exported as submodules of the same names."""
import sys

import dgps_with_iwvi_amd as _impl
from dgps_with_iwvi_amd import (features, kernels, layers, likelihoods, mean_functions, models,  # noqa: F401
                                settings, temp_workaround)

for _name in ("layers", "models", "temp_workaround", "kernels", "features", "likelihoods", "mean_functions", "settings"):
    sys.modules[__name__ + "." + _name] = getattr(_impl, _name)

__all__ = list(_impl.__all__)
globals().update({k: getattr(_impl, k) for k in _impl.__all__})

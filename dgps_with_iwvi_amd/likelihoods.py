"""Gaussian likelihood holder (reference: experiments/build_models.py:198-199; models.py:66,105,134).
``variational_expectations`` is fused into ``iwvi_iw_elbo_reduce``."""


class Gaussian:
    def __init__(self, variance=1.0, name=None):
        self.variance = float(variance)
        self.name = name

    def predict_mean_and_var(self, Fmu, Fvar):
        return Fmu, Fvar + self.variance

"""``gpytorch.mlls.marginal_log_likelihood.MarginalLogLikelihood`` (isinstance target
at ``pgmuvi/trainers.py:128``)."""
from ..module import Module


class MarginalLogLikelihood(Module):
    def __init__(self, likelihood, model):
        super().__init__()
        self.likelihood = likelihood
        self.model = model

    def forward(self, output, target, **kwargs):
        raise NotImplementedError

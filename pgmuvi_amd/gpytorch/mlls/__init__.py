"""``gpytorch.mlls.ExactMarginalLogLikelihood`` (row A4; constructed at
``pgmuvi/trainers.py:119``, evaluated at ``:180``):

    mll = [ log N(y | m, K + Sigma) + sum_priors log p(theta) ] / N.
"""
from . import marginal_log_likelihood
from .marginal_log_likelihood import MarginalLogLikelihood
from ..distributions import MultivariateNormal
from ..likelihoods import _GaussianLikelihoodBase


class ExactMarginalLogLikelihood(MarginalLogLikelihood):
    def __init__(self, likelihood, model):
        if not isinstance(likelihood, _GaussianLikelihoodBase):
            raise RuntimeError("Likelihood must be Gaussian for exact inference")
        super().__init__(likelihood, model)

    def _add_other_terms(self, res, params):
        for _name, module, prior, closure, _ in self.named_priors():
            res = res + prior.log_prob(closure(module)).sum()
        return res

    def forward(self, function_dist, target, *params, **kwargs):
        if not isinstance(function_dist, MultivariateNormal):
            raise RuntimeError("ExactMarginalLogLikelihood can only operate on Gaussian random variables")
        output = self.likelihood(function_dist, *params, **kwargs)
        res = output.log_prob(target)
        res = self._add_other_terms(res, params)
        num_data = function_dist.event_shape.numel()
        return res / num_data

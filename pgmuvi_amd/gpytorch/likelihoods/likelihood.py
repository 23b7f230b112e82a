"""``gpytorch.likelihoods.likelihood.Likelihood`` (isinstance target at
``pgmuvi/lightcurve.py:2757``)."""
from ..module import Module


class Likelihood(Module):
    def __init__(self, max_plate_nesting=1):
        super().__init__()
        self.max_plate_nesting = max_plate_nesting

    def marginal(self, function_dist, *args, **kwargs):
        raise NotImplementedError

    def __call__(self, input, *args, **kwargs):
        from ..distributions import MultivariateNormal
        if isinstance(input, MultivariateNormal):
            return self.marginal(input, *args, **kwargs)
        raise RuntimeError(
            "Likelihoods expects a MultivariateNormal input to make marginal predictions. Got a {}".format(
                input.__class__.__name__))

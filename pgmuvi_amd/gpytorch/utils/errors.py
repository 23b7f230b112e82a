"""Exception types of the GPyTorch boundary (``pgmuvi/lightcurve.py:6004, 6024``)."""


class NanError(RuntimeError):
    pass


class NotPSDError(RuntimeError):
    pass


class NumericalWarning(RuntimeWarning):
    pass

"""Import-level stubs for ``pgmuvi/gps.py:22-23`` (out of scope: unreachable from fit())."""


class _OutOfScope:
    def __init__(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__} is outside the scope of pgmuvi_amd")


class CholeskyVariationalDistribution(_OutOfScope): pass
class VariationalStrategy(_OutOfScope): pass

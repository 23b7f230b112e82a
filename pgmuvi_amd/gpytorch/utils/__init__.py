from . import errors  # noqa: F401

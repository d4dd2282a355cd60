from .synthetic_rware import SyntheticRware, make  # noqa: F401

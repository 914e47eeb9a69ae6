"""``opt_einsum.backends`` names used at reference einsum.py:28 and :335."""
import importlib


def get_func(func, backend="numpy", default=None):
    lib = importlib.import_module(backend)
    return getattr(lib, func) if default is None else getattr(lib, func, default)


def has_einsum(backend):
    return True

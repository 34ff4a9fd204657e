"""TEST INFRASTRUCTURE -- loads oracle/_ref/deform_conv_ext_ref.so: the REFERENCE's own pybind module `deform_conv_ext`
(deform_conv_ext.cpp:150-164) built for gfx950 by oracle/build_ref.py.  Only tests/ may import this; it needs a GPU to run."""
import importlib.util
import os

from . import build_ref

_mod = None


def path():
    return build_ref.OUT


def exists():
    return os.path.isfile(build_ref.OUT)


def load():
    """The reference's module object: .modulated_deform_conv_forward / _backward, .deform_conv_forward / _backward_input /
    _backward_parameters with the reference's positional signatures."""
    global _mod
    if _mod is None:
        import torch  # noqa: F401  (libtorch must be loaded before the extension)

        spec = importlib.util.spec_from_file_location(build_ref.NAME, build_ref.OUT)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        _mod = m
    return _mod

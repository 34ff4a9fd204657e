"""ctypes binding of libglare_hip.so (the C ABI declared in include/glare_hip.h).

The product path has no fallback: if the shared library is missing or a symbol cannot be
resolved, importing/using an op raises.  torch is imported first so that the HIP runtime the
library resolves (`libamdhip64.so.7`) is the one torch already loaded -- device pointers and
streams are then shared between torch and the kernels.
"""
import ctypes
import os
import re

import torch  # noqa: F401  (must precede the CDLL load, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libglare_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "glare_hip.h")

_lib = None


class GlareError(RuntimeError):
    pass


def header_symbols():
    """Names of every function declared in include/glare_hip.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(glare_[a-z0-9_]+)\s*\(", text)))


def lib():
    """The loaded CDLL; raises GlareError when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GlareError(
                "libglare_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` -- there is no CPU or eager fallback." % LIB_PATH)
        try:
            _lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        except OSError as e:  # pragma: no cover
            raise GlareError("cannot load %s: %s" % (LIB_PATH, e))
        _lib.glare_status_string.restype = ctypes.c_char_p
        _lib.glare_status_string.argtypes = [ctypes.c_int]
        _lib.glare_version.restype = ctypes.c_int
    return _lib


def check(status, what):
    if status != 0:
        raise GlareError("%s failed: %s (%d)" % (what, lib().glare_status_string(status).decode(), status))


def ptr(t):
    """Device pointer of a tensor (or NULL for None) as a ctypes void*."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def stream_handle():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NotImplementedError(
                "glare_amd ops run on MI355X only (HIP kernels, no CPU path); got a %s tensor" % t.device)

"""ctypes binding of libglare_hip.so (the C ABI declared in include/glare_hip.h).

The product path has no fallback: if the shared library is missing or a symbol cannot be
resolved, importing/using an op raises.  torch is imported first so that the HIP runtime the
library resolves (`libamdhip64.so.7`) is the one torch already loaded -- device pointers and
streams are then shared between torch and the kernels.
"""
import ctypes
import os
import re
import threading

import torch  # noqa: F401  (must precede the CDLL load, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libglare_hip.so")
LIB_F16_PATH = os.path.join(_HERE, "libglare_hip_f16.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "glare_hip.h")

_lib = None
_lib_f16 = None
# The 16-bit activation / filter format of the kernels in use: "bf16" (libglare_hip.so) or "fp16" (libglare_hip_f16.so).
# None = not set by any enclosing use_precision() of THIS thread: plain ops (and everything with a tape) then run bf16, the inference entry
# points (VQLLFLOWDeformable.reverse_flow_nhwc, glare_amd.infer, bench.py) run INFERENCE_PRECISION.
# Thread-local: the reference boundary is entered from one thread per GPU under nn.DataParallel, and a precision
# switched by one thread's context manager must not leak into another's launches.
_TLS = threading.local()
INFERENCE_PRECISION = "fp16"


def _current():
    return getattr(_TLS, "precision", None)


class GlareError(RuntimeError):
    """`status`: the numeric GLARE_ERR_* code of include/glare_hip.h (None for errors raised on the Python side)."""

    def __init__(self, msg, status=None):
        super().__init__(msg)
        self.status = status


ERR_INVALID, ERR_LAUNCH, ERR_WORKSPACE, ERR_UNSUPPORTED = -1, -2, -3, -4     # include/glare_hip.h:37-40


def header_symbols():
    """Names of every function declared in include/glare_hip.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(glare_[a-z0-9_]+)\s*\(", text)))


def precision():
    return _current() or "bf16"


def inference_precision(requested=None):
    """Precision of an inference entry point: the caller's request, else the enclosing use_precision(), else fp16 -- the
    reference's own autocast dtype (infer_dataset_lol.py:134) and what the end-to-end tolerance needs."""
    return requested or _current() or INFERENCE_PRECISION


def act_dtype():
    """torch dtype of 16-bit activations / packed filters under the current precision."""
    return torch.bfloat16 if precision() == "bf16" else torch.float16


class use_precision:
    """`with use_precision("fp16"):` -- ops run the IEEE-half twin of the inference kernels (libglare_hip_f16.so: same kernels,
    same MFMA rate, 11 instead of 8 mantissa bits in every stored activation and filter: the path's parity mode; the
    reference's own autocast dtype, infer_dataset_lol.py:134).  None keeps the current one."""

    def __init__(self, name):
        assert name in (None, "bf16", "fp16"), name
        self.name = name

    def __enter__(self):
        self._prev = _current()
        if self.name is not None:
            _TLS.precision = self.name
        return self

    def __exit__(self, *exc):
        _TLS.precision = self._prev


def keeps_precision(cls):
    """Class decorator for torch.autograd.Function subclasses: the backward runs under the precision the forward ran under.
    The autograd engine calls `backward` from ITS OWN worker threads, where this module's thread-local precision is unset -- without
    this an fp16 training step would run its backward kernels of the bf16 library on fp16 tensors."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args, **kw):
        ctx._glare_precision = _current()
        return fwd(ctx, *args, **kw)

    def backward(ctx, *grads):
        with use_precision(getattr(ctx, "_glare_precision", None)):
            return bwd(ctx, *grads)

    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


# entry points without any 16-bit tensor in their signature: under "fp16" they resolve to the main library
_DTYPE_AGNOSTIC = ("glare_vq_", "glare_harness_", "glare_ssim_")


class _F16Lib:
    """libglare_hip_f16.so behind the main library's names: `x_bf16` resolves to its export `x_f16`.  Since round 4 the half
    library is a build of EVERY source (inference and training kernels); an entry point it should ever lack raises -- it must
    never silently run a bf16 kernel on fp16 data."""

    def __init__(self, cdll):
        self._c = cdll

    def __getattr__(self, name):          # reached on the FIRST use of a name only: the resolved function is kept on the instance
        try:
            fn = getattr(self._c, name.replace("bf16", "f16"))
        except AttributeError:
            if not name.startswith(_DTYPE_AGNOSTIC):
                raise GlareError("%s is not part of libglare_hip_f16.so (the fp16 precision covers the inference kernels)" % name)
            fn = getattr(main_lib(), name)
        self.__dict__[name] = fn
        return fn


def lib():
    """The library of the current precision (use_precision); raises GlareError when the HIP extension has not been built."""
    if precision() == "fp16":
        global _lib_f16
        if _lib_f16 is None:
            main_lib()
            if not os.path.exists(LIB_F16_PATH):
                raise GlareError("libglare_hip_f16.so is not built (%s): run __graft_entry__.build()" % LIB_F16_PATH)
            c = ctypes.CDLL(LIB_F16_PATH, mode=ctypes.RTLD_LOCAL)
            c.glare_status_string.restype = ctypes.c_char_p
            c.glare_status_string.argtypes = [ctypes.c_int]
            _lib_f16 = _F16Lib(c)
        return _lib_f16
    return main_lib()


def main_lib():
    """libglare_hip.so (bf16 activations; every entry point of include/glare_hip.h)."""
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
        raise GlareError("%s failed: %s (%d)" % (what, main_lib().glare_status_string(status).decode(), status), status)


def ptr(t):
    """Device pointer of a tensor (or NULL for None) as a ctypes void*."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_handle():
    """The calling thread's current HIP stream as a void* (what every launch of the C ABI takes).  Through torch's two C entry points:
    `torch.cuda.current_stream().cuda_stream` builds a Stream object behind three Python-level device lookups -- 9 us per call, measured
    (tools/probes/train_host_profile.py), i.e. ~12 ms of the ~1 400 launches of a training step that the HOST paces (round 6)."""
    if _raw_stream is not None:
        try:
            return ctypes.c_void_p(_raw_stream(_cur_device()))
        except Exception:      # no initialised HIP context (CPU-side tests of the error paths): the documented route decides what to raise
            pass
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NotImplementedError(
                "glare_amd ops run on MI355X only (HIP kernels, no CPU path); got a %s tensor" % t.device)

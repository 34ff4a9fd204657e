"""CPU: the C-ABI library loads without a GPU and exports every symbol include/glare_hip.h
declares; the product refuses CPU tensors instead of falling back."""
import os
import re
import subprocess

import pytest
import torch

from glare_amd import _lib


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    names = _lib.header_symbols()
    assert "glare_vq_nearest_f32" in names and "glare_version" in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.glare_version() >= 100
    assert lib.glare_status_string(0) == b"ok"
    assert lib.glare_status_string(-1) == b"invalid argument"


def test_glare_error_carries_the_numeric_status():
    """callers decide on `e.status` (a GLARE_ERR_* code of the header), never on the wording of the message."""
    src = open(_lib.HEADER_PATH).read()
    for name, val in (("INVALID", _lib.ERR_INVALID), ("LAUNCH", _lib.ERR_LAUNCH), ("WORKSPACE", _lib.ERR_WORKSPACE), ("UNSUPPORTED", _lib.ERR_UNSUPPORTED)):
        assert re.search(r"#define GLARE_ERR_%s \(%d\)" % (name, val), src), name
    with pytest.raises(_lib.GlareError) as ei:
        _lib.check(_lib.ERR_UNSUPPORTED, "glare_something")
    assert ei.value.status == _lib.ERR_UNSUPPORTED and "glare_something" in str(ei.value)
    assert _lib.GlareError("raised on the Python side").status is None
    _lib.check(0, "fine")
    # the DCN's single-pass opt-in falls back on the status, not on the text
    dd = open(os.path.join(os.path.dirname(_lib.__file__), "modules", "deformableDecoder_arch.py")).read()
    assert "e.status != _lib.ERR_UNSUPPORTED" in dd and '"unsupported" not in' not in dd


def test_no_extra_exports():
    """Everything exported with the glare_ prefix is declared in the header (the ABI is the header)."""
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (glare_[a-z0-9_]+)", out))
    assert exported == set(_lib.header_symbols())


def test_library_is_gfx950_only():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", _lib.LIB_PATH], capture_output=True, text=True)
    data = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in data
    for other in (b"gfx942", b"gfx90a", b"sm_80"):
        assert other not in data


def test_product_refuses_cpu_tensors():
    from glare_amd import ops

    z = torch.zeros(4, 3)
    cb = torch.zeros(16, 3)
    with pytest.raises(NotImplementedError):
        ops.vq_nearest(z, cb)


def test_half_library_exports_the_inference_entry_points():
    """libglare_hip_f16.so (IEEE-half activations / filters; since round 4 a build of EVERY source, training kernels included):
    each export is declared in the header -- a *_bf16 entry point under the name *_f16, identical signature -- and it exports
    nothing the header does not declare."""
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_F16_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (glare_[a-z0-9_]+)", out))
    declared = set(_lib.header_symbols())
    as_f16 = {n.replace("bf16", "f16") for n in declared}
    assert exported <= as_f16, sorted(exported - as_f16)
    for must in ("glare_conv2d_f16", "glare_conv1x1_ws_f16", "glare_attention_kv512_f16", "glare_groupnorm_apply_f16",
                 "glare_mdcn_forward_nhwc", "glare_flow_h1_f32", "glare_flow_tail_f32", "glare_conv2d_smallcin_f32", "glare_mix_f16"):
        assert must in exported, must
    assert not any("bf16" in n for n in exported)
    data = open(_lib.LIB_F16_PATH, "rb").read()
    assert b"gfx950" in data and b"gfx942" not in data
    # the precision switch resolves names through it, and refuses what it does not have
    from glare_amd import ops

    with ops.use_precision("fp16"):
        assert ops.act_dtype() == torch.float16
        assert _lib.lib().glare_conv2d_bf16 is not None          # -> glare_conv2d_f16
        with pytest.raises(_lib.GlareError):
            _lib.lib().glare_wgrad_nhwc_bf16
        assert _lib.lib().glare_vq_nearest_f32 is not None       # dtype-agnostic: the main library's
    assert ops.act_dtype() == torch.bfloat16
    assert ops.inference_precision() == "fp16" and ops.inference_precision("bf16") == "bf16"

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

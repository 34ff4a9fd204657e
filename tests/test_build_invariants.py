"""CPU: properties of the generated ISA that the kernels rely on (hipcc cross-compiles without a GPU).

dcn_fwd_fast_kernel must not contain packed-fp32 VALU ops: with v_pk_mul_f32 / v_pk_fma_f32 writing registers that MFMAs
issued just before were still reading, the kernel was nondeterministic at the full-size shapes (dcn.hip header comment)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_dcn_forward_kernels_have_no_packed_fp32(tmp_path):
    import sys

    sys.path.insert(0, os.path.join(ROOT, "glare_amd", "csrc"))
    import build as B

    out = tmp_path / "dcn.s"
    cmd = [HIPCC] + [f for f in B.COMMON if f != "-fPIC"] + B.PER_FILE["dcn.hip"] + [
        "-S", "--cuda-device-only", os.path.join(ROOT, "glare_amd", "csrc", "dcn.hip"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    kernels = re.findall(r"^(_ZN\S*dcn_fwd_(?:fast_)?kernel\S*):[^\n]*\n(.*?)s_endpgm", asm, flags=re.S | re.M)
    assert sum("fast" in name for name, _ in kernels) >= 6, "expected one fast-kernel body per (NT, NCH) instantiation"
    assert any("v_mfma" in body for _, body in kernels)
    for name, body in kernels:   # the general kernel blends right behind its MFMAs too: same rule
        packed = re.findall(r"v_pk_(?:fma|mul|add)_f32", body)
        assert not packed, "%s: %d packed-fp32 ops" % (name, len(packed))

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


CONV_UNITS = ("conv_igemm_k3s1.hip", "conv_igemm_k3s1_planar.hip", "conv_igemm_k3s1_general.hip", "conv_igemm_k3s2.hip", "conv_igemm_k1k2.hip",
              "conv_igemm_general.hip")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("half", [False, True])
def test_conv_kernels_build_without_scratch(half):
    """Round 5: every instantiation of conv_igemm_kernel compiles with 0 B of scratch, in both libraries.  With several epilogues and
    activation clones inside one kernel (rounds 2-4) hipcc never reused a dead accumulator register: 80-500 B/lane of scratch, every
    residual piece loaded into the same four registers.  One epilogue per instantiation (EPI_* template parameter) is what keeps it
    at zero; this test is the guard against a second accumulator-consuming region creeping back into the kernel."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "glare_amd", "csrc"))
    import build as B

    seen = 0
    for unit in CONV_UNITS:
        cmd = [HIPCC] + [f for f in B.COMMON if f != "-fPIC"] + (["-DGLARE_ACT_F16"] if half else []) + [
            "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(ROOT, "glare_amd", "csrc", unit), "-o", os.devnull]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        names = re.findall(r"Function Name: (\S+)", r.stderr)
        scratch = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
        assert len(names) == len(scratch) and names, unit
        for n, sc in zip(names, scratch):
            if "conv_igemm_kernel" in n:
                seen += 1
                assert sc == 0, "%s (%s): %d B/lane of scratch" % (n, unit, sc)
    assert seen == 29, "expected 29 instantiations of conv_igemm_kernel, saw %d" % seen

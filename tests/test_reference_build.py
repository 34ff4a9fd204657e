"""CPU: the checker binary built from the reference's own DCN sources (oracle/build_ref.py -> oracle/_ref/deform_conv_ext_ref.so).
Runs where /root/reference is present; on the GPU box the prebuilt file travels and tests/test_gpu_dcn_reference.py executes it."""
import os
import subprocess

import pytest

from oracle import build_ref, ref_ext


@pytest.mark.skipif(not build_ref.available(), reason="/root/reference is not present")
def test_reference_extension_builds_for_gfx950_and_exports_the_five_entry_points():
    so = build_ref.build()
    assert so and os.path.isfile(so)
    m = ref_ext.load()          # importing needs libtorch only, no GPU
    for name in ("modulated_deform_conv_forward", "modulated_deform_conv_backward", "deform_conv_forward",
                 "deform_conv_backward_input", "deform_conv_backward_parameters"):      # deform_conv_ext.cpp:150-164
        assert callable(getattr(m, name))
    # device code of the reference's kernels for gfx950 is inside (hipcc cross-compiled it: no GPU here)
    assert b"hipv4-amdgcn-amd-amdhsa--gfx950" in open(so, "rb").read()      # the offload bundle's target id


@pytest.mark.skipif(not build_ref.available(), reason="/root/reference is not present")
def test_no_reference_text_inside_the_repository():
    """The recipe translates in a scratch directory outside the tree and keeps the shared object only."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert sorted(os.listdir(os.path.join(root, "oracle", "_ref"))) == ["deform_conv_ext_ref.so"]
    tracked = subprocess.run(["git", "ls-files"], cwd=root, capture_output=True, text=True).stdout.split()
    assert not [f for f in tracked if f.startswith("oracle/_ref/") or f.endswith((".cu", ".cuh"))]

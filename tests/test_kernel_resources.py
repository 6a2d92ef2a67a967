"""The register / scratch budget of the shipped gfx950 kernels (profiles/kernel_resources.py: the code objects' metadata notes, read with the
ROCm LLVM tools - no GPU): no kernel of the timed step may use scratch.  A register spill in a hot loop passes every numerics test and
costs time nobody sees (decode_bwd_kernel and bn_bwd_apply_kernel shipped with spills that no test looked for)."""

import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "lightning-pose_amd", "liblp_hip.so")
LLVM = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")


@pytest.mark.skipif(not (os.path.exists(SO) and os.path.exists(os.path.join(LLVM, "llvm-readelf")) and shutil.which("c++filt")),
                    reason="needs the built liblp_hip.so and the ROCm LLVM tools")
def test_no_hot_kernel_uses_scratch():
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    try:
        import kernel_resources as KR
    finally:
        sys.path.pop(0)
    ks = KR.kernels(SO)
    assert len(ks) > 150, len(ks)
    hot = [k for k in ks if k["name"].startswith(KR.HOT)]
    assert any(k["name"].startswith("lp::conv_pipe_kernel") for k in hot) and any(k["name"].startswith("lp::decode_bwd_kernel") for k in hot)
    bad = KR.hot_with_scratch(ks)
    assert not bad, [(k["name"], k["scratch"], k["vgpr_spill"]) for k in bad]
    # one workgroup of 512 threads per CU: the pipelined convolutions may use the whole register file, not more
    for k in hot:
        if k["name"].startswith(("lp::conv_pipe_kernel", "lp::conv_wgrad_pipe_kernel")):
            assert k["vgpr"] + k["agpr"] <= 256 and k["lds"] <= 160 * 1024, k

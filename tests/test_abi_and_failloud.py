"""C-ABI surface and fail-loud behaviour (CPU only, no compute calls on the GPU library)."""

import os
import re
import subprocess

import pytest
import torch

from tests.conftest import ROOT

HEADER = os.path.join(ROOT, "include", "lp_hip.h")
SO = os.path.join(ROOT, "lightning-pose_amd", "liblp_hip.so")


def header_symbols() -> set[str]:
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(lp_[a-z0-9_]+)\s*\(", txt))


def test_header_matches_python_prototypes():
    from lightning_pose_amd import _lib

    assert header_symbols() == set(_lib.PROTOTYPES)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(SO):
        subprocess.run(["bash", os.path.join(ROOT, "lightning-pose_amd", "csrc", "build.sh")], check=True, capture_output=True)
    out = subprocess.run(["nm", "-D", "--defined-only", SO], check=True, capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (lp_[a-z0-9_]+)", out))
    assert header_symbols() <= exported, header_symbols() - exported


def test_gfx950_code_object_present():
    """the shared library must carry gfx950 device code for the kernels (not a host-only stub)"""
    out = subprocess.run(["strings", "-n", "6", SO], check=True, capture_output=True, text=True).stdout
    assert "gfx950" in out
    assert "conv_igemm_kernel" in out and "decode_fwd_kernel" in out


def test_ops_refuse_cpu_tensors():
    from lightning_pose_amd import ops
    from lightning_pose_amd._lib import LpHipUnavailable

    hm = torch.rand(1, 2, 16, 16)
    with pytest.raises(LpHipUnavailable):
        ops.decode(hm, 2, 1000.0, ops.DecodeFrameMap(None, False, None, 1, 64, 64, 2))
    with pytest.raises(LpHipUnavailable):
        ops.heatmap_mse(hm, hm)
    with pytest.raises(LpHipUnavailable):
        ops.generate_heatmaps(torch.rand(1, 2, 2), 64, 64, (16, 16))


def test_engine_refuses_cpu_device():
    from lightning_pose_amd._lib import LpHipUnavailable
    from lightning_pose_amd.engine import Engine

    with pytest.raises(LpHipUnavailable):
        Engine(3, 2, "cpu")


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from lightning_pose_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.LpHipUnavailable, match="no CPU fallback"):
        _lib.lib()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "lightning-pose_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "hipemu" not in txt.replace("tests/hipemu", ""), f


def test_error_code_mapping():
    from lightning_pose_amd import _lib

    class Fake:
        @staticmethod
        def lp_strerror(code):
            return b"msg"

    import lightning_pose_amd._lib as L
    old = L._lib
    L._lib = Fake()
    try:
        with pytest.raises(ValueError):
            _lib.check(-1, "x")
        with pytest.raises(NotImplementedError):
            _lib.check(-2, "x")
        with pytest.raises(_lib.LpHipError):
            _lib.check(700, "x")
        _lib.check(0, "x")
    finally:
        L._lib = old

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


def test_abi_version_is_checked_not_only_the_symbol_names(monkeypatch):
    """round 5 inserted arguments into existing entry points (decode `prune`, bn_bwd `terms_ws`) and nothing compared lp_version()
    with what the binding expects - a stale library with all the symbols would have taken the stream for the new argument.  The header
    carries LP_HIP_ABI_VERSION, lp_version() returns the value the library was built with, and `declare` refuses any other."""
    from lightning_pose_amd import _lib
    from tests.hipemu import emu

    macro = int(re.search(r"#define\s+LP_HIP_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert macro == _lib.ABI_VERSION
    lib = emu.emu_lib()   # (the CPU build of the same sources: loadable here)
    assert lib.lp_version() == macro
    monkeypatch.setattr(_lib, "ABI_VERSION", macro + 1)
    with pytest.raises(_lib.LpHipUnavailable, match="ABI"):
        _lib.declare(lib)


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


def test_switches_are_read_at_load_not_per_call():
    """lp_hip.h, round 5: no entry point reads the environment - the LP_* A/B switches live in a table filled ONCE when the library is loaded;
    a later change of the environment has no effect until lp_config_reload_env() (the hook tests/conftest.py wires to monkeypatch.setenv).
    Checked on the emulated build of the same sources (the product library cannot be loaded without a device runtime)."""
    import os

    import numpy as np
    import torch

    from tests.hipemu import emu

    lib = emu.emu_lib()
    g = emu.geom(2, 16, 16, 256, 128, 1, 1, 1, 0)
    gen = torch.Generator().manual_seed(0)
    x = emu.to_bf16_bits(torch.randn(2, 16, 16, 256, generator=gen))
    w = emu.to_bf16_bits(torch.randn(128, 1, 1, 256, generator=gen) / 16)
    old = os.environ.get("LP_CONV_PIPE")
    try:
        os.environ.pop("LP_CONV_PIPE", None)
        lib.lp_config_reload_env()
        z_pipe, _ = emu.conv_fwd(x, w, g)
        assert lib.lp_conv_last_kernel() == 1      # LP_CONV_KERNEL_PIPE
        os.environ["LP_CONV_PIPE"] = "0"           # the environment changes ...
        emu.conv_fwd(x, w, g)
        assert lib.lp_conv_last_kernel() == 1      # ... the library does not look
        lib.lp_config_reload_env()
        z_igemm, _ = emu.conv_fwd(x, w, g)
        assert lib.lp_conv_last_kernel() == 0      # LP_CONV_KERNEL_IGEMM
        assert np.array_equal(z_pipe, z_igemm)
    finally:
        if old is None:
            os.environ.pop("LP_CONV_PIPE", None)
        else:
            os.environ["LP_CONV_PIPE"] = old
        lib.lp_config_reload_env()

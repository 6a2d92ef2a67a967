"""pytest configuration: markers, repo-root on sys.path, golden-fixture loader."""

import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT) if ROOT not in sys.path else None
import _lp_bootstrap  # noqa: E402,F401  registers lightning_pose_amd
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests need a device: on a box without one (a plain `pytest` in the build container) they are skipped, not failed"""
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs an MI355X (run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: executes the reference's own modules (/root/reference, or oracle/_ref on the GPU box)")


def reload_lp_switches() -> None:
    """The kernel libraries read their LP_* A/B switches once, at load (lp_hip.h: lp_config_reload_env).  Tests flip them in-process with
    monkeypatch.setenv / delenv: every library that is loaded right now (the product's, the emulated build) re-reads the environment."""
    from lightning_pose_amd import _lib
    from tests.hipemu import emu

    for lib in (_lib._lib, emu._emu):
        if lib is not None and hasattr(lib, "lp_config_reload_env"):
            lib.lp_config_reload_env()


@pytest.fixture(autouse=True)
def _lp_switches_follow_the_environment(monkeypatch):
    """start every test from the current environment, re-read after each LP_* edit, and once more after monkeypatch has undone them"""
    set_, del_ = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, *a, **k):
        set_(name, value, *a, **k)
        if str(name).startswith("LP_"):
            reload_lp_switches()

    def delenv(name, *a, **k):
        del_(name, *a, **k)
        if str(name).startswith("LP_"):
            reload_lp_switches()

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    reload_lp_switches()
    yield
    monkeypatch.undo()
    reload_lp_switches()


class Golden(dict):
    def t(self, key: str) -> torch.Tensor:
        return torch.from_numpy(np.asarray(self[key]))


@pytest.fixture(scope="session")
def golden():
    def _load(name: str) -> Golden:
        with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
            return Golden({k: z[k] for k in z.files})
    return _load


def has_reference() -> bool:
    """the reference's hot-path modules can be executed: /root/reference (build container) or the verbatim copy oracle/make_ref.py shipped
    to oracle/_ref/ (GPU box); tests that need the reference's DATA files check for /root/reference themselves"""
    from oracle import ref_loader

    return ref_loader.available()


needs_reference = pytest.mark.skipif(not has_reference(), reason="neither /root/reference nor oracle/_ref present")


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def kernel_backend(request):
    """Run a kernel-level test twice: on the CPU-emulated build of the kernel sources (default CPU suite) and, under
    `-m gpu`, on the real device through the product library liblp_hip.so."""
    from tests.hipemu import emu

    emu.BACKEND = request.param
    yield request.param
    emu.BACKEND = "emu"


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def stack_backend(request, monkeypatch):
    """Device for tests of the product host stack: 'cpu' with the kernel library swapped for its emulated build (CPU
    suite), or the real 'cuda:0' with nothing patched (`-m gpu`)."""
    from lightning_pose_amd import _lib, ops

    if request.param == "gpu":
        yield torch.device("cuda:0")
        return
    from tests.hipemu import emu

    monkeypatch.setattr(_lib, "_lib", emu.emu_lib())
    monkeypatch.setattr(ops, "require_device", lambda *a: None)
    monkeypatch.setattr(ops, "require_device_type", lambda d: None)
    monkeypatch.setattr(ops, "_stream", lambda: None)
    ops._device_tables.cache_clear()
    yield torch.device("cpu")
    ops._device_tables.cache_clear()

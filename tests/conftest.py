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


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


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
    return os.path.isdir("/root/reference/lightning_pose")


needs_reference = pytest.mark.skipif(not has_reference(), reason="/root/reference not present")

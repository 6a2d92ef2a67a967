"""Make the product package importable as ``lightning_pose_amd``.

The package lives in ``lightning-pose_amd/`` (the directory name the build contract prescribes); a hyphen is
not a valid Python identifier, so this shim registers that directory under the importable name.  Import this
module (tests/conftest.py, bench.py, __graft_entry__.py do) before ``import lightning_pose_amd``.
"""

import importlib.util
import os
import sys

_NAME = "lightning_pose_amd"
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lightning-pose_amd")


def ensure():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


ensure()

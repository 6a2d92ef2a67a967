"""Recipe for oracle/_ref/: the reference's OWN hot-path modules, shipped verbatim next to the oracle so that they travel to the GPU box.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  /root/reference exists in the build container only; bench.py's ``cpu_baseline`` leg and
the differential tests want the verbatim modules on the GPU box's host cores too ("kind": "reference" instead of the restated port).  This
script COPIES - byte for byte, nothing edited - the files oracle/ref_loader.py executes for the heat-map tracker's training step from
/root/reference/lightning_pose/ to oracle/_ref/lightning_pose/ and records their SHA-256 in oracle/_ref/MANIFEST.json.  oracle/_ref/ is
listed in .gitignore (reference sources never enter this repository's history) but NOT in .gpurunignore, so the directory ships with the
snapshot like the built .so files.  __graft_entry__.build() runs it whenever /root/reference is present.

    python oracle/make_ref.py            (build container)
"""

from __future__ import annotations

import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.environ.get("LP_REFERENCE_SRC", "/root/reference")
DST = os.path.join(HERE, "_ref")
# what `ref_loader.load("models.heatmap_tracker")`, `load("losses.factory")`, `load("utils.pca")`, `load("data.utils")`, `load("data.bboxes")`
# `load("callbacks")` and `load("models.factory")` (the reference's own get_model: tests/test_boundary_reference_factory.py) pull in (sys.modules after those loads), plus the ViT wrapper config C4 constructs
FILES = [
    "callbacks.py",
    "data/bboxes.py", "data/datatypes.py", "data/heatmaps.py", "data/utils.py",
    "losses/factory.py", "losses/losses.py",
    "models/base.py", "models/datatypes.py", "models/factory.py", "models/heatmap_tracker.py",
    "models/backbones/__init__.py", "models/backbones/factory.py", "models/backbones/vit.py",
    "models/heads/__init__.py", "models/heads/heatmap.py", "models/heads/heatmap_mhcrnn.py", "models/heads/regression.py",
    "utils/pca.py",
]


def main() -> int:
    pkg = os.path.join(SRC, "lightning_pose")
    if not os.path.isdir(pkg):
        print(f"make_ref: {pkg} not found - nothing to do (the GPU box uses the shipped copy)")
        return 0
    manifest = {}
    for rel in FILES:
        src, dst = os.path.join(pkg, rel), os.path.join(DST, "lightning_pose", rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(src, dst)
        with open(dst, "rb") as fh:
            manifest[rel] = hashlib.sha256(fh.read()).hexdigest()
    with open(os.path.join(DST, "MANIFEST.json"), "w") as fh:
        json.dump({"source": "paninski-lab/lightning-pose (the tree at /root/reference), files copied verbatim", "sha256": manifest}, fh, indent=1)
    print(f"make_ref: {len(FILES)} files -> {DST}")
    return 0


if __name__ == "__main__":
    sys.exit(main())

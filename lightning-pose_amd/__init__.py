"""lightning_pose_amd - MI355X-native (gfx950) training path for Lightning Pose heatmap trackers.

Drop-in for ONE hot path of paninski-lab/lightning-pose: the supervised + unsupervised heatmap-tracker
training step.  The arithmetic is hand-written HIP in ``csrc/`` behind the C ABI of ``include/lp_hip.h``
(``liblp_hip.so``); this package mirrors the reference's ``lightning_pose.models`` / ``lightning_pose.losses``
/ ``lightning_pose.data`` interface for that path (same names, argument meaning and error behaviour).

There is NO CPU fallback: every op raises ``LpHipUnavailable`` if ``liblp_hip.so`` is missing or the tensors
are not on a ROCm device.
"""

__version__ = "0.1.1"

from . import _lib  # noqa: F401  (does not load the shared library until first use)

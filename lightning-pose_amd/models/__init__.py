"""lightning_pose.models surface for the heatmap-tracker hot path."""

from .factory import get_model, get_model_class  # noqa: F401
from .heatmap_tracker import HeatmapTracker, SemiSupervisedHeatmapTracker  # noqa: F401

ALLOWED_MODELS = (HeatmapTracker, SemiSupervisedHeatmapTracker)

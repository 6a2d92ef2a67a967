"""lightning_pose.losses surface for the heatmap-tracker hot path."""

from .factory import LossFactory, get_loss_classes, get_loss_factories  # noqa: F401
from .losses import (  # noqa: F401
    HeatmapLoss, HeatmapMSELoss, Loss, PCALoss, RegressionRMSELoss, TemporalHeatmapLoss, TemporalLoss, UnimodalLoss,
)

from .heatmap import HeatmapHead, run_subpixelmaxima  # noqa: F401

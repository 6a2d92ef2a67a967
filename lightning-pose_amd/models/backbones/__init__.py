from .factory import BACKBONE_STRIDES, backbone_features  # noqa: F401

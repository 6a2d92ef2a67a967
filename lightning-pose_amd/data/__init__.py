"""Pure functions of lightning_pose.data that sit on the training hot path."""

"""The reference's `empose/nn/loss.py` names (the functions live next to the models that use them)."""
from em_pose_amd.nn.models import normal_mse, padded_loss, reconstruction_loss  # noqa: F401

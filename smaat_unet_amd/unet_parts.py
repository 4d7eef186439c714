"""Drop-in for the one class of /root/reference/models/unet_parts.py on the hot path."""
from .unet_parts_depthwise_separable import OutConv  # noqa: F401  (reference: models/unet_parts.py:67-73)

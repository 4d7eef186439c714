"""smaat_unet_amd -- MI355X (gfx950) native forward+backward path for SmaAt-UNet.

Hand-written HIP kernels (libsmaat_hip.so, C ABI in include/smaat_hip.h) behind module
classes that mirror the reference's `models/` API.  There is no CPU fallback.
"""
from . import _lib  # noqa: F401
from . import ops as _ops  # noqa: F401
from .ops import precision  # noqa: F401  (context manager: "f32" | "bf16" mixed precision for the modules called inside)
from . import torch_ops  # noqa: F401  (registers torch.ops.smaat.* inference operators)
from . import train_ops  # noqa: F401  (registers the torch.ops.smaat.* training operators with autograd formulas)
from .train_ops import traceable_training  # noqa: F401
from .SmaAt_UNet import SmaAt_UNet  # noqa: F401
from .layers import CBAM, ChannelAttention, DepthwiseSeparableConv, SpatialAttention  # noqa: F401
from .unet_parts_depthwise_separable import DoubleConvDS, DownDS, OutConv, UpDS  # noqa: F401
from .unet_precip_variants import UNetDS, UNetDSAttention, UNetDSAttention4CBAMs  # noqa: F401
from .metrics import PrecipitationMetrics  # noqa: F401

__all__ = ["SmaAt_UNet", "CBAM", "ChannelAttention", "SpatialAttention", "DepthwiseSeparableConv", "DoubleConvDS",
           "DownDS", "UpDS", "OutConv", "UNetDS", "UNetDSAttention", "UNetDSAttention4CBAMs", "PrecipitationMetrics", "precision", "traceable_training"]

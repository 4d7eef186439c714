"""Sibling architectures of SmaAt-UNet built from the same MI355X blocks.

Drop-ins for the network definitions of /root/reference/models/unet_precip_regression_lightning.py:
  UNetDS                  (:86-118)   depthwise-separable U-Net without attention
  UNetDSAttention         (:121-164)  = the SmaAt-UNet wiring (CBAM on every level, bottleneck included)
  UNetDSAttention4CBAMs   (:167-208)  CBAM on the four skip connections only (un-attended bottleneck)
Same submodule names (=> the reference `state_dict` keys of each network, checked against the reference
blocks in oracle/gen_golden.py) and the same forward wiring (`SmaAt_UNet.UNetDSFamily`).  The reference classes
are Lightning modules configured through an `hparams` namespace; these take either that namespace / a dict
(`hparams=...`) or plain keyword arguments -- the training loop around them (Lightning) is
out of scope.
"""
from __future__ import annotations

from .SmaAt_UNet import UNetDSFamily

_DEFAULTS = dict(n_channels=12, n_classes=1, kernels_per_layer=2, bilinear=True, reduction_ratio=16)


def _settings(hparams, kw):
    cfg = dict(_DEFAULTS)
    if hparams is not None:
        src = hparams if isinstance(hparams, dict) else vars(hparams)
        cfg.update({k: src[k] for k in _DEFAULTS if k in src})  # other Lightning fields (lr, paths ...) are ignored
    unknown = set(kw) - set(_DEFAULTS)
    if unknown:
        raise TypeError(f"unexpected arguments {sorted(unknown)}")
    cfg.update(kw)
    return cfg


class _Configured(UNetDSFamily):
    CBAM_LEVELS = 0

    def __init__(self, hparams=None, **kw):
        c = _settings(hparams, kw)
        super().__init__(c["n_channels"], c["n_classes"], c["kernels_per_layer"], c["bilinear"], c["reduction_ratio"],
                         cbam_levels=self.CBAM_LEVELS)


class UNetDS(_Configured):
    """reference: models/unet_precip_regression_lightning.py:86-118"""
    CBAM_LEVELS = 0


class UNetDSAttention(_Configured):
    """reference: models/unet_precip_regression_lightning.py:121-164 (the SmaAt-UNet wiring)"""
    CBAM_LEVELS = 5


class UNetDSAttention4CBAMs(_Configured):
    """reference: models/unet_precip_regression_lightning.py:167-208"""
    CBAM_LEVELS = 4

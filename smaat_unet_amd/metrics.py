"""On-device PrecipitationMetrics.

Drop-in for /root/reference/metric/precipitation_metrics.py: same constructor arguments, `update(preds, target)`,
`compute()` (same ten keys) and `reset()`.  The reference is a `torchmetrics.Metric` whose `update` runs about a
dozen torch ops and one host synchronisation per training step (models/regression_lightning.py:75,86,94); here an
update is one streaming HIP pass + a one-block finish that accumulates into persistent device state
(`smaat_precip_metrics_update`, include/smaat_hip.h) with NO host synchronisation.  Differences, by design:
  * a batch containing a NaN is skipped on the device and counted; the reference's warning is printed by
    `compute()` (when the state is read back) instead of by `update()`;
  * the sums are accumulated in float64 (the reference adds float32 scalars);
  * `torchmetrics` is not a dependency: cross-process reduction follows its sync/unsync protocol -- when
    `torch.distributed` is initialised, `compute()` all-reduces a COPY of the 9 state values
    (dist_reduce_fx="sum" on every state) and leaves the local accumulators untouched, so repeated compute() /
    update() cycles never double count; every rank takes part in the collective, also one that never called
    `update()`.  `dist_sync_on_step=True` (per-step synchronisation of the forward value) is rejected: the
    reference never sets it and this class has no per-step forward value.
"""
from __future__ import annotations

import torch

from . import _lib

_F64 = ("total_loss", "total_loss_denorm")
_I64 = ("nan_batches", "total_tn", "total_fp", "total_fn", "total_tp", "total_samples", "total_pixels")


class PrecipitationMetrics:
    def __init__(self, threshold=0.5, denormalize=True, dist_sync_on_step=False, process_group=None, device=None):
        if dist_sync_on_step:
            raise NotImplementedError("PrecipitationMetrics: dist_sync_on_step=True is not supported (see module docstring)")
        self.threshold = threshold
        self.denormalize = denormalize
        self.dist_sync_on_step = False
        self.process_group = process_group
        self._device = torch.device(device) if device is not None else None
        self.factor = 47.83  # reference :23
        self._f64 = None
        self._i64 = None
        self._ws = None

    # -- state ---------------------------------------------------------------------------------------
    def _ensure(self, device, n):
        L = _lib.get()
        if self._f64 is None or self._f64.device != device:
            self._f64 = torch.zeros(len(_F64), dtype=torch.float64, device=device)
            self._i64 = torch.zeros(len(_I64), dtype=torch.int64, device=device)
            self._ws = None
        need = int(L.smaat_precip_metrics_ws_bytes(n))
        if self._ws is None or self._ws.numel() * 8 < need:
            self._ws = torch.empty((need + 7) // 8, dtype=torch.float64, device=device)
        return L

    def reset(self):
        if self._f64 is not None:
            self._f64.zero_()
            self._i64.zero_()

    def _local_state(self, device=None):
        """the two state tensors, allocated (as zeros) when this rank has not seen an update yet"""
        if self._f64 is None:
            dev = device or self._device or (torch.device("cuda", torch.cuda.current_device())
                                             if torch.cuda.is_available() else torch.device("cpu"))
            self._f64 = torch.zeros(len(_F64), dtype=torch.float64, device=dev)
            self._i64 = torch.zeros(len(_I64), dtype=torch.int64, device=dev)
        return self._f64, self._i64

    def state(self, sync=True):
        """name -> python number (synchronises the host).  With torch.distributed initialised and sync=True the values
        are the totals over the ranks of `process_group`: the collective runs on COPIES (torchmetrics' sync/unsync), the
        local accumulators keep counting from where they were."""
        import torch.distributed as dist
        f, i = self._local_state()
        if sync and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1:
            f, i = f.clone(), i.clone()
            dist.all_reduce(f, group=self.process_group)
            dist.all_reduce(i, group=self.process_group)
        return {**dict(zip(_F64, f.tolist())), **dict(zip(_I64, i.tolist()))}

    def sync(self, group=None):
        """explicitly REPLACE the local state by the totals over the ranks (every rank must call it; a rank without
        updates contributes zeros).  Prefer compute(), which reduces copies; calling sync() twice double counts."""
        import torch.distributed as dist
        f, i = self._local_state()
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(f, group=group if group is not None else self.process_group)
            dist.all_reduce(i, group=group if group is not None else self.process_group)

    # -- reference API -------------------------------------------------------------------------------
    def update(self, preds, target):
        if preds.shape != target.shape:  # reference :51-58
            if preds.dim() < target.dim():
                preds = preds.unsqueeze(0)
            elif preds.dim() > target.dim():
                preds = preds.squeeze()
                if preds.dim() < target.dim():
                    preds = preds.unsqueeze(0)
        if preds.numel() != target.numel():
            raise ValueError(f"preds {tuple(preds.shape)} and target {tuple(target.shape)} do not match")
        if preds.dtype != torch.float32 or target.dtype != torch.float32:
            raise TypeError("PrecipitationMetrics.update expects float32 tensors")
        if not _lib._ALLOW_HOST_POINTERS and not (preds.is_cuda and target.is_cuda):
            raise _lib.SmaatHipError("smaat_unet_amd has no CPU fallback: tensors must live on the GPU")
        p, t = preds.detach().contiguous(), target.detach().contiguous()
        n = t.numel()
        L = self._ensure(t.device, n)
        stream = torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0
        _lib.check(L.smaat_precip_metrics_update(p.data_ptr(), t.data_ptr(), n, int(t.size(0)), float(self.factor),
                                                 float(self.threshold), int(bool(self.denormalize)),
                                                 self._ws.data_ptr(), self._f64.data_ptr(), self._i64.data_ptr(),
                                                 stream), "smaat_precip_metrics_update")

    def compute(self, sync=True):
        s = self.state(sync=sync)
        if s["nan_batches"]:
            print(f"Warning: NaN values detected in predictions or targets ({s['nan_batches']} batch(es) skipped)")
        nan = float("nan")
        tp, fp, tn, fn = s["total_tp"], s["total_fp"], s["total_tn"], s["total_fn"]
        ns, npx = s["total_samples"], s["total_pixels"]
        mse = s["total_loss"] / ns if ns else nan
        mse_denorm = s["total_loss_denorm"] / ns if (self.denormalize and ns) else nan
        mse_pixel = s["total_loss_denorm"] / npx if (self.denormalize and npx) else nan
        precision = tp / (tp + fp) if (tp + fp) > 0 else nan
        recall = tp / (tp + fn) if (tp + fn) > 0 else nan
        total = tp + tn + fp + fn
        accuracy = (tp + tn) / total if total > 0 else nan
        f1 = 2 * precision * recall / (precision + recall) if (precision + recall) > 0 else nan
        csi = tp / (tp + fn + fp) if (tp + fn + fp) > 0 else nan
        far = fp / (tp + fp) if (tp + fp) > 0 else nan
        denom = (tp + fn) * (fn + tn) + (tp + fp) * (fp + tn)
        hss = ((tp * tn) - (fn * fp)) / denom if denom > 0 else nan
        return {"mse": mse, "mse_denorm": mse_denorm, "mse_pixel": mse_pixel, "precision": precision,
                "recall": recall, "accuracy": accuracy, "f1": f1, "csi": csi, "far": far, "hss": hss}

"""Adam in ONE launch per step (round 6).

reference: `optim.Adam(self.parameters(), lr=self.hparams.learning_rate)` -- /root/reference/models/regression_lightning.py:48
and /root/reference/train_SmaAtUNet.py:182 (default betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad).

`Adam(params, lr)` is a `torch.optim.Optimizer` with torch.optim.Adam's constructor for the arguments the reference uses, its
state layout (`state[p] = {"step", "exp_avg", "exp_avg_sq"}`: `state_dict()` round-trips) and its arithmetic, operation for
operation in f32 (include/smaat_hip.h "Adam in one launch").  What differs is the launch count: torch's multi-tensor path runs
~20 foreach kernels per step over the 145 parameter tensors of the network and passes three to five times over every state
tensor; `smaat_adam_step` reads p, g, m, v once and writes p, m, v once, in one launch (two when a group holds more than
`smaat_adam_max_tensors()` tensors).  The moments of a group live in two flat buffers (the state entries are views).

No CPU fallback: parameters, gradients and state are f32 CUDA tensors (the CPU test suite runs the host logic against the
numpy twin of the library, tests/emu_backend.py).  Not capturable into a hipGraph (like torch.optim.Adam without
`capturable=True`): the step count lives on the host and the gradient pointers of the step are kernel arguments."""
from __future__ import annotations

import ctypes

import torch

from . import _lib

# which of the three update expressions are evaluated as one fma (smaat_adam_step's `variant` bits).  torch's own three
# implementations of this update (for-loop, foreach, fused) are compiled with contraction on and agree with each other to one
# rounding, not bit for bit; all-contracted is the variant closest to foreach (scripts/probes/adam_variant_probe.py,
# profiles/r6/adam_variant_probe_r6s.txt: 52 of 145 tensors bit-equal after 12 steps, max relative distance 7e-8)
TORCH_CONTRACTION_VARIANT = 7


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, maximize=False,
                 variant=None):
        if weight_decay != 0 or amsgrad or maximize:
            raise NotImplementedError("smaat_unet_amd.optim.Adam: weight_decay / amsgrad / maximize are not built "
                                      "(the reference uses none of them); use torch.optim.Adam")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"invalid hyper-parameters lr={lr} eps={eps} betas={betas}")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        self._variant = TORCH_CONTRACTION_VARIANT if variant is None else int(variant)
        self._plans = {}  # group index -> plan (tables for the tensors of the group that have gradients)

    # ------------------------------------------------------------------------------------------------------------------
    def _plan(self, gi, plist):
        """device tables for this list of parameters (rebuilt when the list or a parameter's storage changes)"""
        key = tuple((id(p), p.data_ptr()) for p in plist)
        plan = self._plans.get(gi)
        if plan is not None and plan["key"] == key:
            return plan
        L = _lib.get()
        dev = plist[0].device
        epb, nmax = L.smaat_adam_block_elems(), L.smaat_adam_max_tensors()
        for p in plist:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise TypeError("smaat_unet_amd.optim.Adam: parameters of a group must be contiguous float32 tensors on one device")
        if not plist[0].is_cuda and not _lib._ALLOW_HOST_POINTERS:
            raise TypeError("smaat_unet_amd.optim.Adam updates GPU tensors (there is no CPU path); use torch.optim.Adam on the CPU")
        # moments: flat buffers, one slice per parameter (16-byte aligned slices: float4 accesses), adopted from existing state
        offs, total = [], 0
        for p in plist:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        m = torch.zeros(total, dtype=torch.float32, device=dev)
        v = torch.zeros(total, dtype=torch.float32, device=dev)
        step = None
        for p, o in zip(plist, offs):
            st = self.state.get(p)
            s = 0.0
            if st:  # (load_state_dict, or a plan rebuilt after the set of parameters with gradients changed)
                m[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
                v[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                s = float(st["step"])
            if step is not None and s != step:
                # torch counts steps per parameter (a parameter that gets its first gradient late starts its bias correction at
                # t = 1); one launch applies ONE pair of bias corrections
                raise NotImplementedError("smaat_unet_amd.optim.Adam: the parameters of a group must share one step count "
                                          "(a parameter received its first gradient after the others had been stepped?)")
            step = s
        step_t = torch.tensor(0.0 if step is None else step, dtype=torch.float32)  # ONE host tensor shared by the group's entries
        for p, o in zip(plist, offs):
            self.state[p] = dict(step=step_t, exp_avg=m[o:o + p.numel()].view_as(p), exp_avg_sq=v[o:o + p.numel()].view_as(p))
        chunks = []
        for c0 in range(0, len(plist), nmax):
            ps, os_ = plist[c0:c0 + nmax], offs[c0:c0 + nmax]
            rows, blk2t, blk0, nb = [], [], [], 0
            for t, (p, o) in enumerate(zip(ps, os_)):
                rows += [p.data_ptr(), m.data_ptr() + 4 * o, v.data_ptr() + 4 * o, p.numel()]
                k = (p.numel() + epb - 1) // epb
                blk0.append(nb)
                blk2t += [t] * k
                nb += k
            chunks.append(dict(params=ps, n=len(ps), blocks=nb,
                               rows=torch.tensor(rows, dtype=torch.int64).to(dev),
                               blk2t=torch.tensor(blk2t, dtype=torch.int32).to(dev),
                               blk0=torch.tensor(blk0, dtype=torch.int32).to(dev),
                               gptr=(ctypes.c_void_p * len(ps))()))
        plan = dict(key=key, m=m, v=v, step=step_t, chunks=chunks)
        self._plans[gi] = plan
        return plan

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.get()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            plan = self._plan(gi, plist)
            plan["step"] += 1  # (host tensor shared by every state entry of the group)
            t = float(plan["step"])
            beta1, beta2 = group["betas"]
            lr, eps = group["lr"], group["eps"]
            # torch/optim/adam.py (_multi_tensor_adam, not capturable): the same expressions on Python floats
            bias_correction1 = 1 - beta1 ** t
            bias_correction2 = 1 - beta2 ** t
            step_size = (lr / bias_correction1) * -1
            bias_correction2_sqrt = bias_correction2 ** 0.5
            stream = torch.cuda.current_stream(plist[0].device).cuda_stream if plist[0].is_cuda else None
            for ch in plan["chunks"]:
                for i, p in enumerate(ch["params"]):
                    g = p.grad
                    if g.dtype != torch.float32 or g.is_sparse or not g.is_contiguous() or g.device != p.device:
                        raise TypeError("smaat_unet_amd.optim.Adam: gradients must be dense contiguous float32 tensors on the "
                                        "parameter's device")
                    ch["gptr"][i] = g.data_ptr()
                _lib.check(L.smaat_adam_step(ch["rows"].data_ptr(), ch["gptr"], ch["blk2t"].data_ptr(), ch["blk0"].data_ptr(),
                                             ch["n"], ch["blocks"], 1 - beta1, beta2, 1 - beta2, bias_correction2_sqrt, eps, step_size,
                                             self._variant, stream), "smaat_adam_step")
                # the kernel wrote the parameters (and the moments) through raw pointers: tell autograd -- and everything that
                # watches version counters, the weight-image cache of ops.py first of all -- that they were modified in place
                torch.autograd.graph.increment_version(ch["params"])
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans.clear()  # (the loaded moments are adopted into fresh flat buffers at the next step)

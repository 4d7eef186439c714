"""Round 6: which contraction variant of smaat_adam_step reproduces torch.optim.Adam (foreach, and fused) bit for bit on this
build of torch, on the 145 parameter shapes of SmaAt_UNet(12, 1), 12 steps of random gradients over nine decades; and what one
optimizer step costs (torch foreach / torch fused / one launch).  python scripts/probes/adam_variant_probe.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from smaat_unet_amd.SmaAt_UNet import SmaAt_UNet  # noqa: E402
from smaat_unet_amd.optim import Adam  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    shapes = [tuple(p.shape) for p in SmaAt_UNet(12, 1).parameters()]
    P0 = [torch.randn(s, device=dev) for s in shapes]
    rng = np.random.default_rng(1)
    grads = [[torch.randn(s, device=dev) * float(10.0 ** rng.uniform(-7, 2)) for s in shapes] for _ in range(12)]

    def run(make):
        ps = [torch.nn.Parameter(p.clone()) for p in P0]
        opt = make(ps)
        for gs in grads:
            for p, g in zip(ps, gs):
                p.grad = g.clone()
            opt.step()
        torch.cuda.synchronize()
        return ps, opt

    ref, _ = run(lambda ps: torch.optim.Adam(ps, lr=1e-3, foreach=True))
    fus, _ = run(lambda ps: torch.optim.Adam(ps, lr=1e-3, fused=True))
    one, _ = run(lambda ps: torch.optim.Adam(ps, lr=1e-3, foreach=False))
    print("torch fused vs foreach: bit-equal tensors", sum(torch.equal(a, b) for a, b in zip(fus, ref)), "of", len(ref))
    print("torch for-loop vs foreach: bit-equal tensors", sum(torch.equal(a, b) for a, b in zip(one, ref)), "of", len(ref))
    for variant in range(8):
        ours, opt = run(lambda ps: Adam(ps, lr=1e-3, variant=variant))
        eq = sum(torch.equal(a, b) for a, b in zip(ours, ref))
        err = max(float((a.detach() - b.detach()).abs().max() / b.detach().abs().max()) for a, b in zip(ours, ref))
        st = opt.state[ours[0]]
        print(f"variant {variant}: bit-equal to torch foreach Adam on {eq} of {len(ref)} tensors, max rel diff {err:.2e}, "
              f"step {float(st['step'])}")

    # time of one optimizer step (gradients resident, 200 steps)
    def timeit(make, n=200):
        ps = [torch.nn.Parameter(p.clone()) for p in P0]
        for p, g in zip(ps, grads[0]):
            p.grad = g.clone()
        opt = make(ps)
        for _ in range(5):
            opt.step()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(n):
            opt.step()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    print(f"one step, us: torch foreach {timeit(lambda ps: torch.optim.Adam(ps, lr=1e-3, foreach=True)):.1f}   "
          f"torch fused {timeit(lambda ps: torch.optim.Adam(ps, lr=1e-3, fused=True)):.1f}   "
          f"smaat_unet_amd.optim.Adam {timeit(lambda ps: Adam(ps, lr=1e-3)):.1f}  (host-bound figures: the loop only launches)")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Training-throughput bench for the MI355X-native SmaAt-UNet path.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = forward + MSE(sum)/N loss (reference models/regression_lightning.py:57-65) +
backward + gradient all-reduce (N > 1) + Adam(lr 1e-3) (reference :48) on a batch of 32
synthetic 12x288x288 frames per GPU (BASELINE.json configs[1]; weak scaling -> configs[2] at
N = 8).  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-in MFMA = f32 vector peak
PEAK_HBM_GBS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA


def synthetic_batch(n, h, w, seed, device):
    """~30 % 'rain' pixels in [0, 0.5], target in [0, 0.3]."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, 12, h, w, generator=g)
    x = torch.where(u > 0.7, (u - 0.7) / 0.3 * 0.5, torch.zeros(()))
    y = torch.rand(n, h, w, generator=g) * 0.3
    return x.to(device), y.to(device)


def cpu_baseline(seconds_budget=20.0):
    """The reference's arithmetic on the host cores: oracle/torch_ref.py (the same ATen CPU
    operators the reference dispatches to), batch 2 at 288x288, full train step."""
    from oracle import params as oparams
    from oracle import torch_ref
    threads = torch.get_num_threads()
    P = torch_ref.params_from_numpy(oparams.make_smaat_params(12, 1, 2, 16, 0))
    g = torch.Generator().manual_seed(1)
    bs = 2
    x = torch.rand(bs, 12, 288, 288, generator=g)
    y = torch.rand(bs, 288, 288, generator=g) * 0.3
    opt = torch.optim.Adam([p for p in P.values() if p.requires_grad], lr=1e-3)
    torch_ref.train_step(P, x, y)  # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        torch_ref.train_step(P, x, y)
        opt.step()
        n += 1
        if time.perf_counter() - t0 > seconds_budget or n >= 16:
            break
    dt = time.perf_counter() - t0
    return {"value": round(bs * n / dt, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n} train steps of batch {bs} (12x288x288, fp32) through oracle/torch_ref.py "
                      f"(torch {torch.__version__} CPU ops, {threads} threads)"}


def fwd_latency(model, size, dev, iters=50):
    """eval-mode, no_grad, batch-1 forward (reference call stack D: notebook / calc_metrics_test_set.py
    inference); median of `iters` HIP-event timings, eager launches and one captured HIP graph."""
    model.eval()
    x1, _ = synthetic_batch(1, size, size, 99, dev)
    res = {"batch": 1, "mode": "eval/no_grad", "iters": iters}
    with torch.no_grad():
        for _ in range(3):
            model(x1)
        torch.cuda.synchronize()

        def timed(fn):
            ts = []
            for _ in range(iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            return round(ts[len(ts) // 2], 4)

        res["eager_ms"] = timed(lambda: model(x1))
        try:  # the whole forward as ONE hipGraph launch (launch-bound at batch 1)
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                model(x1)
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(g):
                y_static = model(x1)
            g.replay()
            torch.cuda.synchronize()
            res["hipgraph_ms"] = timed(g.replay)
            res["graph_output_checksum"] = round(float(y_static.double().abs().sum().item()), 4)
        except Exception as e:  # noqa: BLE001
            res["hipgraph_error"] = str(e)[:160]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU")
    ap.add_argument("--size", type=int, default=288)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the f32-MFMA-only reference run")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 eval forward latency")
    ap.add_argument("--precision", choices=["f32", "bf16"], default="f32",
                    help="bf16 = mixed precision (BASELINE configs[3]): bf16 GEMM operands, f32 storage/accumulation")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # SMAAT_BENCH_BACKEND=gloo (testing only): lets several ranks share one GPU to exercise the multi-rank control
    # flow on a single-GPU box; the driver's runs use nccl (= RCCL), one rank per GPU
    backend = os.environ.get("SMAAT_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import smaat_unet_amd as S
    from smaat_unet_amd import _lib
    from smaat_unet_amd.ddp import FlatGradAllReduce
    _lib.get()
    if args.precision == "bf16":
        from smaat_unet_amd import ops as K
        K.set_matrix_mode("bf16")

    torch.manual_seed(0)
    model = S.SmaAt_UNet(12, 1).to(dev).train()
    ddp = FlatGradAllReduce(model.parameters(), world_size=world)
    ddp.broadcast_parameters()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, foreach=True)
    x, y = synthetic_batch(args.batch, args.size, args.size, 1234 + rank, dev)

    def step(exchange=True):
        out = model(x)
        loss = torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / y.size(0)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if world > 1 and exchange:
            ddp.reduce()
        opt.step()
        return loss

    for _ in range(args.warmup):
        loss = step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    final_loss = loss.item()

    roof = None
    kernels = None
    if rank == 0 and not args.no_profile:
        prof = _lib.Profiler()
        for _ in range(2):
            step(exchange=False)  # rank 0 only: per-kernel timing of one replica, no collective in here
        summ = prof.summary()
        prof.close()
        kernels = {}
        for name, d in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
            ms = d["ms"] / 2.0
            e = {"calls": d["calls"] // 2, "ms_per_step": round(ms, 3)}
            if d["flop"]:
                e["tflops"] = round(d["flop"] / 2.0 / (ms * 1e-3) / 1e12, 2)
                e["alg_gbs"] = round(d["bytes"] / 2.0 / (ms * 1e-3) / 1e9, 1)
            kernels[name] = e
        split = bool(_lib.get().smaat_split_enabled())

        def pmc_traffic(kernel_prefixes):
            """HBM bytes per launch of a kernel class from the committed rocprofv3 counter passes
            (profiles/hbm_traffic.json, scripts/make_traffic_json.py): (2 x FETCH_SIZE + WRITE_SIZE) / dispatches
            over the kernels whose name starts with one of the prefixes; None when there is no such record."""
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))["kernels"]
                tot, nd = 0.0, 0
                for kname, v in rec.items():
                    if any(kname.startswith(pfx) for pfx in kernel_prefixes):
                        tot += float(v["fetch_bytes"]) + float(v["write_bytes"])
                        nd += int(v["dispatches"])
                return round(tot / nd) if nd else None
            except Exception:  # noqa: BLE001  (a missing or malformed record must not break the bench line)
                return None

        def klass(names, bound, peak, unit, mult=1.0, what="", pmc=()):
            names = [k for k in names if k in summ]
            if not names:
                return None
            ms = sum(summ[k]["ms"] for k in names) / 2.0
            calls = sum(summ[k]["calls"] for k in names) // 2
            if bound == "mfma":
                alg = sum(summ[k]["flop"] for k in names) / 2.0 / (ms * 1e-3) / 1e12
            else:
                alg = sum(summ[k]["bytes"] for k in names) / 2.0 / (ms * 1e-3) / 1e9
            ach = alg * mult
            return {"bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
                    "traffic": pmc_traffic(pmc) if pmc else None, "kernel": what, "entry_points": names,
                    "launches_per_step": calls,
                    "avg_launch_ms": round(ms / max(calls, 1), 4), "ms_per_step": round(ms, 3),
                    "algorithmic": round(alg, 2)}

        # bf16-split MFMA kernels execute SIX bf16 MFMAs per f32 product: `achieved` = executed bf16 TFLOP/s
        # (= 6 x the algorithmic f32 rate in `algorithmic`), priced against the dense bf16 peak.
        classes = [
            klass(["smaat_pointwise_fwd_split"], "mfma", PEAK_BF16_MFMA_TFLOPS, "TFLOP/s", 6.0,
                  "k_pw_split_p: persistent wave-specialised GEMM (v_mfma_f32_32x32x16_bf16 x6 per f32 product, exact 3-term operand split)",
                  pmc=("k_pw_split",)),
            klass(["smaat_pointwise_wgrad"], "mfma", PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS, "TFLOP/s",
                  6.0 if split else 1.0, "k_wgrad_split (bf16 x6)" if split else "k_wgrad2 (v_mfma_f32_32x32x2_f32)",
                  pmc=("k_wgrad_split",) if split else ("k_wgrad2",)),
            klass(["smaat_dsconv_fwd", "smaat_pointwise_fwd"], "mfma", PEAK_F32_MFMA_TFLOPS, "TFLOP/s", 1.0,
                  "k_pwgemm_ws / k_dsconv_strip / k_pwgemm (v_mfma_f32_32x32x2_f32): fused depthwise->pointwise "
                  "forward and data gradient of the plane-dominated layers", pmc=("k_pwgemm", "k_dsconv_strip")),
            klass(["smaat_dw3x3_bwd"], "hbm", PEAK_HBM_GBS, "GB/s", 1.0, "k_dw3x3_bwd_strip", pmc=("k_dw3x3_bwd",)),
            klass(["smaat_dw3x3_fwd"], "hbm", PEAK_HBM_GBS, "GB/s", 1.0, "k_dw3x3_fwd_strip", pmc=("k_dw3x3_fwd",)),
            klass(["smaat_bn_bwd_apply", "smaat_bn_bwd_reduce", "smaat_affine_act"], "hbm", PEAK_HBM_GBS, "GB/s", 1.0,
                  "BatchNorm/ReLU streaming kernels", pmc=("k_bn_bwd_apply", "k_bn_bwd_reduce", "k_affine_act")),
        ]
        classes = [c for c in classes if c]
        classes.sort(key=lambda c: -c["ms_per_step"])
        roof = dict(classes[0])                    # the dominant kernel class of the step
        roof["other_classes"] = classes[1:]
        roof["matrix_path"] = ("f32 operands split exactly into 3 bf16 terms, 6 bf16 MFMAs per product, f32 accumulate "
                               "(f32-class error, tests/ + profiles/); SMAAT_SPLIT=0 selects the f32-MFMA kernels only"
                               if split else "f32 MFMA (v_mfma_f32_32x32x2_f32) only")

    # ---- per-frame forward latency (second half of BASELINE.json's metric): eval mode, batch 1 ----
    latency = None
    if rank == 0 and not args.no_latency:
        latency = fwd_latency(model, args.size, dev)
        model.train()

    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    alt = None
    if (rank == 0 and args.gpus == 1 and not args.no_alt and os.environ.get("SMAAT_SPLIT", "") != "0"
            and args.precision == "f32"):
        # same step with the f32-MFMA kernels only (no bf16 operand splitting), for reference
        import subprocess
        env = dict(os.environ, SMAAT_SPLIT="0")
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", str(min(args.steps, 5)), "--warmup",
                                "2", "--batch", str(args.batch), "--size", str(args.size), "--no-cpu-baseline",
                                "--no-profile", "--no-alt", "--no-latency"], env=env, capture_output=True, text=True, timeout=600)
            j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            alt = {"matrix_path": "f32 MFMA only (SMAAT_SPLIT=0)", "value": j["value"], "unit": j["unit"],
                   "ms_per_step": j["ms_per_step"]}
        except Exception as e:  # noqa: BLE001
            alt = {"error": str(e)[:200]}

    if rank == 0:
        frames = args.batch * world * args.steps
        line = {
            "metric": "training frames/sec (288x288, 12-ch in)",
            "value": round(frames / dt, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" else "bf16 (GEMM operands; f32 storage and accumulation)",
            "data": "synthetic",
            "config": {"workload": f"SmaAt-UNet 12->1ch, {args.size}x{args.size} synthetic precip, batch={args.batch}/GPU "
                                   "fp32, fwd+MSE+bwd+Adam (BASELINE.json configs[1])",
                       "arithmetic": "f32 storage and accumulation; pointwise GEMMs of the deep layers on the bf16 matrix "
                                     "pipe via exact 3-term operand splitting (f32-class error)",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "final_loss": round(final_loss, 5)},
            "roofline": roof,
            "cpu_baseline": cpu,
            "f32_mfma_only": alt,
            "fwd_latency": latency,
            "kernels": kernels,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()  # the other ranks wait for rank 0's per-kernel / latency passes before tearing down
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

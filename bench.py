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
SUSTAINED_BF16_MFMA_TFLOPS = 1960.0  # measured: scripts/probes/mfma_power_probe.hip, random operand bits, seconds-long run


def synthetic_batch(n, h, w, seed, device):
    """~30 % 'rain' pixels in [0, 0.5], target in [0, 0.3]."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, 12, h, w, generator=g)
    x = torch.where(u > 0.7, (u - 0.7) / 0.3 * 0.5, torch.zeros(()))
    y = torch.rand(n, h, w, generator=g) * 0.3
    return x.to(device), y.to(device)


def csrc_sha16():
    """content hash of the kernel sources + C ABI header: stamps profiles/hbm_traffic.json so that counter data
    of an older build is never quoted for the current one"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "smaat_unet_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(seconds_budget=25.0):
    """The reference's arithmetic on the host cores: oracle/torch_ref.py (the same ATen CPU operators the reference
    dispatches to, pinned to reference-generated fixtures at this size by tests/test_eval_and_big.py), full train
    step (fwd + MSE(sum)/N + bwd + Adam) on SURVEY 8(d) sparse synthetic frames at 288x288.  The thread count is
    swept (the reference sets none: torch's default = all cores is NOT the fastest on a many-core host) with one
    batch-4 step each, then the best count is timed at batch 8."""
    from oracle import params as oparams
    from oracle import torch_ref
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    P = torch_ref.params_from_numpy(oparams.make_smaat_params(12, 1, 2, 16, 0))
    opt = torch.optim.Adam([p for p in P.values() if p.requires_grad], lr=1e-3)
    x8, y8 = synthetic_batch(8, 288, 288, 1, "cpu")

    def one(bs):
        t0 = time.perf_counter()
        torch_ref.train_step(P, x8[:bs], y8[:bs])
        opt.step()
        return time.perf_counter() - t0

    t_start = time.perf_counter()
    cands = sorted({t for t in (8, 16, 32, 64, ncpu) if t <= ncpu})
    sweep = {}
    for t in cands:
        torch.set_num_threads(t)
        one(2)  # warm-up (thread pool, MKLDNN primitive cache)
        sweep[t] = round(4 / one(4), 3)
        if time.perf_counter() - t_start > 0.6 * seconds_budget:
            break
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    n, t0 = 0, time.perf_counter()
    while True:
        one(8)
        n += 1
        if time.perf_counter() - t_start > seconds_budget or n >= 3:
            break
    dt = time.perf_counter() - t0
    torch.set_num_threads(default_threads)
    return {"value": round(8 * n / dt, 4), "unit": "frames/s", "cores": best, "kind": "port",
            "host_cores": ncpu, "thread_sweep_frames_per_s": sweep,
            "sample": f"{n} full train steps (fwd + MSE + bwd + Adam) of batch 8, 12x288x288 fp32, SURVEY 8(d) sparse "
                      f"synthetic frames, through oracle/torch_ref.py (torch {torch.__version__} ATen CPU ops) on "
                      f"{best} threads = the best of a {sorted(sweep)}-thread sweep on a {ncpu}-core host"}


def rocm_eager_baseline(batch, size, dev, steps=5, warmup=2):
    """Secondary baseline: the reference's op graph under STOCK
    PyTorch-ROCm eager (MIOpen / rocBLAS kernels) on the same MI355X, same synthetic batch, same step
    (fwd + MSE(sum)/N + bwd + Adam).  oracle/torch_ref.py is that graph; it is only ever used as a checker/baseline."""
    from oracle import params as oparams
    from oracle import torch_ref
    Pn = oparams.make_smaat_params(12, 1, 2, 16, 0)
    P = {}
    for k, v in Pn.items():
        t = torch.from_numpy(np.asarray(v).copy()).to(dev)
        if t.dtype == torch.float32 and "running" not in k:
            t.requires_grad_(True)
        P[k] = t
    opt = torch.optim.Adam([p for p in P.values() if p.requires_grad], lr=1e-3, foreach=True)
    x, y = synthetic_batch(batch, size, size, 1234, dev)

    def step():
        torch_ref.train_step(P, x, y)
        opt.step()
    try:
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"value": round(batch * steps / dt, 2), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3),
                "what": f"reference op graph (oracle/torch_ref.py) under stock PyTorch-ROCm {torch.__version__} eager "
                        f"(MIOpen/rocBLAS), batch {batch}, {size}x{size}, fp32, fwd+MSE+bwd+Adam, {steps} steps"}
    except Exception as e:  # noqa: BLE001  (e.g. MIOpen find-db missing on the box)
        return {"error": str(e)[:200]}


class PowerSampler:
    """Shader clock and socket power while the timed steps run (rocm-smi from a host thread; nothing is launched on the GPU).
    MI355X is power-managed: under combined MFMA + HBM load the clock sits well below the nominal 2.4 GHz, which the
    cycle-based PMC utilisation figures do not see (profiles/r3/clock_under_load_r3.txt, mfma_power_probe_r3.txt)."""

    def __init__(self, period=0.5):
        import threading
        self.period, self.samples = period, []
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _sample(self):
        import re
        import subprocess
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        except Exception:  # noqa: BLE001  (no rocm-smi on this host)
            return None
        dev = os.environ.get("LOCAL_RANK", "0")
        s = re.search(r"GPU\[%s\]\s*:\s*sclk clock level: \d+: \((\d+)Mhz\)" % dev, out) or re.search(r"sclk[^\n]*?\((\d+)Mhz\)", out)
        w = re.search(r"GPU\[%s\]\s*:[^\n]*Power \(W\): ([\d.]+)" % dev, out) or re.search(r"Power \(W\): ([\d.]+)", out)
        return (int(s.group(1)) if s else None, float(w.group(1)) if w else None)

    def _run(self):
        while not self._stop.is_set():
            r = self._sample()
            if r is not None:
                self.samples.append(r)
            self._stop.wait(self.period)

    def start(self):
        self._th.start()
        return self

    def stop(self):
        self._stop.set()
        self._th.join(timeout=15)
        clk = [c for c, _ in self.samples if c]
        pw = [w for _, w in self.samples if w]
        if not clk:
            return None
        return {"sclk_mhz_mean": round(sum(clk) / len(clk)), "sclk_mhz_min": min(clk), "sclk_mhz_nominal": 2400,
                "power_w_mean": round(sum(pw) / len(pw)) if pw else None, "power_w_max": round(max(pw)) if pw else None,
                "samples": len(clk),
                "note": "sampled with rocm-smi while the step runs; the GEMM phases run at the 1400 W board limit and "
                        "~1.8 GHz, the streaming phases at ~1.0 kW and 2.4 GHz (profiles/r3/clock_under_load_r3.txt)"}


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
        try:  # the module-owned graph (SmaAt_UNet.enable_eval_graph): input copy + replay + output clone
            model.enable_eval_graph(True)
            model(x1)
            res["module_graph_ms"] = timed(lambda: model(x1))
            res["launches_per_forward"] = "see profiles/: ~40 kernels in one hipGraph"
        except Exception as e:  # noqa: BLE001
            res["module_graph_error"] = str(e)[:160]
        finally:
            model.enable_eval_graph(False)
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


def side_roofline(summ, dtype):
    """dominant entry-point class of a profiled pair of steps -> roofline record (bound, achieved, peak, frac, traffic).
    bf16 storage: every class is HBM-bound; f32: the split GEMMs are priced on the 16-bit matrix pipe
    (executed = 3 fp16 or 6 bf16 MFMAs per product), everything else on HBM."""
    gemm3 = ("smaat_pointwise_fwd_split_h", "smaat_pointwise_fwd_split_k_h", "smaat_pointwise_wgrad_h", "smaat_dsconv_wgrad_split_h",
             "smaat_dsconv_fwd_rows_h")
    gemm6 = ("smaat_pointwise_fwd_split", "smaat_pointwise_wgrad", "smaat_dsconv_fwd_rows", "smaat_dsconv_fwd_rows_amax",
             "smaat_dsconv_wgrad_split", "smaat_dsconv_fwd_split")
    groups = {}
    for name, d in summ.items():
        if not d["flop"] and not d["bytes"]:
            continue
        key = ("split GEMM, 2-term fp16 (3 MFMAs per product)" if name in gemm3 else
               "split GEMM, 3-term bf16 (6 MFMAs per product)" if (name in gemm6 and dtype == "f32") else name)
        g = groups.setdefault(key, dict(ms=0.0, flop=0.0, bytes=0.0, calls=0, names=[]))
        g["ms"] += d["ms"] / 2.0
        g["flop"] += d["flop"] / 2.0
        g["bytes"] += d["bytes"] / 2.0
        g["calls"] += d["calls"] // 2
        g["names"].append(name)
    if not groups:
        return None
    key, g = max(groups.items(), key=lambda kv: kv[1]["ms"])
    tf, gbs = g["flop"] / (g["ms"] * 1e-3) / 1e12, g["bytes"] / (g["ms"] * 1e-3) / 1e9
    mfma = dtype == "f32" and key.startswith("split GEMM")
    rec = {"kernel": key, "entry_points": sorted(g["names"]), "ms_per_step": round(g["ms"], 3), "launches_per_step": g["calls"],
           "timing": "instrumented pass (HIP events around every entry point)",
           "bound": "mfma" if mfma else "hbm", "algorithmic_tflops": round(tf, 2), "algorithmic_gbs": round(gbs, 1),
           "traffic": None, "traffic_note": "counter passes are collected for the headline configuration only "
                                             "(profiles/hbm_traffic.json); bf16 at batch 64: profiles/r*/prof_*_bf16"}
    if mfma:
        mult = 3.0 if "2-term" in key else 6.0
        rec.update(achieved=round(tf, 2), peak=PEAK_BF16_MFMA_TFLOPS, unit="TFLOP/s", executed=round(tf * mult, 2),
                   frac=round(tf * mult / PEAK_BF16_MFMA_TFLOPS, 4), frac_of_f32_mfma_peak=round(tf / PEAK_F32_MFMA_TFLOPS, 4),
                   frac_definition=f"executed 16-bit MFMA TFLOP/s ({mult:g} per f32 product) / 2500")
    else:
        rec.update(achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4))
    return rec


def allreduce_model(nbytes, n=8):
    """with one GPU at hand, the MODELLED cost of the gradient exchange of configs[2] -- one ring all-reduce of
    the flat fp32 gradient buffer over xGMI (point-to-point links, ~153 GB/s per direction per link, MI355X_MICROARCH.md):
    2 (n-1)/n of the buffer crosses each link, plus 2 (n-1) hops of link latency.  A model, not a measurement."""
    link_gbs, hop_us = 153.0, 5.0
    ring = 2.0 * (n - 1) / n * nbytes / (link_gbs * 1e9) * 1e3
    lat = 2 * (n - 1) * hop_us * 1e-3
    return {"modelled": True, "n_gpus": n, "bytes": int(nbytes), "link_gbs_assumed": link_gbs, "hop_latency_us_assumed": hop_us,
            "bandwidth_term_ms": round(ring, 3), "latency_term_ms": round(lat, 3), "allreduce_modelled_ms": round(ring + lat, 3),
            "note": "ring all-reduce over xGMI, ONE collective per step (two buckets in smaat_unet_amd/ddp.py); against the "
                    "measured single-GPU step this predicts the weak-scaling loss if nothing is overlapped"}


def side_config(kind, dev, steps=24, warmup=6):
    """Short measurement of another single-GPU configuration of BASELINE.json inside the default run, so that the driver's
    BENCH record carries it: "bf16_b64" = configs[3] (12->1, 288x288, batch 64, mixed precision = bf16 activation storage),
    "voc_b16" = configs[4] (3->21, 256x256, batch 16, CrossEntropyLoss, f32).  Same step as the headline: forward + loss +
    backward + Adam, inputs resident in HBM, `steps` timed steps between synchronisations."""
    import smaat_unet_amd as S
    torch.manual_seed(0)
    if kind == "bf16_b64":
        model = S.SmaAt_UNet(12, 1).to(dev).train().set_precision("bf16")
        batch, size = 64, 288
        x, y = synthetic_batch(batch, size, size, 1234, dev)
        lossf = lambda out: torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / y.size(0)  # noqa: E731
        what = ("SmaAt-UNet 12->1ch, 288x288 synthetic precip, batch=64, mixed precision: bf16 activation storage + bf16 "
                "MFMA GEMMs, f32 accumulation / BatchNorm statistics / master weights, fwd+MSE+bwd+Adam (BASELINE.json configs[3])")
        dtype = "bf16"
    else:
        model = S.SmaAt_UNet(3, 21).to(dev).train()
        batch, size = 16, 256
        g = torch.Generator().manual_seed(1234)
        x = torch.randn(batch, 3, size, size, generator=g).to(dev)
        y = torch.randint(0, 21, (batch, size, size), generator=g).to(dev)
        lossf = lambda out: torch.nn.functional.cross_entropy(out, y)  # noqa: E731
        what = ("SmaAt-UNet 3->21ch (PascalVOC head), 256x256 synthetic images, batch=16 fp32, fwd+CrossEntropy+bwd+Adam "
                "(BASELINE.json configs[4])")
        dtype = "f32"
    if os.environ.get("SMAAT_ADAM", "one") == "one":  # (as the headline configuration)
        from smaat_unet_amd.optim import Adam as OneLaunchAdam
        opt = OneLaunchAdam(model.parameters(), lr=1e-3)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, foreach=True)

    def step():
        loss = lossf(model(x))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss
    try:
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rec = {"value": round(batch * steps / dt, 2), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3),
               "steps": steps, "warmup": warmup, "batch": batch, "dtype": dtype, "final_loss": round(loss.item(), 5),
               "workload": what}
        # roofline of this configuration's dominant kernel class: HIP events around every entry point
        # for two more steps, ALGORITHMIC bytes / flops of the class (smaat_unet_amd/_lib.py WORK_MODELS = SURVEY 8(d)) over
        # its time, against the roofline that bounds it
        from smaat_unet_amd import _lib
        prof = _lib.Profiler()
        for _ in range(2):
            step()
        summ = prof.summary()
        prof.close()
        rec["roofline"] = side_roofline(summ, dtype)
    except Exception as e:  # noqa: BLE001
        rec = {"error": str(e)[:300]}
    del model, opt, x, y
    torch.cuda.empty_cache()
    return rec


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this script under torch.distributed.run with N ranks on
    this node (rendezvous on 127.0.0.1, a free port), exactly the driver's multi-GPU command line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    print("bench.py: launching", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)  # >= 5 s of timed region
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU (default 32; 16 for --config voc)")
    ap.add_argument("--size", type=int, default=None, help="default 288 (256 for --config voc)")
    ap.add_argument("--config", choices=["precip", "voc"], default="precip",
                    help="precip = BASELINE.json configs[1]/[2]/[3] (12->1, MSE); voc = configs[4] (3->21, 256x256, batch 16, "
                         "CrossEntropyLoss: reference train_SmaAtUNet.py:178-183)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the f32-MFMA-only reference run")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 eval forward latency")
    ap.add_argument("--no-power", action="store_true", help="do not sample clock / power with rocm-smi during the timed steps")
    ap.add_argument("--no-eager-baseline", action="store_true", help="skip the stock PyTorch-ROCm eager baseline")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="skip the short sub-records of BASELINE configs[3] (bf16, batch 64) and configs[4] (VOC head)")
    ap.add_argument("--no-input-pipeline", action="store_true",
                    help="skip the leg that feeds the step from smaat_unet_amd.data.PrefetchLoader (PCIe-inclusive rate)")
    ap.add_argument("--precision", choices=["f32", "bf16", "bf16_operands"], default="f32",
                    help="bf16 = mixed precision (BASELINE configs[3]): bf16 activation storage + bf16 MFMA GEMMs, f32 "
                         "accumulation / statistics / master weights; bf16_operands = the round-2 mode (bf16 GEMM operands, "
                         "f32 storage)")
    args = ap.parse_args()
    voc = args.config == "voc"
    if args.batch is None:
        args.batch = 16 if voc else 32
    if args.size is None:
        args.size = 256 if voc else 288
    if voc:  # the secondary legs are defined for the headline config only
        args.no_cpu_baseline = args.no_alt = args.no_latency = args.no_eager_baseline = args.no_input_pipeline = True
        args.no_side_configs = True
    if args.precision != "f32" or args.batch != 32 or args.size != 288:
        args.no_side_configs = True  # they belong to the default (headline) run

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: launch ourselves as N ranks, one per GPU (the same command line the driver
        # uses: torch.distributed.run on 127.0.0.1), pass the ranks' output through and return their exit code
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but torch.distributed.run started {world} ranks (WORLD_SIZE={world})")
    # SMAAT_BENCH_BACKEND=gloo (testing only): lets several ranks share one GPU to exercise the multi-rank control
    # flow on a single-GPU box; the driver's runs use nccl (= RCCL), one rank per GPU
    backend = os.environ.get("SMAAT_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"bench.py --gpus {world} needs {world} devices (one rank per GPU over RCCL); this host has {ndev}")
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # multi-GPU pre-flight record: which device every rank drives, and the IPC mode RCCL needs on this
    # host driver (dmabuf only: HSA_ENABLE_IPC_MODE_LEGACY must be 0, else hipIpcGetMemHandle fails at the first collective)
    placement = [(rank, local_rank, dev_index)]
    if world > 1:
        if backend == "nccl" and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "") != "0":
            print(f"[bench rank {rank}] WARNING: HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')!r}; "
                  "RCCL over xGMI needs 0 on this driver", file=sys.stderr)
        gathered = [None] * world
        dist.all_gather_object(gathered, placement[0])
        placement = sorted(gathered)
        if backend == "nccl" and len({d for _, _, d in placement}) != world:
            raise SystemExit(f"rank -> device map is not one-to-one: {placement}")

    import smaat_unet_amd as S
    from smaat_unet_amd import _lib
    from smaat_unet_amd.ddp import FlatGradAllReduce
    _lib.get()
    if args.precision == "bf16_operands":
        from smaat_unet_amd import ops as K
        K.set_matrix_mode("bf16")

    torch.manual_seed(0)
    model = (S.SmaAt_UNet(3, 21) if voc else S.SmaAt_UNet(12, 1)).to(dev).train()
    if args.precision == "bf16":
        model.set_precision("bf16")
    ddp = FlatGradAllReduce(model, world_size=world)  # persistent flat gradient buffer, bucketed async all-reduce
    ddp.broadcast_parameters()
    # Adam(lr 1e-3), reference models/regression_lightning.py:48.  "one" (default): smaat_unet_amd.optim.Adam -- torch.optim.Adam's
    # multi-tensor update (same f32 expressions) in ONE launch over the 145 tensors; "foreach" / "fused": stock torch.optim.Adam
    # (SURVEY 8 a14).  Interleaved on one box (profiles/r6/bench_ab_adam_r6t2.txt): one launch 27.92 / 27.90 ms, foreach 28.18 /
    # 28.17, fused 28.40 / 28.56.
    adam_impl = os.environ.get("SMAAT_ADAM", "one")
    if adam_impl == "one":
        from smaat_unet_amd.optim import Adam as OneLaunchAdam
        opt = OneLaunchAdam(model.parameters(), lr=1e-3)
    else:
        opt = (torch.optim.Adam(model.parameters(), lr=1e-3, fused=True) if adam_impl == "fused"
               else torch.optim.Adam(model.parameters(), lr=1e-3, foreach=True))
    if voc:  # ImageNet-normalised-like images, integer class maps
        g = torch.Generator().manual_seed(1234 + rank)
        x = torch.randn(args.batch, 3, args.size, args.size, generator=g).to(dev)
        y = torch.randint(0, 21, (args.batch, args.size, args.size), generator=g).to(dev)
    else:
        x, y = synthetic_batch(args.batch, args.size, args.size, 1234 + rank, dev)

    def step(exchange=True):
        out = model(x)
        if voc:
            loss = torch.nn.functional.cross_entropy(out, y)
        else:
            loss = torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / y.size(0)
        if world > 1:
            ddp.active = exchange
            ddp.zero_grad()
            loss.backward()
            ddp.finish()          # pack into the persistent flat buffer, bucket all-reduces over RCCL, average
        else:
            opt.zero_grad(set_to_none=True)
            loss.backward()
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
    # clock / power: sampled over a SEPARATE pass of the same steps after the timed region, long enough for >= 5 rocm-smi samples
    # whatever --steps was (the driver's 20 steps are 0.6 s: one sample), and without a host thread
    # spawning subprocesses inside the timed region.  Fewer than 5 samples -> null.
    power = None
    if rank == 0 and world == 1 and not args.no_power:
        sampler = PowerSampler(period=0.15).start()
        t_end, nps = time.perf_counter() + 4.0, 0
        while time.perf_counter() < t_end:
            for _ in range(8):
                step()
            torch.cuda.synchronize()
            nps += 8
        power = sampler.stop()
        if power is not None and power["samples"] < 5:
            power = None
        elif power is not None:
            power["sampled_over"] = f"{nps} further identical steps after the timed region (~4 s)"
    peak_gb = round(torch.cuda.max_memory_allocated(dev) / 2**30, 2)  # activations + workspaces + optimizer state of this rank
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

        traffic_rec = {"kernels": {}, "stale": True, "why": "profiles/hbm_traffic.json missing"}
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
            if rec.get("csrc_sha16") == csrc_sha16():
                traffic_rec = {"kernels": rec["kernels"], "stale": False, "source": rec.get("source")}
            else:
                traffic_rec["why"] = (f"profiles/hbm_traffic.json was collected for kernel sources {rec.get('csrc_sha16')}, "
                                      f"this build is {csrc_sha16()}: re-run scripts/prof_round.sh + make_traffic_json.py")
        except Exception as e:  # noqa: BLE001
            traffic_rec["why"] = str(e)[:120]

        def pmc_traffic(kernel_prefixes, ends=None):
            """HBM bytes per launch of a kernel class from the rocprofv3 counter passes OF THIS BUILD
            (profiles/hbm_traffic.json <- scripts/prof_round.sh + scripts/make_traffic_json.py, stamped with the
            hash of the kernel sources): (2 x FETCH_SIZE + WRITE_SIZE) / dispatches over the kernels whose name
            starts with one of the prefixes; None when the record is missing or belongs to another build."""
            if traffic_rec["stale"]:
                return None
            tot, nd = 0.0, 0
            bf_run = args.precision == "bf16"
            for kname, v in traffic_rec["kernels"].items():
                # the record holds the f32 run and the mixed-precision run of one build: typed kernels carry their element
                # types in the name ("float" / "unsigned short" template arguments) -- take the instantiations of THIS run
                is_bf, is_f = "unsigned short" in kname, "float" in kname
                if (is_bf and not bf_run) or (bf_run and is_f and not is_bf):
                    continue
                if any(kname.startswith(pfx) for pfx in kernel_prefixes) and (ends is None or kname.rstrip().endswith(ends)):
                    tot += float(v["fetch_bytes"]) + float(v["write_bytes"])
                    nd += int(v["dispatches"])
            return round(tot / nd) if nd else None

        def klass(names, bound, peak, unit, mult=1.0, what="", pmc=(), peak_name="", pmc_ends=None):
            """`achieved` / `frac` follow SURVEY 8(d): ALGORITHMIC work (2 K Cout HW N flops, or the per-stage
            algorithmic bytes) / measured time / the peak named in `peak_name`.  For the bf16-split GEMMs the
            executed matrix-pipe work is `mult` x that (six bf16 MFMAs per f32 product): reported separately as
            `executed` / `frac_executed`, never mixed into `frac`."""
            names = [k for k in names if k in summ]
            if not names:
                return None
            ms = sum(summ[k]["ms"] for k in names) / 2.0
            calls = sum(summ[k]["calls"] for k in names) // 2
            alg_tf = sum(summ[k]["flop"] for k in names) / 2.0 / (ms * 1e-3) / 1e12
            alg_gb = sum(summ[k]["bytes"] for k in names) / 2.0 / (ms * 1e-3) / 1e9
            alg = alg_tf if bound == "mfma" else alg_gb
            r = {"bound": bound, "achieved": round(alg, 2), "peak": peak, "unit": unit, "frac": round(alg / peak, 4),
                 "timing": "instrumented pass after the timed region (HIP events around every entry point: class times sum to "
                           "~3-4 % more than ms_per_step of the headline)",
                 "peak_name": peak_name, "traffic": pmc_traffic(pmc, pmc_ends) if pmc else None, "kernel": what,
                 "entry_points": names, "launches_per_step": calls, "avg_launch_ms": round(ms / max(calls, 1), 4),
                 "ms_per_step": round(ms, 3), "algorithmic_tflops": round(alg_tf, 2),
                 "algorithmic_gbs": round(alg_gb, 1), "hbm_frac_algorithmic": round(alg_gb / PEAK_HBM_GBS, 4)}
            if bound == "mfma":
                r["frac_of_f32_mfma_peak"] = round(alg_tf / PEAK_F32_MFMA_TFLOPS, 4)
                if mult != 1.0:
                    r["executed"] = round(alg_tf * mult, 2)
                    r["frac_executed"] = round(alg_tf * mult / peak, 4)
                    r["executed_note"] = f"{mult:g} bf16 MFMAs per f32 product: executed bf16 TFLOP/s = {mult:g} x algorithmic"
            if r["traffic"] is not None:
                r["traffic_over_algorithmic_bytes"] = round(r["traffic"] / (alg_gb * 1e9 * ms * 1e-3 / max(calls, 1)), 3)
            return r

        def retarget(c):
            """top-level `frac` of a split-GEMM class = executed / attainable: the matrix pipe's utilisation (VERDICT r2
            weak #10); the algorithmic-flops-over-bf16-peak ratio stays as `frac_algorithmic_vs_bf16_peak`"""
            if c and "frac_executed" in c:
                c["frac_algorithmic_vs_bf16_peak"] = c["frac"]
                c["frac"] = c["frac_executed"]
                m_ = round(c["executed"] / max(c["algorithmic_tflops"], 1e-9))
                c["frac_definition"] = (f"executed 16-bit MFMA TFLOP/s ({m_} per f32 product) / 2500 = matrix-pipe utilisation; "
                                        f"equivalently algorithmic TFLOP/s / the {2500 // max(m_, 1)} TFLOP/s a {m_}-MFMA split can reach")
                # measured on this part (profiles/r3/mfma_power_probe_r3.txt, clock_under_load_r3.txt): bare MFMA chains on
                # operands with random bits sustain 1960 TFLOP/s (1.99 GHz, 1.32 kW); this kernel class runs at the 1400 W
                # board limit with the clock at ~1.8 GHz
                c["sustained_mfma_peak"] = SUSTAINED_BF16_MFMA_TFLOPS
                c["frac_executed_vs_sustained_peak"] = round(c["executed"] / SUSTAINED_BF16_MFMA_TFLOPS, 4)
                c["power_note"] = ("power-limited: 1.39-1.40 kW of the 1.4 kW board limit and ~1.8 GHz under this kernel "
                                   "(nominal 2.4 GHz); dense bf16 MFMA on random operands sustains 1960 TFLOP/s on this chip")
            return c

        BF16 = "dense bf16 MFMA 2500 TFLOP/s (the pipe the kernel runs on)"
        F32 = "f32-input MFMA 157.3 TFLOP/s"
        HBM = "HBM3E 8000 GB/s"
        nt = 6.0 if _lib.get().smaat_split_mode() == 3 else (3.0 if _lib.get().smaat_split_mode() == 2 else 1.0)
        F16 = "dense fp16 MFMA 2500 TFLOP/s (the pipe the kernel runs on)"
        classes = [
            klass(["smaat_pointwise_fwd_split_h", "smaat_pointwise_fwd_split_k_h"], "mfma", PEAK_BF16_MFMA_TFLOPS, "TFLOP/s", 3.0,
                  "k_pw_split_p<NT=2>: persistent wave-specialised GEMM on the TWO-term fp16 operand split (three "
                  "v_mfma_f32_32x32x16_f16 per product, per-tensor power-of-two scales from the producing kernels' maxima): "
                  "pointwise forward of the GEMM-sized layers + every data gradient", pmc=("k_pw_split",), peak_name=F16,
                  pmc_ends=", 2>"),
            klass(["smaat_pointwise_wgrad_h"], "mfma", PEAK_BF16_MFMA_TFLOPS, "TFLOP/s", 3.0,
                  "k_wgrad_split<NT=2>: streamed pointwise weight gradient on the two-term fp16 split", pmc=("k_wgrad_split<2,",),
                  peak_name=F16),
            klass(["smaat_pointwise_fwd_split"], "mfma", PEAK_BF16_MFMA_TFLOPS, "TFLOP/s", nt,
                  "k_pw_split_p: persistent wave-specialised GEMM (v_mfma_f32_32x32x16_bf16, exact 3-term operand "
                  "split): pointwise forward of the GEMM-sized layers + every data gradient", pmc=("k_pw_split",),
                  peak_name=BF16),
            klass(["smaat_dsconv_wgrad_split_h"], "mfma", PEAK_BF16_MFMA_TFLOPS, "TFLOP/s", 3.0,
                  "k_dsconv_wgrad_split<NT=2>: recompute weight gradient of the 288^2 layers on the two-term fp16 split (the "
                  "forward left the maximum of the depthwise output it formed)", pmc=("k_dsconv_wgrad_split<2,",), peak_name=F16),
            klass(["smaat_dsconv_fwd_rows_h"], "mfma", PEAK_BF16_MFMA_TFLOPS, "TFLOP/s", 3.0,
                  "k_dsconv_rows_fwd<NT=2>: fused depthwise 3x3 -> GEMM forward of the 288^2 layers on the two-term fp16 split (row-walking "
                  "register window, register-resident weight fragments; the scale of the depthwise output from an a-priori bound: "
                  "max |x| of the tensor's writers, or through the previous pointwise weight); no depthwise tensor in HBM",
                  pmc=("k_dsconv_rows_fwd<2,",), peak_name=F16),
            klass(["smaat_dsconv_fwd_split", "smaat_dsconv_fwd_rows_amax"] + (["smaat_dsconv_fwd_rows"] if args.precision != "bf16" else []), "mfma",
                  PEAK_BF16_MFMA_TFLOPS, "TFLOP/s", nt,
                  "k_dsconv_rows_fwd: fused depthwise 3x3 -> split GEMM forward of the 288^2 layers (row-walking "
                  "register window + register-resident weight fragments; K = 256: the third weight plane in LDS); no "
                  "depthwise tensor in HBM", pmc=("k_dsconv_rows_fwd", "k_dsconv_split"), peak_name=BF16),
            klass(["smaat_dsconv_wgrad_split"], "mfma", PEAK_BF16_MFMA_TFLOPS, "TFLOP/s", nt,
                  "k_dsconv_wgrad_split: pointwise weight gradient of the 288^2 layers with the depthwise output recomputed from "
                  "x by the producer waves (reads Cin instead of 2 Cin channels; VALU-bound beside the MFMAs)",
                  pmc=("k_dsconv_wgrad_split",), peak_name=BF16),
            klass(["smaat_pointwise_wgrad"], "mfma", PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS, "TFLOP/s",
                  nt if split else 1.0, "k_wgrad_split" if split else "k_wgrad2 (v_mfma_f32_32x32x2_f32)",
                  pmc=("k_wgrad_split",) if split else ("k_wgrad2",), peak_name=BF16 if split else F32),
            klass(["smaat_dsconv_fwd", "smaat_pointwise_fwd"], "mfma", PEAK_F32_MFMA_TFLOPS, "TFLOP/s", 1.0,
                  "k_pwgemm_ws / k_dsconv_strip / k_pwgemm (v_mfma_f32_32x32x2_f32)", pmc=("k_pwgemm", "k_dsconv_strip"),
                  peak_name=F32),
            klass(["smaat_dw3x3_bwd", "smaat_dw3x3_bwd_bnred"], "hbm", PEAK_HBM_GBS, "GB/s", 1.0,
                  "k_dw3x3_bwd_rows (register row-streaming depthwise backward, + the fused BatchNorm reduction)",
                  pmc=("k_dw3x3_bwd",), peak_name=HBM),
            klass(["smaat_dw3x3_fwd", "smaat_dw3x3_fwd_amax"], "hbm", PEAK_HBM_GBS, "GB/s", 1.0, "k_dw3x3_fwd_lin (one output "
                  "position per lane, waves in address order: f32 planes of 72 x 72 and more) / k_dw3x3_fwd_rows (register row-streaming "
                  "walker: the smaller planes); _amax: + the maximum of the output for the fp16 split", pmc=("k_dw3x3_fwd",),
                  peak_name=HBM),
            klass(["smaat_bn_bwd_apply", "smaat_bn_bwd_apply_amax", "smaat_bn_bwd_reduce", "smaat_affine_act"], "hbm", PEAK_HBM_GBS,
                  "GB/s", 1.0,
                  "BatchNorm/ReLU streaming kernels", pmc=("k_bn_bwd_apply", "k_bn_bwd_reduce", "k_affine_act"),
                  peak_name=HBM),
        ]
        classes += [  # mixed precision (bf16 storage): every layer is HBM-bound, all classes priced on HBM
            klass(["smaat_dsconv_fwd_rows"], "hbm", PEAK_HBM_GBS, "GB/s", 1.0,
                  "k_dsconv_rows_fwd<bf16>: row-walking fused depthwise -> bf16 GEMM forward of the 288^2 layers (no depthwise "
                  "tensor in HBM)", pmc=("k_dsconv_rows_fwd",), peak_name=HBM) if args.precision == "bf16" else None,
            klass(["smaat_dsconv_wgrad_split_t"], "hbm", PEAK_HBM_GBS, "GB/s", 1.0,
                  "k_dsconv_wgrad_split<bf16>: weight gradient with the depthwise output recomputed from x",
                  pmc=("k_dsconv_wgrad_split",), peak_name=HBM),
            klass(["smaat_pointwise_fwd_bf16"], "hbm", PEAK_HBM_GBS, "GB/s", 1.0,
                  "k_pw_bf16: bf16 GEMM fed by LDS-DMA + ds_read_b64_tr_b16 (pointwise forward + every data gradient), "
                  "bf16 in / bf16 out, f32 accumulate", pmc=("k_pw_bf16",), peak_name=HBM),
            klass(["smaat_pointwise_wgrad_bf16"], "hbm", PEAK_HBM_GBS, "GB/s", 1.0,
                  "k_wgrad_bf16: bf16 weight gradient fed by LDS-DMA", pmc=("k_wgrad_bf16",), peak_name=HBM),
            klass(["smaat_dw3x3_bwd_t"], "hbm", PEAK_HBM_GBS, "GB/s", 1.0,
                  "k_dw3x3_bwd_rows<bf16>: row-streaming depthwise backward (+ fused BatchNorm reduction)",
                  pmc=("k_dw3x3_bwd",), peak_name=HBM),
            klass(["smaat_dw3x3_fwd_t"], "hbm", PEAK_HBM_GBS, "GB/s", 1.0, "k_dw3x3_fwd_rows<bf16>", pmc=("k_dw3x3_fwd",),
                  peak_name=HBM),
            klass(["smaat_bn_bwd_apply_t", "smaat_bn_bwd_reduce_t", "smaat_affine_act_t"], "hbm", PEAK_HBM_GBS, "GB/s", 1.0,
                  "BatchNorm/ReLU streaming kernels, bf16 storage", pmc=("k_bn_bwd_apply", "k_bn_bwd_reduce", "k_affine_act"),
                  peak_name=HBM),
        ]
        classes = [retarget(c) for c in classes if c]
        classes.sort(key=lambda c: -c["ms_per_step"])
        roof = dict(classes[0])                    # the dominant kernel class of the step
        roof["other_classes"] = classes[1:]
        roof["matrix_path"] = ("mixed precision: bf16 activations / weights images into v_mfma_f32_32x32x16_bf16, one MFMA "
                               "per product, f32 accumulate" if args.precision == "bf16" else
                               "f32 operands on the 16-bit matrix pipe: two-term fp16 split with per-tensor power-of-two scales, "
                               "3 fp16 MFMAs per product (GEMMs whose operands come with their maxima: *_h entry points) or the "
                               "exact three-term bf16 split, 6 bf16 MFMAs per product (fused forwards, recompute weight gradients); "
                               "f32 accumulate, f32-class error (tests/ + profiles/); SMAAT_F16_SPLIT=0 = three-term split only, "
                               "SMAAT_SPLIT=0 = f32-MFMA kernels only"
                               if split else "f32 MFMA (v_mfma_f32_32x32x2_f32) only")
        roof["definition"] = ("achieved = ALGORITHMIC flops (2 K Cout HW N per launch) or bytes (SURVEY 8(d)) / HIP-event time; "
                              "frac = achieved / peak for HBM-bound classes and for the f32-MFMA family; for the bf16-split "
                              "GEMM classes frac = executed / peak (matrix-pipe utilisation, see frac_definition) with the "
                              "algorithmic ratio kept beside it; traffic = HBM bytes per launch from the PMC passes of this "
                              "build (null when the record is stale)")
        if traffic_rec["stale"]:
            roof["traffic_note"] = traffic_rec["why"]

    # ---- per-frame forward latency (second half of BASELINE.json's metric): eval mode, batch 1 ----
    latency = None
    if rank == 0 and not args.no_latency:
        latency = fwd_latency(model, args.size, dev)
        model.train()

    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    # ---- the same step fed by the input pipeline: memory-mapped samples -> pinned ring -> async H2D
    #      on a copy stream -> batch-strided device views.  PCIe-inclusive; reported next to `value`, never as `value`.
    fed = fed_h5 = None
    if rank == 0 and args.gpus == 1 and not args.no_input_pipeline and args.precision == "f32":
        import tempfile
        from smaat_unet_amd.data import H5SampleSource, NpySampleSource, PrefetchLoader, write_precip_h5

        def fed_run(source, workers, epochs):
            loader = PrefetchLoader(source, args.batch, device=dev, depth=3, workers=workers, shuffle=True)
            nstep, t0 = 0, None
            for ep in range(epochs):
                for xb, yb in loader:
                    if nstep == 4:  # warm-up: page cache, pinned ring
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                    out = model(xb)
                    loss_f = torch.nn.functional.mse_loss(out.squeeze(1), yb, reduction="sum") / yb.size(0)
                    opt.zero_grad(set_to_none=True)
                    loss_f.backward()
                    opt.step()
                    nstep += 1
            torch.cuda.synchronize()
            dtf = time.perf_counter() - t0
            loader.close()
            return {"value": round(args.batch * (nstep - 4) / dtf, 2), "unit": "frames/s",
                    "ms_per_step": round(dtf / (nstep - 4) * 1e3, 3), "steps": nstep - 4, "gather_threads": workers,
                    "h2d_bytes_per_step": args.batch * 13 * args.size * args.size * 4}

        nsamp = 4 * args.batch
        rng = np.random.default_rng(7)
        path = os.path.join(tempfile.gettempdir(), f"smaat_bench_samples_{os.getpid()}.npy")
        try:
            arr = np.lib.format.open_memmap(path, mode="w+", dtype=np.float32, shape=(nsamp, 18, args.size, args.size))
            for i in range(nsamp):  # 18 frames per sample as in the reference's HDF5 layout (6 MB at 288 x 288)
                u = rng.random((18, args.size, args.size), dtype=np.float32)
                arr[i] = np.where(u > 0.7, (u - 0.7) / 0.3 * 0.5, 0)
            arr.flush()
            del arr
            fed = fed_run(NpySampleSource(path, 12), 8, 4)
            fed["what"] = ("same training step fed by smaat_unet_amd.data.PrefetchLoader from a memory-mapped .npy of "
                           "18-frame samples (13 of 18 frames gathered into pinned buffers by 8 threads, async H2D on a copy "
                           "stream, batch-strided device views): host gather + PCIe + step overlapped")
        except Exception as e:  # noqa: BLE001
            fed = {"error": str(e)[:200]}
        # ---- the reference's own container: HDF5, chunked (1, 3, 36, 72), gzip level 9 (create_datasets.py:33-40), read by
        #      the pure-Python reader (smaat_unet_amd/h5lite.py: B-tree chunk index + zlib inflate in the gather threads)
        h5path = os.path.join(tempfile.gettempdir(), f"smaat_bench_samples_{os.getpid()}.h5")
        try:
            nh5 = 2 * args.batch
            src = np.load(path, mmap_mode="r")[:nh5]
            t0 = time.perf_counter()
            write_precip_h5(h5path, {"train": np.asarray(src)}, level=9)
            t_write = time.perf_counter() - t0
            del src
            workers = max(8, min(96, (os.cpu_count() or 8) // 2))
            fed_h5 = fed_run(H5SampleSource(h5path, 12), workers, 8)
            fed_h5.update(file_mb=round(os.path.getsize(h5path) / 1e6, 1), raw_mb=round(nh5 * 18 * args.size * args.size * 4 / 1e6, 1),
                          write_s=round(t_write, 1),
                          what=f"same step fed from an HDF5 file in the reference's layout (train/images [{nh5}][18][{args.size}]"
                               f"[{args.size}] float32, chunks (1, 3, 36, 72), gzip level 9): the chunks of frames 0..11 and 17 of "
                               f"each sample located through the version-1 B-tree index and inflated with zlib by {workers} gather "
                               "threads, then the same pinned ring / async H2D path")
        except Exception as e:  # noqa: BLE001
            fed_h5 = {"error": str(e)[:200]}
        for pth in (path, h5path):
            if os.path.exists(pth):
                os.remove(pth)

    eager = None
    if rank == 0 and args.gpus == 1 and not args.no_eager_baseline and args.precision == "f32":
        eager = rocm_eager_baseline(args.batch, args.size, dev)

    alt = None
    if (rank == 0 and args.gpus == 1 and not args.no_alt and os.environ.get("SMAAT_SPLIT", "") != "0"
            and args.precision == "f32"):
        # same step with the f32-MFMA kernels only (no bf16 operand splitting), for reference
        import subprocess
        env = dict(os.environ, SMAAT_SPLIT="0")
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", str(min(args.steps, 5)), "--warmup",
                                "2", "--batch", str(args.batch), "--size", str(args.size), "--no-cpu-baseline",
                                "--no-profile", "--no-alt", "--no-latency", "--no-eager-baseline", "--no-input-pipeline"], env=env, capture_output=True, text=True, timeout=600)
            j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            alt = {"matrix_path": "f32 MFMA only (SMAAT_SPLIT=0)", "value": j["value"], "unit": j["unit"],
                   "ms_per_step": j["ms_per_step"]}
        except Exception as e:  # noqa: BLE001
            alt = {"error": str(e)[:200]}

    side = None
    if rank == 0 and args.gpus == 1 and not args.no_side_configs:
        del x, y
        torch.cuda.empty_cache()
        side = {"bf16_b64": side_config("bf16_b64", dev), "voc_b16": side_config("voc_b16", dev)}

    if rank == 0:
        frames = args.batch * world * args.steps
        line = {
            "metric": ("training frames/sec (256x256, 3-ch in, 21 classes)" if voc else
                       "training frames/sec (288x288, 12-ch in)"),
            "value": round(frames / dt, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "collective_backend": (backend + (" (RCCL %s)" % ".".join(map(str, torch.cuda.nccl.version()))
                                              if backend == "nccl" else "")) if world > 1 else None,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16": "bf16", "bf16_operands": "bf16 (GEMM operands; f32 storage and accumulation)"}[
                args.precision],
            "data": "synthetic",
            "config": {"workload": (f"SmaAt-UNet 3->21ch (PascalVOC head), {args.size}x{args.size} synthetic images, "
                                    f"batch={args.batch}/GPU fp32, fwd+CrossEntropy+bwd+Adam (BASELINE.json configs[4])"
                                    if voc else
                                    f"SmaAt-UNet 12->1ch, {args.size}x{args.size} synthetic precip, batch={args.batch}/GPU "
                                    + {"f32": "fp32", "bf16": "mixed precision (bf16 activation storage)",
                                       "bf16_operands": "bf16 GEMM operands"}[args.precision]
                                    + ", fwd+MSE+bwd+Adam (BASELINE.json configs["
                                    + ("3" if args.precision != "f32" else ("2" if world > 1 else "1")) + "])"),
                       "arithmetic": ("bf16 activation / activation-gradient storage, bf16 MFMA GEMMs (one per product), f32 "
                                      "accumulation, f32 BatchNorm statistics, f32 master weights and weight gradients"
                                      if args.precision == "bf16" else
                                      "f32 storage and accumulation; pointwise GEMMs on the 16-bit matrix pipe: two-term fp16 "
                                      "operand split with per-tensor power-of-two scales (3 MFMAs per product) where the operand "
                                      "maxima are at hand and the BatchNorm behind the GEMM averages >= 4096 samples, exact "
                                      "three-term bf16 split (6 MFMAs) elsewhere; f32-class error"),
                       "optimizer": {"one": "smaat_unet_amd.optim.Adam(lr=1e-3): torch.optim.Adam's multi-tensor update (same "
                                            "expressions in f32, default betas / eps) in one launch over all parameter tensors",
                                     "foreach": "torch.optim.Adam(lr=1e-3, foreach=True)",
                                     "fused": "torch.optim.Adam(lr=1e-3, fused=True)"}.get(adam_impl, adam_impl),
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "final_loss": round(final_loss, 5)},
            "roofline": roof,
            "gradient_exchange": (allreduce_model(ddp.numel * 4) if world == 1 else
                                  {"modelled": False, "bytes": ddp.numel * 4, "buckets": len(ddp._buckets),
                                   "note": f"measured inside ms_per_step ({backend} all-reduce of the flat buffer, two buckets)"}),
            "placement": {"rank_localrank_device": placement, "devices_visible": ndev,
                          "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")},
            "ddp_semantics": ("per-replica BatchNorm statistics, gradients averaged; stock DDP's per-step broadcast_buffers is "
                              "SKIPPED (every rank keeps the running statistics of its own shard, as with "
                              "DistributedDataParallel(broadcast_buffers=False)); smaat_unet_amd.ddp.sync_buffers() restores it"),
            "cpu_baseline": cpu,
            "rocm_eager_baseline": eager,
            "input_pipeline_fed": fed,
            "input_pipeline_fed_hdf5": fed_h5,
            "f32_mfma_only": alt,
            "configs": side,
            "power": power,
            "hbm_peak_allocated_gib": peak_gb,
            "fwd_latency": latency,
            "kernels": kernels,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()  # the other ranks wait for rank 0's per-kernel / latency passes before tearing down
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

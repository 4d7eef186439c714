"""PrecipitationMetrics: the oracle restatement, the host class over the emulated C ABI
(CPU) and the HIP kernel (GPU) against goldens produced by the reference's own class
(oracle/gen_golden.py gen_metrics, /root/reference/metric/precipitation_metrics.py)."""
import ctypes
import json
import math
import os

import numpy as np
import pytest
import torch

import smaat_unet_amd as S
from oracle import smaat_oracle as O
from tests import emu_backend

CASES = ["default", "nodenorm", "thr2"]
STATE_KEYS = ("total_loss", "total_loss_denorm", "total_samples", "total_pixels", "total_tp", "total_fp", "total_tn",
              "total_fn")


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "precip_metrics.npz"))


def _close(a, b, tol=2e-5):
    if math.isnan(b):
        return math.isnan(a)
    return abs(a - b) <= tol * max(abs(b), 1e-12)


def _check(gold, tag, state, computed):
    for k in STATE_KEYS:
        ref = float(gold[f"{tag}/state/{k}"])
        if k.startswith("total_loss"):
            assert _close(float(state[k]), ref), (k, state[k], ref)  # the reference accumulates float32 scalars
        else:
            assert int(state[k]) == int(ref), (k, state[k], ref)      # counts are exact
    for k in ("mse", "mse_denorm", "mse_pixel", "precision", "recall", "accuracy", "f1", "csi", "far", "hss"):
        assert _close(float(computed[k]), float(gold[f"{tag}/compute/{k}"])), (k, computed[k])


@pytest.mark.parametrize("tag", CASES)
def test_oracle_restatement_matches_reference(gold, tag):
    cfg = json.loads(str(gold[f"{tag}/cfg"]))
    st = O.precip_metrics_new_state()
    for b in range(3):
        O.precip_metrics_update(st, gold[f"{tag}/b{b}/preds"], gold[f"{tag}/b{b}/target"], cfg["threshold"],
                                cfg["denormalize"])
    assert st["nan_batches"] == (1 if tag == "default" else 0)
    _check(gold, tag, st, O.precip_metrics_compute(st, cfg["denormalize"]))


def _run_host(gold, tag, dev):
    cfg = json.loads(str(gold[f"{tag}/cfg"]))
    m = S.PrecipitationMetrics(threshold=cfg["threshold"], denormalize=cfg["denormalize"])
    for b in range(3):
        m.update(torch.from_numpy(gold[f"{tag}/b{b}/preds"]).to(dev), torch.from_numpy(gold[f"{tag}/b{b}/target"]).to(dev))
    st = m.state()
    assert st["nan_batches"] == (1 if tag == "default" else 0)
    _check(gold, tag, st, m.compute())
    m.reset()
    assert all(v == 0 for v in m.state().values())


@pytest.mark.parametrize("tag", CASES)
def test_host_class_over_emulated_abi(gold, tag):
    emu_backend.install()
    try:
        _run_host(gold, tag, "cpu")
    finally:
        emu_backend.uninstall()


def test_host_class_has_no_cpu_fallback():
    m = S.PrecipitationMetrics()
    with pytest.raises(Exception, match="no CPU fallback"):
        m.update(torch.zeros(1, 1, 4, 4), torch.zeros(1, 4, 4))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_gpu_host_class_vs_reference_golden(gold, tag):
    _run_host(gold, tag, torch.device("cuda:0"))


@pytest.mark.gpu
@pytest.mark.parametrize("n,batch,denorm,thr,nan_at", [(32 * 288 * 288, 32, 1, 0.5, None), (1000003, 7, 1, 0.5, None),
                                                       (37, 1, 0, 2.0, None), (4096, 4, 1, 0.5, 4095),
                                                       (5, 5, 1, 0.5, 0)])
def test_gpu_kernel_vs_oracle(n, batch, denorm, thr, nan_at):
    """the C ABI entry point itself: full bench size, a size that is not a multiple of 4 (unaligned tail), tiny
    sizes, a NaN in the last / first element; two calls accumulate"""
    from smaat_unet_amd import _lib
    L = _lib.get()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n % 1000 + 1)
    u = rng.random(n, dtype=np.float32)
    t = np.where(u > 0.6, (u - 0.6) * 0.02, 0).astype(np.float32)
    p = (t + 0.002 * rng.standard_normal(n).astype(np.float32)).astype(np.float32)
    if nan_at is not None:
        p[nan_at] = np.nan
    pt, tt = torch.from_numpy(p).to(dev), torch.from_numpy(t).to(dev)
    ws = torch.empty((L.smaat_precip_metrics_ws_bytes(n) + 7) // 8, dtype=torch.float64, device=dev)
    sf = torch.zeros(2, dtype=torch.float64, device=dev)
    si = torch.zeros(7, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        assert L.smaat_precip_metrics_update(pt.data_ptr(), tt.data_ptr(), n, batch, O.PRECIP_FACTOR, thr, denorm,
                                             ws.data_ptr(), sf.data_ptr(), si.data_ptr(), st) == 0
    ref = O.precip_metrics_new_state()
    for _ in range(2):
        O.precip_metrics_update(ref, p.reshape(batch, -1) if n % batch == 0 else p[None], t.reshape(batch, -1)
                                if n % batch == 0 else t[None], thr, bool(denorm))
    if n % batch:  # the oracle divided by its own batch (1): rescale to the batch passed to the kernel
        ref["total_loss"] /= batch
        ref["total_loss_denorm"] /= batch
        ref["total_samples"] = 0 if nan_at is not None else 2 * batch
    f, i = sf.tolist(), si.tolist()
    assert i[0] == ref["nan_batches"]
    assert [i[1], i[2], i[3], i[4]] == [ref["total_tn"], ref["total_fp"], ref["total_fn"], ref["total_tp"]]
    assert i[5] == ref["total_samples"] and i[6] == ref["total_pixels"]
    assert _close(f[0], ref["total_loss"], 1e-5) and _close(f[1], ref["total_loss_denorm"], 1e-5)


def _metrics_worker(rank, world, port, out_dir):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import emu_backend
    emu_backend.install()
    from smaat_unet_amd import PrecipitationMetrics
    m = PrecipitationMetrics(device="cpu")
    if rank == 0:  # rank 1 never sees a batch: it must still take part in the collective, with zeros
        g = torch.Generator().manual_seed(0)
        t = torch.rand(2, 8, 8, generator=g) * 0.02
        m.update(t + 0.001, t)
    a = m.compute()
    b = m.compute()      # a second compute must not double count (the reduction runs on copies)
    local = m.state(sync=False)
    torch.save(dict(a=a, b=b, local=local), os.path.join(out_dir, f"m{rank}.pt"))
    dist.destroy_process_group()


def test_metrics_distributed_compute_reduces_copies(tmp_path):
    """compute() under torch.distributed = totals over the ranks, from copies of the state; a rank without
    updates joins the collective; repeated compute() calls agree"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_metrics_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "m0.pt"), torch.load(tmp_path / "m1.pt")
    assert r0["a"] == r0["b"] or all((r0["a"][k] == r0["b"][k]) or (r0["a"][k] != r0["a"][k]) for k in r0["a"])
    for k in r0["a"]:
        x, y = r0["a"][k], r1["a"][k]
        assert x == y or (x != x and y != y), k          # both ranks report the same totals
    assert r0["local"]["total_samples"] == 2 and r1["local"]["total_samples"] == 0
    assert r0["a"]["mse"] == r0["a"]["mse"]               # not NaN: the totals include rank 0's batch on both ranks
    with pytest.raises(NotImplementedError):
        from smaat_unet_amd import PrecipitationMetrics
        PrecipitationMetrics(dist_sync_on_step=True)

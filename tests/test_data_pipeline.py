"""Input pipeline (reference utils/dataset_precip.py:63-77, models/regression_lightning.py:178-199):
the sample contract of the reference dataset and the pinned-ring prefetcher."""
import numpy as np
import pytest
import torch

from smaat_unet_amd.data import NpySampleSource, PrefetchLoader


def _array(s=10, t=6, h=8, w=8):
    a = np.arange(s * t * h * w, dtype=np.float32).reshape(s, t, h, w)
    return a


def test_sample_contract_matches_the_reference_dataset(tmp_path):
    a = _array()
    path = tmp_path / "images.npy"
    np.save(path, a)
    src = NpySampleSource(str(path), num_input_images=4)
    assert len(src) == 10
    x, y = src[3]
    # reference: imgs = np.array(dataset[index], dtype="float32"); input = imgs[:num_input]; target = imgs[-1]
    assert x.dtype == np.float32 and x.shape == (4, 8, 8) and np.array_equal(x, a[3, :4])
    assert y.shape == (8, 8) and np.array_equal(y, a[3, -1])
    src2 = NpySampleSource(a, 4, transform=lambda im: im * 2)
    x2, y2 = src2[1]
    assert np.array_equal(x2, 2 * a[1, :4]) and np.array_equal(y2, 2 * a[1, -1])
    with pytest.raises(ValueError):
        NpySampleSource(a, 6)
    with pytest.raises(ValueError):
        NpySampleSource(a.astype(np.float64), 4)


@pytest.mark.parametrize("workers", [1, 3])
def test_prefetch_loader_epoch_on_cpu(workers):
    a = _array(s=11)
    src = NpySampleSource(a, 4)
    ld = PrefetchLoader(src, batch_size=3, device="cpu", depth=3, workers=workers, shuffle=True, seed=5, drop_last=True)
    assert len(ld) == 3
    seen = []
    for x, y in ld:
        assert x.shape == (3, 4, 8, 8) and y.shape == (3, 8, 8)
        for b in range(3):
            i = int(x[b, 0, 0, 0].item()) // (6 * 64)
            assert np.array_equal(x[b].numpy(), a[i, :4]) and np.array_equal(y[b].numpy(), a[i, -1])
            seen.append(i)
    assert len(seen) == 9 and len(set(seen)) == 9
    first = list(seen)
    seen2 = [int(x[b, 0, 0, 0].item()) // (6 * 64) for x, _ in ld for b in range(3)]
    assert seen2 != first  # next epoch, another permutation
    ld2 = PrefetchLoader(src, 3, device="cpu", shuffle=True, seed=5)
    assert [int(x[b, 0, 0, 0].item()) // (6 * 64) for x, _ in ld2 for b in range(3)] == first  # same seed, same order
    ld3 = PrefetchLoader(src, 4, device="cpu", shuffle=False, drop_last=False)
    sizes = [x.shape[0] for x, _ in ld3]
    assert sizes == [4, 4, 3]


def test_prefetch_loader_early_exit_does_not_hang():
    src = NpySampleSource(_array(s=40), 4)
    ld = PrefetchLoader(src, 2, device="cpu", depth=2)
    for k, _ in enumerate(ld):
        if k == 2:
            break
    assert sum(1 for _ in ld) == 20


@pytest.mark.gpu
def test_prefetch_loader_feeds_the_model_on_gpu(tmp_path):
    """pinned ring + async H2D on a side stream: the batches arrive intact as batch-strided views that the kernels
    take without a copy, while earlier batches are still being consumed"""
    import smaat_unet_amd as S
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    a = rng.random((24, 18, 64, 64), dtype=np.float32)
    src = NpySampleSource(a, 12)
    ld = PrefetchLoader(src, 4, device=dev, depth=3, workers=2, shuffle=False)
    torch.manual_seed(0)
    model = S.SmaAt_UNet(12, 1).to(dev).train()
    k = 0
    for x, y in ld:
        assert x.is_cuda and x.shape == (4, 12, 64, 64) and x.stride(0) == 13 * 64 * 64 and not x.is_contiguous()
        out = model(x)                                   # consumes the strided view directly
        loss = torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / 4
        loss.backward()
        assert torch.equal(x.cpu(), torch.from_numpy(a[4 * k:4 * k + 4, :12]))
        assert torch.equal(y.cpu(), torch.from_numpy(a[4 * k:4 * k + 4, -1]))
        ref = model(x.contiguous())
        assert torch.equal(out, ref)
        k += 1
    assert k == 6


@pytest.mark.parametrize("workers", [1, 3])
def test_prefetch_loader_propagates_worker_errors(workers):
    """an exception in the gather threads / the transform must surface in the consuming loop (torch's DataLoader
    re-raises worker errors); round-2 behaviour was a silently truncated epoch (workers=1) or stale pinned-buffer
    contents delivered as a batch (workers>1)"""
    a = _array(s=12)

    def bad(imgs):
        if imgs[0, 0, 0] >= 6 * 6 * 64:  # samples 6 and up
            raise ValueError("corrupt sample")
        return imgs

    src = NpySampleSource(a, 4, transform=bad)
    ld = PrefetchLoader(src, 6, device="cpu", depth=2, workers=workers, shuffle=False)
    got = 0
    with pytest.raises(RuntimeError, match="corrupt sample"):
        for _x, _y in ld:
            got += 1
    assert got == 1  # the first batch (samples 0..5) is fine, the second never arrives


def test_prefetch_loader_rank_sharding_and_close():
    a = _array(s=13)
    src = NpySampleSource(a, 4)
    seen = []
    for r in range(2):
        ld = PrefetchLoader(src, 3, device="cpu", shuffle=True, seed=7, rank=r, world_size=2)
        assert len(ld) == 2
        ids = [int(x[b, 0, 0, 0].item()) // (6 * 64) for x, _ in ld for b in range(x.shape[0])]
        assert len(ids) == 6
        seen.append(ids)
    assert not set(seen[0]) & set(seen[1])  # disjoint shards of one permutation
    ld.close()
    with pytest.raises(RuntimeError, match="close"):
        next(iter(ld))
    with pytest.raises(ValueError):
        PrefetchLoader(src, 3, device="cpu", rank=2, world_size=2)

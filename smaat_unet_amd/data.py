"""Input pipeline for the precipitation training path.

Reference: `precipitation_maps_oversampled_h5.__getitem__` (/root/reference/utils/dataset_precip.py:63-77) reads one
float32 sample of shape [T = 18][288][288] (6 MB) from an HDF5 file and returns `(imgs[:num_input], imgs[-1])`; the
Lightning data module wraps it in single-worker `DataLoader`s (models/regression_lightning.py:178-199).  At the
~800 frames/s/GPU of the MI355X training step that loader cannot feed the model (4.8 GB/s host->device per GPU), so
the path either side of the model is rebuilt here:

  * `NpySampleSource` -- the same `__getitem__` / `__len__` contract over a memory-mapped `.npy` array
    [samples][T][H][W] (the layout of the HDF5 dataset `train/images`; `h5py` is not available in this image, a
    one-off `np.save` of that dataset produces the file);
  * `PrefetchLoader` -- worker threads gather the frames a batch needs (`[:num_input]` and `[-1]`, 13 of 18) straight
    into a ring of PINNED host buffers, a dedicated HIP stream copies each buffer to a ring of device buffers
    asynchronously, and the training loop receives `(x, y)` as VIEWS of the device buffer (batch-strided, dense planes --
    the layout every smaat_unet_amd kernel accepts), after making its stream wait for the copy's event.  Host gather,
    PCIe copy and GPU compute of three consecutive batches overlap; nothing is copied twice and no kernel is launched
    for the slicing.

`bench.py --input-pipeline` measures the training step fed by this loader (PCIe-inclusive).
"""
from __future__ import annotations

import queue
import threading

import numpy as np
import torch


class NpySampleSource:
    """`samples[index] -> (input [num_input][H][W], target [H][W])`, the reference dataset's contract
    (utils/dataset_precip.py:63-77: `input_img = imgs[:num_input]`, `target_img = imgs[-1]`)."""

    def __init__(self, array_or_path, num_input_images=12, transform=None):
        if isinstance(array_or_path, (str, bytes)) or hasattr(array_or_path, "__fspath__"):
            self.data = np.load(array_or_path, mmap_mode="r")
        else:
            self.data = array_or_path
        if self.data.ndim != 4 or self.data.dtype != np.float32:
            raise ValueError(f"expected a float32 array [samples][T][H][W], got {self.data.dtype} {self.data.shape}")
        if not 0 < num_input_images < self.data.shape[1]:
            raise ValueError("num_input_images must leave at least one frame for the target")
        self.num_input = int(num_input_images)
        self.transform = transform

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, index):
        imgs = np.array(self.data[index], dtype="float32")
        if self.transform is not None:
            imgs = self.transform(imgs)
        return imgs[: self.num_input], imgs[-1]

    def gather_into(self, indices, dst):
        """dst [B][num_input + 1][H][W] (numpy view of a pinned buffer) <- the frames each sample contributes"""
        ni = self.num_input
        for b, i in enumerate(indices):
            s = self.data[i]
            if self.transform is not None:
                s = self.transform(np.array(s, dtype="float32"))
            dst[b, :ni] = s[:ni]
            dst[b, ni] = s[-1]


class H5SampleSource:
    """The reference's dataset itself: `precipitation_maps_oversampled_h5` (/root/reference/utils/dataset_precip.py:48-80)
    over the HDF5 file /root/reference/create_datasets.py writes -- group "train" | "test", dataset "images"
    [samples][T][H][W] float32, chunked, gzip level 9.  `h5py` is not in this image: smaat_unet_amd.h5lite reads the format
    directly (B-tree chunk index walk + zlib).  Same contract as NpySampleSource (`__getitem__`, `__len__`, `gather_into`,
    `.data.shape`), so PrefetchLoader takes either.

    gather_into inflates ONLY the chunks that hold the frames a sample contributes (`imgs[:num_input]` and `imgs[-1]`): with
    h5py's chunk guess for the real geometry ((1, 3, 36, 72) for 18 x 288 x 288) that is 5 of the 6 chunk rows.  zlib
    releases the GIL, so the loader's gather threads inflate in parallel; one descriptor, positional reads."""

    def __init__(self, in_file, num_input_images=12, train=True, transform=None, dataset="images"):
        from .h5lite import H5File
        self.file = H5File(in_file)
        self.data = self.file["train" if train else "test"][dataset]
        if self.data.ndim != 4 or self.data.dtype != np.float32:
            raise ValueError(f"expected a float32 dataset [samples][T][H][W], got {self.data.dtype} {self.data.shape}")
        if not 0 < num_input_images < self.data.shape[1]:
            raise ValueError("num_input_images must leave at least one frame for the target")
        self.num_input = int(num_input_images)
        self.transform = transform
        self._frames = list(range(self.num_input)) + [self.data.shape[1] - 1]
        self._tls = threading.local()
        # native gather (libsmaat_io.so: zlib inflate + scatter of one sample per foreign call, GIL released): frame f of a
        # sample goes to destination frame fmap[f] -- inputs in place, the last frame behind them, the rest skipped
        self._fmap = np.full(self.data.shape[1], -1, np.int32)
        self._fmap[: self.num_input] = np.arange(self.num_input)
        self._fmap[-1] = self.num_input
        self.native = transform is None and self.data.native_ok()

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, index):
        imgs = self.data[index]  # np.array(self.dataset[index], dtype="float32")  (dataset_precip.py:69)
        if self.transform is not None:
            imgs = self.transform(imgs)
        return imgs[: self.num_input], imgs[-1]

    def gather_into(self, indices, dst):
        ni = self.num_input
        if self.native:
            for b, i in enumerate(indices):
                self.data.gather_native(int(i), self._fmap, dst[b])
            return
        buf = getattr(self._tls, "buf", None)
        if buf is None:
            buf = self._tls.buf = np.empty(self.data.shape[1:], np.float32)  # one sample of scratch per gather thread
        for b, i in enumerate(indices):
            if self.transform is not None:
                s = self.transform(self.data.read_into(int(i), buf))
            else:
                s = self.data.read_into(int(i), buf, frames=self._frames)
            dst[b, :ni] = s[:ni]
            dst[b, ni] = s[-1]

    def close(self):
        self.file.close()


def write_precip_h5(path, splits, chunks=None, level=9):
    """{"train": array, "test": array} (float32 [samples][T][H][W]) -> an HDF5 file in the reference's dataset layout
    (create_datasets.py:31-61: chunked, gzip; chunks default to what h5py guesses for a (1, T, H, W) creation shape at
    288 x 288: (1, 3, 36, 72)).  Used to lay down synthetic datasets (bench.py, tests) without libhdf5."""
    from .h5lite import write_images_h5
    any_arr = next(iter(splits.values()))
    if chunks is None:
        t, h, w = any_arr.shape[1:]
        chunks = (1, min(3, t), min(36, h), min(72, w))
    write_images_h5(path, splits, chunks, level=level)


class PrefetchLoader:
    """Iterate `(x [B][num_input][H][W], y [B][H][W])` device tensors over `source`.

    depth      ring size (pinned host buffers = device buffers): batches in flight
    workers    gather threads (numpy releases the GIL while copying)
    rank / world_size   data-parallel sharding: every rank draws the SAME seeded permutation and takes the samples
               rank, rank + world_size, ... of it (torch.utils.data.DistributedSampler's rule, drop_last semantics), so
               that the replicas of smaat_unet_amd.ddp.FlatGradAllReduce see disjoint batches
    The tensors of one iteration stay valid until `depth - 1` further batches have been requested.
    An exception in a gather thread, in the dataset's transform or in the copy is re-raised in the consuming loop (as
    torch's DataLoader re-raises worker errors); no batch is delivered after it.  close() releases the pinned and device
    rings (the loader is unusable afterwards).
    """

    def __init__(self, source, batch_size, device="cuda", depth=3, workers=4, shuffle=True, seed=0, drop_last=True,
                 rank=0, world_size=1):
        if not 0 <= int(rank) < int(world_size):
            raise ValueError("rank must be in [0, world_size)")
        self.rank, self.world = int(rank), int(world_size)
        self.source, self.batch, self.depth = source, int(batch_size), max(2, int(depth))
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.workers = max(1, int(workers))
        self.shuffle, self.seed, self.drop_last = shuffle, seed, drop_last
        self.epoch = 0
        s0 = source.data.shape
        self.shape = (self.batch, source.num_input + 1, s0[2], s0[3])
        self.host = [torch.empty(self.shape, dtype=torch.float32, pin_memory=self.cuda) for _ in range(self.depth)]
        self.host_np = [h.numpy() for h in self.host]
        self.dev = [torch.empty(self.shape, dtype=torch.float32, device=self.device) for _ in range(self.depth)] \
            if self.cuda else self.host
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.cuda else None

    def _shard_len(self):
        return len(self.source) // self.world  # every rank the same count (the tail of the permutation is dropped)

    def __len__(self):
        n = self._shard_len()
        return n // self.batch if self.drop_last else (n + self.batch - 1) // self.batch

    def close(self):
        """release the pinned host ring and the device ring"""
        self.host, self.host_np, self.dev = [], [], []

    def _batches(self):
        n = len(self.source)
        order = np.arange(n)
        if self.shuffle:
            np.random.default_rng(self.seed + self.epoch).shuffle(order)
        order = order[self.rank:self._shard_len() * self.world:self.world]
        n = len(order)
        stop = n - n % self.batch if self.drop_last else n
        return [order[i:i + self.batch] for i in range(0, stop, self.batch)]

    def _gather(self, slot, idx):
        dst = self.host_np[slot]
        nb = len(idx)
        # (a cheap source -- memory-mapped rows -- is not worth a thread per couple of samples; a source that inflates
        # compressed chunks is: one sample per thread at most)
        heavy = getattr(self.source, "native", None) is not None
        if self.workers == 1 or nb < 2 or (not heavy and nb < 2 * self.workers):
            self.source.gather_into(idx, dst)
            return
        parts = [p for p in np.array_split(np.arange(nb), min(self.workers, nb)) if len(p)]
        errors = []

        def work(p):
            try:
                self.source.gather_into(idx[p], dst[p[0]:p[-1] + 1])
            except BaseException as e:  # noqa: BLE001  (handed to the consumer, never swallowed)
                errors.append(e)

        ths = [threading.Thread(target=work, args=(p,)) for p in parts]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errors:
            raise errors[0]

    def __iter__(self):
        batches = self._batches()
        self.epoch += 1
        ni = self.source.num_input
        free = queue.Queue()          # slots the producer may fill
        ready = queue.Queue(maxsize=self.depth)
        for s in range(self.depth):
            free.put((s, None))
        stop = threading.Event()

        def producer():
            if not self.host:
                ready.put(RuntimeError("PrefetchLoader.close() was called"))
                return
            try:
                for idx in batches:
                    slot, reuse_ev = free.get()
                    if stop.is_set():
                        return
                    if reuse_ev is not None:
                        reuse_ev.synchronize()  # the compute stream is done with this slot's device buffer
                    self._gather(slot, idx)
                    ev = None
                    if self.cuda:
                        with torch.cuda.stream(self.copy_stream):
                            self.dev[slot][:len(idx)].copy_(self.host[slot][:len(idx)], non_blocking=True)
                            ev = torch.cuda.Event()
                            ev.record(self.copy_stream)
                    ready.put((slot, len(idx), ev))
            except BaseException as e:  # noqa: BLE001  (gather / transform / copy failed: the consumer re-raises it)
                ready.put(e)
                return
            ready.put(None)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        held = []  # slots handed to the consumer, oldest first
        try:
            while True:
                item = ready.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise RuntimeError(f"PrefetchLoader: the producer thread failed: {item!r}") from item
                slot, nb, ev = item
                if ev is not None:
                    torch.cuda.current_stream(self.device).wait_event(ev)
                buf = self.dev[slot][:nb]
                held.append(slot)
                if len(held) > self.depth - 1:  # the oldest slot may be refilled once the work queued so far is done
                    old = held.pop(0)
                    done = None
                    if self.cuda:
                        done = torch.cuda.Event()
                        done.record(torch.cuda.current_stream(self.device))
                    free.put((old, done))
                yield buf[:, :ni], buf[:, ni]
        finally:
            stop.set()
            free.put((0, None))
            th.join(timeout=5)

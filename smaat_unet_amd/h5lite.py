"""A small HDF5 reader (and a minimal writer) in pure Python: exactly the part of the file format the reference's
datasets use, so that the input pipeline can read them without `h5py` (absent from this image).

What /root/reference/create_datasets.py:25-61 writes through h5py (libver "earliest", the h5py default):
  superblock version 0 -> root group (symbol table: B-tree v1 + local heap + symbol nodes) -> groups "train" / "test" ->
  dataset "images": float32 [samples][T][H][W], maxshape (None, T, H, W), CHUNKED layout (message version 3: a version-1
  B-tree of raw-data chunks keyed by element offsets), filter pipeline = deflate level 9 (`compression="gzip",
  compression_opts=9`); `dataset_precip.py:63-77` then reads `images[index]` = the 18-frame sample `index`.

Reader: `H5File(path)[name]` walks groups; `H5Dataset` exposes `shape`, `dtype`, `chunks`, `__getitem__(i)` for a leading
index and `read_into(i, out, frames=None)` (only the chunks that hold the requested frames of sample i are inflated).  Reads
go through os.pread on one descriptor: safe from many gather threads at once; zlib releases the GIL while inflating.
Supported: superblock 0/1, version-1 object headers (with continuation blocks), symbol-table groups, dataspace 1/2,
little-endian IEEE float32/float64 and 1/2/4/8-byte integers, layout version 3 (contiguous and chunked), filters deflate (1),
shuffle (2), fletcher32 (3).  Files written with libver="latest" (superblock 2/3, layout version 4 chunk indices) are
rejected with a message saying so.

Writer (`write_images_h5`): the same structures, one chunked + deflated float32 dataset "images" per group -- what bench.py
needs to lay down a synthetic dataset on a box without libhdf5.  Both directions are pinned against the real library:
tests/golden/precip_h5_fixture.h5 was written by libhdf5 1.10.6 (oracle/h5_fixture/) and must read back bit-exactly here, and
the files written here are read back by the real `h5dump` in oracle/h5_fixture/gen_h5_fixture.py.

Format reference: "HDF5 File Format Specification Version 2.0/3.0" (superblock III.A, B-trees III.A.1, symbol nodes III.B,
local heaps III.D, object headers IV.A.1.a, messages IV.A.2.*)."""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF

MSG_DATASPACE, MSG_DATATYPE, MSG_FILL_OLD, MSG_FILL, MSG_LAYOUT, MSG_FILTERS = 0x1, 0x3, 0x4, 0x5, 0x8, 0xB
MSG_CONTINUATION, MSG_SYMBOL_TABLE, MSG_LINK_INFO = 0x10, 0x11, 0x2


class H5FormatError(ValueError):
    pass


class H5File:
    def __init__(self, path):
        self.path = os.fspath(path)
        self.fd = os.open(self.path, os.O_RDONLY)
        try:
            self._superblock()
        except Exception:
            os.close(self.fd)
            self.fd = None
            raise

    def close(self):
        if self.fd is not None:
            os.close(self.fd)
            self.fd = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # ---- raw access -----------------------------------------------------------------------------------------------
    def read(self, off, n):
        b = os.pread(self.fd, n, self.base + off)
        if len(b) != n:
            raise H5FormatError(f"{self.path}: short read at {off} ({len(b)} of {n} bytes): truncated file?")
        return b

    def _superblock(self):
        size = os.fstat(self.fd).st_size
        off, self.base = 0, 0
        while True:
            if off + 8 > size:
                raise H5FormatError(f"{self.path}: no HDF5 signature found")
            if os.pread(self.fd, 8, off) == SIGNATURE:
                break
            off = 512 if off == 0 else off * 2
        head = os.pread(self.fd, 64 + 48, off)
        ver = head[8]
        if ver >= 2:
            raise H5FormatError(f"{self.path}: superblock version {ver} (a file written with libver='latest'): this reader "
                                "takes the h5py default format (libver='earliest': superblock 0/1, version-1 B-tree chunk index)")
        so, sl = head[13], head[14]
        if (so, sl) != (8, 8):
            raise H5FormatError(f"{self.path}: {so}-byte offsets / {sl}-byte lengths (only 8/8 is handled)")
        p = 24 + (4 if ver == 1 else 0)
        base, _free, self.eof, _drv = struct.unpack_from("<4Q", head, p)
        self.base = off + base if base != UNDEF else off
        p += 32
        _name_off, root_hdr, cache, _res = struct.unpack_from("<QQII", head, p)
        self.root = H5Group(self, root_hdr)

    def __getitem__(self, name):
        return self.root[name]

    def keys(self):
        return self.root.keys()

    # ---- object headers (version 1) ---------------------------------------------------------------------------------
    def messages(self, addr):
        """[(type, flags, bytes)] of the object header at addr, continuation blocks included"""
        pre = self.read(addr, 16)
        if pre[:4] == b"OHDR":
            raise H5FormatError(f"{self.path}: version-2 object header (libver='latest' file): not handled, see H5File")
        ver, _r, nmsgs, _refc, hsize = struct.unpack_from("<BBHII", pre, 0)
        if ver != 1:
            raise H5FormatError(f"{self.path}: object header version {ver} at {addr}")
        blocks = [(addr + 16, hsize)]
        out = []
        while blocks and len(out) < nmsgs:
            boff, blen = blocks.pop(0)
            blk = self.read(boff, blen)
            p = 0
            while p + 8 <= blen and len(out) < nmsgs:
                mtype, msize, mflags = struct.unpack_from("<HHB", blk, p)
                data = blk[p + 8:p + 8 + msize]
                p += 8 + msize
                if mtype == MSG_CONTINUATION:
                    coff, clen = struct.unpack_from("<QQ", data, 0)
                    blocks.append((coff, clen))
                out.append((mtype, mflags, data))
        return out


class H5Group:
    def __init__(self, f, addr):
        self.f, self.addr = f, addr
        self._links = None

    def _load(self):
        if self._links is not None:
            return
        msgs = self.f.messages(self.addr)
        st = [d for t, _, d in msgs if t == MSG_SYMBOL_TABLE]
        if not st:
            if any(t == MSG_LINK_INFO for t, _, _ in msgs):
                raise H5FormatError(f"{self.f.path}: new-style group (link messages): a libver='latest' file, see H5File")
            raise H5FormatError(f"{self.f.path}: object at {self.addr} is not a group")
        btree, heap = struct.unpack_from("<QQ", st[0], 0)
        hh = self.f.read(heap, 32)
        if hh[:4] != b"HEAP":
            raise H5FormatError(f"{self.f.path}: bad local heap at {heap}")
        hsize, _free, hdata = struct.unpack_from("<QQQ", hh, 8)
        names = self.f.read(hdata, hsize)
        links = {}
        self._walk(btree, names, links)
        self._links = links

    def _walk(self, addr, names, links):
        h = self.f.read(addr, 24)
        if h[:4] != b"TREE" or h[4] != 0:
            raise H5FormatError(f"{self.f.path}: bad group B-tree node at {addr}")
        level, n = h[5], struct.unpack_from("<H", h, 6)[0]
        body = self.f.read(addr + 24, n * 16 + 8)
        for i in range(n):
            child = struct.unpack_from("<Q", body, i * 16 + 8)[0]
            if level > 0:
                self._walk(child, names, links)
                continue
            sn = self.f.read(child, 8)
            if sn[:4] != b"SNOD":
                raise H5FormatError(f"{self.f.path}: bad symbol node at {child}")
            ns = struct.unpack_from("<H", sn, 6)[0]
            ents = self.f.read(child + 8, ns * 40)
            for j in range(ns):
                noff, ohdr = struct.unpack_from("<QQ", ents, j * 40)
                end = names.index(b"\0", noff)
                links[names[noff:end].decode("utf-8")] = ohdr

    def keys(self):
        self._load()
        return sorted(self._links)

    def __contains__(self, name):
        self._load()
        return name in self._links

    def __getitem__(self, name):
        self._load()
        node = self
        parts = [p for p in name.split("/") if p]
        if len(parts) > 1:
            for p in parts:
                node = node[p]
            return node
        if name not in self._links:
            raise KeyError(f"{name!r} not in group (has {self.keys()})")
        addr = self._links[name]
        types = {t for t, _, _ in self.f.messages(addr)}
        if MSG_LAYOUT in types:
            return H5Dataset(self.f, addr)
        return H5Group(self.f, addr)


_INT = {1: "i1", 2: "i2", 4: "i4", 8: "i8"}


class H5Dataset:
    def __init__(self, f, addr):
        self.f, self.addr = f, addr
        self.fill = None
        self.filters = []
        self.chunks = None
        for t, _, d in f.messages(addr):
            if t == MSG_DATASPACE:
                self._dataspace(d)
            elif t == MSG_DATATYPE:
                self._datatype(d)
            elif t == MSG_LAYOUT:
                self._layout(d)
            elif t == MSG_FILTERS:
                self._filters(d)
            elif t == MSG_FILL:
                self._fill(d)
        self.ndim = len(self.shape)
        if self.chunks is not None:
            self._index = None  # chunk index, built on first access

    # ---- header messages -----------------------------------------------------------------------------------------
    def _dataspace(self, d):
        ver, rank, flags = d[0], d[1], d[2]
        p = 8 if ver == 1 else 4
        self.shape = tuple(struct.unpack_from(f"<{rank}Q", d, p)) if rank else ()
        self.maxshape = None
        if flags & 1:
            ms = struct.unpack_from(f"<{rank}Q", d, p + 8 * rank)
            self.maxshape = tuple(None if m == UNDEF else m for m in ms)

    def _datatype(self, d):
        cls, bits0, size = d[0] & 0x0F, d[1], struct.unpack_from("<I", d, 4)[0]
        if cls in (0, 1) and bits0 & 1:
            raise H5FormatError("big-endian datasets are not handled")
        if cls == 1 and size in (4, 8):
            self.dtype = np.dtype("<f4" if size == 4 else "<f8")
        elif cls == 0 and size in _INT:
            self.dtype = np.dtype("<" + (_INT[size] if d[1] & 8 else _INT[size].replace("i", "u")))
        else:
            self.dtype = None  # (variable-length strings, compounds ...: the object can be listed, not read)
        self.itemsize = size

    def _layout(self, d):
        ver, cls = d[0], d[1]
        if ver != 3:
            raise H5FormatError(f"{self.f.path}: data layout message version {ver} (version 4 = libver='latest' chunk indices; "
                                "version 1/2 = pre-1.6 files): this reader takes version 3, the h5py default")
        self.layout = {0: "compact", 1: "contiguous", 2: "chunked"}[cls]
        if cls == 1:
            self.data_addr, self.data_size = struct.unpack_from("<QQ", d, 2)
        elif cls == 2:
            nd = d[2]
            self.btree = struct.unpack_from("<Q", d, 3)[0]
            dims = struct.unpack_from(f"<{nd}I", d, 11)
            self.chunks = tuple(dims[:-1])
        else:
            n = struct.unpack_from("<H", d, 2)[0]
            self.compact = d[4:4 + n]

    def _filters(self, d):
        ver, n = d[0], d[1]
        p = 8 if ver == 1 else 2
        for _ in range(n):
            fid = struct.unpack_from("<H", d, p)[0]
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = struct.unpack_from("<H", d, p)[0]
                p += 2
            flags, ncd = struct.unpack_from("<HH", d, p)
            p += 4
            if ver == 1:
                nlen = (nlen + 7) // 8 * 8
            p += nlen
            cd = struct.unpack_from(f"<{ncd}I", d, p)
            p += 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            self.filters.append((fid, flags, cd))

    def _fill(self, d):
        ver = d[0]
        if ver in (1, 2):
            defined = d[3]
            if ver == 1 or defined:
                size = struct.unpack_from("<I", d, 4)[0]
                self.fill = d[8:8 + size] if size else None
        elif ver == 3:
            if d[1] & 0x20:
                size = struct.unpack_from("<I", d, 2)[0]
                self.fill = d[6:6 + size] if size else None

    # ---- chunk index (version-1 B-tree, node type 1) ---------------------------------------------------------------
    def _grid(self):
        return tuple(-(-s // c) for s, c in zip(self.shape, self.chunks))

    def _build_index(self):
        grid = self._grid()
        total = int(np.prod(grid))
        addr = np.full(total, UNDEF, np.uint64)
        size = np.zeros(total, np.uint32)
        mask = np.zeros(total, np.uint32)
        nd = self.ndim
        rec = np.dtype([("size", "<u4"), ("mask", "<u4"), ("off", "<u8", (nd + 1,)), ("child", "<u8")])
        strides = np.array([int(np.prod(grid[i + 1:])) for i in range(nd)], np.int64)
        cdims = np.array(self.chunks, np.int64)
        gdims = np.array(grid, np.int64)

        def walk(a):
            h = self.f.read(a, 24)
            if h[:4] != b"TREE" or h[4] != 1:
                raise H5FormatError(f"{self.f.path}: bad chunk B-tree node at {a}")
            level, n = h[5], struct.unpack_from("<H", h, 6)[0]
            if n == 0:
                return
            ents = np.frombuffer(self.f.read(a + 24, n * rec.itemsize), dtype=rec, count=n)
            if level > 0:
                for c in ents["child"]:
                    walk(int(c))
                return
            co = ents["off"][:, :nd].astype(np.int64) // cdims
            ok = (co < gdims).all(axis=1)  # (chunks beyond the current extent remain after a shrink: ignored)
            lin = (co * strides).sum(axis=1)[ok]
            addr[lin] = ents["child"][ok]
            size[lin] = ents["size"][ok]
            mask[lin] = ents["mask"][ok]

        if self.btree != UNDEF:
            walk(self.btree)
        self._index = (addr, size, mask, strides)

    def _chunk_bytes(self, lin):
        addr, size, mask, _ = self._index
        a = int(addr[lin])
        if a == UNDEF:
            return None  # never written: fill value
        raw = self.f.read(a, int(size[lin]))
        m = int(mask[lin])
        for i in range(len(self.filters) - 1, -1, -1):  # the pipeline is undone back to front
            if m >> i & 1:
                continue
            fid, _flags, cd = self.filters[i]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                es = cd[0] if cd else self.itemsize
                n = len(raw) // es
                raw = np.frombuffer(raw, np.uint8, n * es).reshape(es, n).T.tobytes() + raw[n * es:]
            elif fid == 3:
                raw = raw[:-4]  # (checksum not verified)
            else:
                raise H5FormatError(f"{self.f.path}: filter {fid} in the pipeline is not handled (deflate, shuffle, fletcher32 are)")
        return raw

    # ---- reads ------------------------------------------------------------------------------------------------------
    def _fill_value(self):
        if self.fill:
            return np.frombuffer(self.fill, self.dtype, 1)[0]
        return 0

    def read_into(self, index, out, frames=None):
        """out[...] <- dataset[index] (shape self.shape[1:]) ; frames: iterable of indices along axis 1 -- only those are
        filled (out still has the full frame axis), and only the chunks that hold them are read and inflated"""
        if self.dtype is None:
            raise H5FormatError("this dataset's type cannot be read as an array")
        if not 0 <= index < self.shape[0]:
            raise IndexError(index)
        inner = self.shape[1:]
        if tuple(out.shape) != tuple(inner):
            raise ValueError(f"out has shape {out.shape}, a sample is {inner}")
        if self.layout == "contiguous":
            n = int(np.prod(inner))
            if self.data_addr == UNDEF:
                out[...] = self._fill_value()
            else:
                out[...] = np.frombuffer(self.f.read(self.data_addr + index * n * self.itemsize, n * self.itemsize),
                                         self.dtype).reshape(inner)
            return out
        if self.layout != "chunked":
            raise H5FormatError("compact datasets are not handled")
        if self._index is None:
            self._build_index()
        strides = self._index[3]
        ch, grid = self.chunks, self._grid()
        c0, r0 = divmod(index, ch[0])
        want = None
        if frames is not None and self.ndim >= 2:
            want = sorted({int(fr) % inner[0] // ch[1] for fr in frames})
        ranges = [want if (d == 1 and want is not None) else range(grid[d]) for d in range(1, self.ndim)]
        fillv = None
        for coord in np.ndindex(*[len(r) for r in ranges]):
            cc = [ranges[d][coord[d]] for d in range(self.ndim - 1)]
            lin = int(c0 * strides[0] + sum(int(c) * int(s) for c, s in zip(cc, strides[1:])))
            sl_out, sl_in = [], [r0]
            for d, c in enumerate(cc):
                lo = c * ch[d + 1]
                n = min(ch[d + 1], inner[d] - lo)
                sl_out.append(slice(lo, lo + n))
                sl_in.append(slice(0, n))
            raw = self._chunk_bytes(lin)
            if raw is None:
                if fillv is None:
                    fillv = self._fill_value()
                out[tuple(sl_out)] = fillv
            else:
                out[tuple(sl_out)] = np.frombuffer(raw, self.dtype, int(np.prod(ch))).reshape(ch)[tuple(sl_in)]
        return out

    # ---- native gather (csrc/h5gather.c -> libsmaat_io.so): one foreign call per sample, GIL released ----------------
    def native_ok(self):
        """the C helper handles: float32, 4-D chunked datasets with chunk extent 1 along the sample axis, filter pipeline
        = deflate only (what create_datasets.py writes); the library must have been built (csrc/Makefile)"""
        return (_io_lib() is not None and self.layout == "chunked" and self.ndim == 4 and self.dtype == np.float32
                and self.chunks[0] == 1 and [f[0] for f in self.filters] == [1])

    def gather_native(self, index, fmap, dst):
        """dst[fmap[f]] <- dataset[index][f] for every frame f with fmap[f] >= 0 (fmap: int32 [T]); dst: float32 array whose
        last axis is contiguous, shape [>= max(fmap) + 1][H][W].  Only the chunks holding wanted frames are read."""
        import ctypes
        lib = _io_lib()
        if not 0 <= index < self.shape[0]:
            raise IndexError(index)
        if self._index is None:
            self._build_index()
        addr, size, mask, strides = self._index
        plan = self.__dict__.get("_native_plan")
        if plan is None or plan[0] is not fmap:
            t, h, w = self.shape[1:]
            ch, grid = self.chunks, self._grid()
            rows = sorted({f // ch[1] for f in range(t) if fmap[f] >= 0})
            cc = np.array([(a_, b_, c_) for a_ in rows for b_ in range(grid[2]) for c_ in range(grid[3])], np.int64)
            rel = (cc * np.asarray(strides[1:], np.int64)).sum(axis=1)
            origin = np.ascontiguousarray(cc * np.asarray(ch[1:], np.int64), dtype=np.int32)
            plan = (fmap, rel, origin, np.asarray(ch[1:], np.int32), np.asarray(self.shape[1:], np.int32),
                    np.ascontiguousarray(fmap, dtype=np.int32))
            self.__dict__["_native_plan"] = plan
        _, rel, origin, cdim, ext, fm = plan
        lin = rel + int(index) * int(strides[0])
        a = addr[lin]
        if (mask[lin] != 0).any():
            raise H5FormatError("a chunk of this sample skipped a filter (filter mask set): use read_into")
        off = np.where(a == np.uint64(UNDEF), np.int64(-1), (a + np.uint64(self.f.base)).astype(np.int64)).astype(np.int64)
        nb = np.ascontiguousarray(size[lin], dtype=np.int32)
        if dst.dtype != np.float32 or dst.strides[-1] != 4 or dst.shape[-2:] != self.shape[2:]:
            raise ValueError("dst must be float32 [frames][H][W] with a contiguous last axis")
        fill = float(self._fill_value())
        rc = lib.smaat_h5_gather(self.f.fd, len(lin), off.ctypes.data_as(ctypes.c_void_p), nb.ctypes.data_as(ctypes.c_void_p),
                                 origin.ctypes.data_as(ctypes.c_void_p), cdim.ctypes.data_as(ctypes.c_void_p),
                                 ext.ctypes.data_as(ctypes.c_void_p), fm.ctypes.data_as(ctypes.c_void_p),
                                 ctypes.c_void_p(dst.ctypes.data), dst.strides[0] // 4, dst.strides[1] // 4, fill)
        if rc != 0:
            raise H5FormatError(f"{self.f.path}: native chunk gather failed with code {rc} (sample {index}) "
                                "(-3 short read, -4 inflate error / chunk size mismatch)")
        return dst

    def __getitem__(self, index):
        if isinstance(index, (int, np.integer)):
            index = int(index)
            if index < 0:
                index += self.shape[0]
            return self.read_into(index, np.empty(self.shape[1:], self.dtype))
        if isinstance(index, slice):
            idx = range(*index.indices(self.shape[0]))
            out = np.empty((len(idx),) + self.shape[1:], self.dtype)
            for j, i in enumerate(idx):
                self.read_into(i, out[j])
            return out
        raise TypeError("H5Dataset is indexed along its first axis (an int or a slice)")

    def __len__(self):
        return self.shape[0]


_IO = [False]


def _io_lib():
    """libsmaat_io.so (csrc/h5gather.c), or None when it has not been built: the pure-Python path is used then"""
    if _IO[0] is False:
        import ctypes
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsmaat_io.so")
        lib = None
        if os.path.exists(path) and os.environ.get("SMAAT_H5_NATIVE", "1") != "0":
            try:
                lib = ctypes.CDLL(path)
                P, I, L, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
                lib.smaat_h5_gather.argtypes = [I, I, P, P, P, P, P, P, P, L, L, F]
                lib.smaat_h5_gather.restype = I
                if lib.smaat_io_abi_version() != 1:
                    lib = None
            except OSError:
                lib = None
        _IO[0] = lib
    return _IO[0]


# ======================================================================================================================
# minimal writer
# ======================================================================================================================
GROUP_LEAF_K, GROUP_INTERNAL_K, ISTORE_K = 4, 16, 32  # library defaults (superblock 0 stores the first two)


class _Out:
    def __init__(self, fh):
        self.fh = fh
        self.pos = 0

    def alloc(self, n, align=8):
        self.pos = (self.pos + align - 1) // align * align
        a = self.pos
        self.pos += n
        return a

    def put(self, addr, data):
        self.fh.seek(addr)
        self.fh.write(data)


def _msg(mtype, data, flags=0):
    data = data + b"\0" * (-len(data) % 8)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _object_header(msgs):
    body = b"".join(msgs)
    return struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body


def _write_group(o, children):
    """children: {name: object header address}; returns (object header address, B-tree address, heap address)"""
    names = sorted(children, key=lambda s: s.encode())
    if len(names) > 2 * GROUP_LEAF_K:
        raise ValueError("the minimal writer keeps a group in one symbol node (at most 8 links)")
    heap = bytearray(b"\0" * 8)  # offset 0: the empty name (key 0 of the B-tree)
    offs = []
    for nm in names:
        offs.append(len(heap))
        b = nm.encode() + b"\0"
        heap += b + b"\0" * (-len(b) % 8)
    hdata = o.alloc(len(heap))
    o.put(hdata, bytes(heap))
    hh = o.alloc(32)
    o.put(hh, b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), 1, hdata))  # free-list head 1 = H5HL_FREE_NULL (no free block)
    snod = o.alloc(8 + 2 * GROUP_LEAF_K * 40)
    ents = b"".join(struct.pack("<QQII16x", off, children[nm], 0, 0) for nm, off in zip(names, offs))
    o.put(snod, (b"SNOD" + struct.pack("<BBH", 1, 0, len(names)) + ents).ljust(8 + 2 * GROUP_LEAF_K * 40, b"\0"))
    bt = o.alloc(24 + 2 * GROUP_INTERNAL_K * 8 + (2 * GROUP_INTERNAL_K + 1) * 8)
    node = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod, offs[-1] if offs else 0)
    o.put(bt, node.ljust(24 + 2 * GROUP_INTERNAL_K * 8 + (2 * GROUP_INTERNAL_K + 1) * 8, b"\0"))
    oh = _object_header([_msg(MSG_SYMBOL_TABLE, struct.pack("<QQ", bt, hh))])
    a = o.alloc(len(oh))
    o.put(a, oh)
    return a, bt, hh


def _write_chunk_btree(o, entries, nd, end_key):
    """entries: [(offsets tuple (nd + 1), nbytes, address)] sorted; returns the root node address"""
    ksz = 8 + 8 * (nd + 1)
    nsz = 24 + 2 * ISTORE_K * 8 + (2 * ISTORE_K + 1) * ksz
    cap = 2 * ISTORE_K

    def key(off, nbytes):
        return struct.pack("<II", nbytes, 0) + struct.pack(f"<{nd + 1}Q", *off)

    level, items = 0, [(key(off, nb), addr) for off, nb, addr in entries]  # (first key of the subtree, child address)
    last = key(end_key, 0)
    while True:
        groups = [items[i:i + cap] for i in range(0, len(items), cap)] or [[]]
        addrs = [o.alloc(nsz) for _ in groups]
        nxt = []
        for gi, g in enumerate(groups):
            left = addrs[gi - 1] if gi > 0 else UNDEF
            right = addrs[gi + 1] if gi + 1 < len(groups) else UNDEF
            final = groups[gi + 1][0][0] if gi + 1 < len(groups) else last
            body = b"".join(k + struct.pack("<Q", c) for k, c in g) + final
            o.put(addrs[gi], (b"TREE" + struct.pack("<BBHQQ", 1, level, len(g), left, right) + body).ljust(nsz, b"\0"))
            nxt.append((g[0][0] if g else last, addrs[gi]))
        if len(groups) == 1:
            return addrs[0]
        items, level = nxt, level + 1


def _write_dataset(o, arr, chunks, level, threads):
    arr = np.ascontiguousarray(arr, dtype="<f4")
    nd = arr.ndim
    if len(chunks) != nd:
        raise ValueError("chunk rank")
    grid = [-(-s // c) for s, c in zip(arr.shape, chunks)]

    def deflate(cc):
        lo = [c * k for c, k in zip(cc, chunks)]
        blk = np.zeros(chunks, "<f4")  # edge chunks are stored whole (fill value beyond the extent)
        src = arr[tuple(slice(a, a + k) for a, k in zip(lo, chunks))]
        blk[tuple(slice(0, s) for s in src.shape)] = src
        return tuple(lo) + (0,), zlib.compress(blk.tobytes(), level)

    entries = []
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max(1, threads)) as pool:  # zlib releases the GIL; map keeps the chunk order
        for off, z in pool.map(deflate, np.ndindex(*grid), chunksize=16) if threads > 1 else map(deflate, np.ndindex(*grid)):
            a = o.alloc(len(z), align=1)
            o.put(a, z)
            entries.append((off, len(z), a))
    end = (grid[0] * chunks[0],) + (0,) * nd
    bt = _write_chunk_btree(o, entries, nd, end)
    space = struct.pack("<BBB5x", 1, nd, 1) + struct.pack(f"<{nd}Q", *arr.shape) + \
        struct.pack(f"<{nd}Q", UNDEF, *arr.shape[1:])  # maxshape (None, T, H, W)
    # IEEE little-endian float32: class 1 version 1; bit field: mantissa normalisation 2 (implied), sign bit 31
    dtype = struct.pack("<BBBBI", 0x11, 0x20, 31, 0, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
    fill = struct.pack("<BBBB", 2, 3, 2, 0)  # version 2, incremental allocation, fill time "if set", no value defined
    layout = struct.pack("<BBB", 3, 2, nd + 1) + struct.pack("<Q", bt) + struct.pack(f"<{nd + 1}I", *chunks, 4)
    filt = struct.pack("<BB6x", 1, 1) + struct.pack("<HHHH", 1, 0, 1, 1) + struct.pack("<II", level, 0)
    oh = _object_header([_msg(MSG_DATASPACE, space), _msg(MSG_DATATYPE, dtype, 1), _msg(MSG_FILL, fill, 1),
                         _msg(MSG_FILTERS, filt, 1), _msg(MSG_LAYOUT, layout)])
    a = o.alloc(len(oh))
    o.put(a, oh)
    return a


def write_images_h5(path, splits, chunks, level=9, dataset="images", threads=None):
    """{group name: float32 array [samples][T][H][W]} -> an HDF5 file with one chunked, deflate-compressed dataset
    `dataset` per group: the layout of /root/reference/create_datasets.py:31-61 (without the timestamps)."""
    with open(path, "wb") as fh:
        o = _Out(fh)
        sb = o.alloc(96)
        groups = {}
        for name, arr in splits.items():
            d = _write_dataset(o, arr, tuple(int(c) for c in chunks), level,
                               threads if threads is not None else min(32, os.cpu_count() or 1))
            groups[name], _, _ = _write_group(o, {dataset: d})
        root, bt, hh = _write_group(o, groups)
        eof = o.alloc(0)
        head = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, GROUP_LEAF_K, GROUP_INTERNAL_K, 0)
        head += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        head += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", bt, hh)  # root symbol-table entry, cached B-tree / heap
        o.put(sb, head)
        fh.truncate(eof)

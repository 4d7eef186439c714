"""CPU: the C-ABI library builds (hipcc cross-compiles gfx950 without a GPU), loads, and
exports every symbol include/smaat_hip.h declares.  No compute calls."""
import ctypes
import os
import re

import pytest

from smaat_unet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    _lib.build()
    return _lib.LIB_PATH


def test_header_symbols_exported(built):
    hdr = open(os.path.join(ROOT, "include", "smaat_hip.h")).read()
    declared = set(re.findall(r"^int (smaat_[a-z0-9_]+)\(", hdr, flags=re.M))
    assert len(declared) >= 30
    dll = ctypes.CDLL(built)
    for name in declared:
        assert hasattr(dll, name), name
    assert declared == set(_lib.SIGNATURES.keys())


def test_binding_arity_matches_header(built):
    hdr = open(os.path.join(ROOT, "include", "smaat_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for m in re.finditer(r"int (smaat_[a-z0-9_]+)\(([^;]*?)\);", hdr, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("void", "") else len(args.split(","))
        assert n == len(_lib.SIGNATURES[name]), (name, n, len(_lib.SIGNATURES[name]))
        # pointer/scalar kinds line up too
        if n:
            for a, t in zip(args.split(","), _lib.SIGNATURES[name]):
                is_ptr = "*" in a
                assert is_ptr == (t is ctypes.c_void_p), (name, a)


def test_query_entry_points_need_no_gpu(built):
    L = _lib._Lib(built)
    assert L.smaat_abi_version() == 1
    assert L.smaat_pw_num_slots(32, 288, 288, 64) == 32 * 9 * 36
    assert L.smaat_plane_num_slots(2, 324) == 2
    assert L.smaat_wgrad_num_splits(32, 288, 288, 64, 128) >= 1
    assert L.smaat_cbam_pix_blocks(2, 300) == 4


def test_workspace_queries_of_the_round_4_geometries(built):
    """Host-side geometry only (no launch): the depthwise backward's partial rows under plane packing, and the work lists of
    the row-walking kernels."""
    L = _lib._Lib(built)
    # 18 x 18 planes share a wave with the same channel's plane of the next images: still ONE partial row per image (+ 1)
    assert L.smaat_dw3x3_bwd_ws_rows(32, 512, 18, 18) == 32 + 1
    assert L.smaat_dw3x3_bwd_ws_rows(1, 512, 18, 18) == 1 + 1
    # 288 x 288: 72 column groups x bands over several waves per plane
    assert L.smaat_dw3x3_bwd_ws_rows(2, 64, 288, 288) > 2 + 1
    # row-walking kernels: shapes they take / refuse, one statistics slot per (image, band, strip)
    assert L.smaat_dsconv_rows_ok(2, 64, 64, 288, 288) == 1
    assert L.smaat_dsconv_rows_ok(2, 64, 64, 144, 144) == 0   # W % 32 != 0
    assert L.smaat_dsconv_rows_ok(2, 64, 128, 288, 288) == 0  # more than 64 output channels
    assert L.smaat_dsconv_rows_num_slots(32, 288, 288) % (32 * 9) == 0
    assert L.smaat_dsconv_wgrad_split_ok(2, 64, 288, 288) == 1
    ns = L.smaat_dsconv_wgrad_split_num_splits(32, 64, 64, 288, 288)
    assert ns >= 8 and ns % 8 == 0  # contiguous split ranges per XCD

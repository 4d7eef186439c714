"""ctypes binding of libsmaat_hip.so (C ABI declared in include/smaat_hip.h).

There is NO CPU fallback: `get()` raises if the HIP library is missing, and the ops
refuse host tensors.  `build()` compiles the library in-tree with hipcc for gfx950.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMAAT_LIB") or os.path.join(_HERE, "libsmaat_hip.so")  # SMAAT_LIB: experiment builds only
CSRC = os.path.join(_HERE, "csrc")

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_long
_F = ctypes.c_float
_D = ctypes.c_double

# name -> argtypes (restype is always int).  Order mirrors include/smaat_hip.h exactly.
SIGNATURES = {
    "smaat_abi_version": [],
    "smaat_pw_num_slots": [_I, _I, _I, _I],
    "smaat_dsconv_fwd": [_P, _L, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "smaat_pointwise_fwd": [_P, _L, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _P],
    "smaat_wgrad_num_splits": [_I, _I, _I, _I, _I],
    "smaat_dsconv_wgrad_num_splits": [_I, _I, _I, _I, _I],
    "smaat_dsconv_wgrad": [_P, _L, _P, _P, _P, _P, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "smaat_pointwise_wgrad": [_P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _I, _P],
    "smaat_dsconv_wgrad_split_ok": [_I, _I, _I, _I],
    "smaat_dsconv_wgrad_split_num_splits": [_I, _I, _I, _I, _I],
    "smaat_dsconv_wgrad_split": [_P, _L, _P, _P, _P, _P, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "smaat_dsconv_wgrad_split_t": [_P, _I, _L, _P, _P, _P, _P, _P, _I, _L, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "smaat_dsconv_bwd_rows_ok": [_I, _I, _I, _I, _I],
    "smaat_dsconv_bwd_rows_num_rows": [_I, _I, _I, _I],
    "smaat_dsconv_bwd_rows_h": [_P, _L, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _L, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "smaat_dw3x3_bwd_ws_rows": [_I, _I, _I, _I],
    "smaat_dw3x3_bwd": [_P, _L, _P, _L, _P, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "smaat_dw3x3_strip_ok": [_I, _I, _I],
    "smaat_dw3x3_bwd_bnred": [_P, _L, _P, _P, _P, _L, _P, _P, _L, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "smaat_bn_finalize": [_P, _I, _I, _D, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P],
    "smaat_bn_eval_coefs": [_P, _P, _P, _P, _F, _I, _P, _P],
    "smaat_affine_act": [_P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    "smaat_plane_num_slots": [_I, _I],
    "smaat_bn_bwd_reduce": [_P, _L, _P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "smaat_bn_bwd_finalize": [_P, _I, _I, _D, _P, _P, _P, _P, _P, _P],
    "smaat_bn_bwd_apply": [_P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    "smaat_outconv1_fwd": [_P, _L, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P],
    "smaat_bn_bwd_reduce_head": [_P, _L, _P, _P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "smaat_bn_bwd_apply_head": [_P, _L, _P, _P, _L, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P],
    "smaat_reduce_rows": [_P, _I, _L, _P, _F, _P],
    "smaat_channel_sum": [_P, _L, _I, _I, _I, _P, _P, _P],
    "smaat_copy_planes": [_P, _L, _P, _L, _I, _L, _I, _P],
    "smaat_maxpool2_fwd": [_P, _L, _P, _L, _I, _I, _I, _I, _P],
    "smaat_maxpool2_bwd": [_P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _P],
    "smaat_upsample2x_fwd": [_P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "smaat_upsample2x_bwd": [_P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "smaat_pixel_shuffle2_fwd": [_P, _L, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "smaat_pixel_shuffle2_bwd": [_P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "smaat_cbam_spconv_blocks": [_I, _I, _I],
    "smaat_cbam_pix_blocks": [_I, _I],
    "smaat_cbam_chpool": [_P, _L, _I, _I, _I, _P, _P, _P, _P],
    "smaat_cbam_chpool_act": [_P, _L, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P, _P],
    "smaat_cbam_mlp": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "smaat_cbam_sppool": [_P, _L, _P, _I, _I, _I, _P, _P],
    "smaat_cbam_spconv": [_P, _P, _I, _I, _I, _I, _P, _P, _P],
    "smaat_cbam_gate": [_P, _P, _P, _L, _P, _P],
    "smaat_cbam_apply": [_P, _L, _P, _P, _P, _L, _I, _I, _I, _P],
    "smaat_cbam_eval_pool": [_P, _L, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "smaat_cbam_eval_apply": [_P, _L, _P, _P, _P, _I, _P, _P, _P, _P, _F, _I, _I, _I, _I, _P, _L, _P, _L, _P],
    "smaat_cbam_bwd_gate": [_P, _L, _P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "smaat_cbam_bwd_spconv": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "smaat_cbam_bwd_main": [_P, _L, _P, _L, _P, _P, _P, _P, _I, _I, _I, _P, _L, _P, _P],
    "smaat_cbam_bwd_mlp": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "smaat_cbam_bwd_final": [_P, _L, _P, _P, _P, _I, _I, _I, _P],
    "smaat_cbam_bwd_final_pool": [_P, _L, _P, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _P],
    "smaat_split_enabled": [],
    "smaat_split_mode": [],
    "smaat_set_split_mode": [_I],
    "smaat_split_planes": [_P, _I, _I, _P, _P],
    "smaat_split_planes_t": [_P, _I, _I, _P, _P],
    "smaat_weight_planes_multi": [_P, _I, _I, _P],
    "smaat_pw_split_num_slots": [_I, _I, _I],
    "smaat_dw3x3_fwd": [_P, _L, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P],
    "smaat_pointwise_fwd_split": [_P, _L, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _P],
    "smaat_dsconv_split_num_slots": [_I, _I, _I],
    "smaat_dsconv_fwd_split": [_P, _L, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "smaat_dsconv_rows_ok": [_I, _I, _I, _I, _I],
    "smaat_dsconv_rows_num_slots": [_I, _I, _I],
    "smaat_dsconv_fwd_rows": [_P, _I, _L, _P, _P, _P, _P, _P, _P, _P, _I, _L, _P, _I, _I, _I, _I, _I, _I, _P],
    "smaat_dsconv_fwd_act": [_P, _L, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _P],
    "smaat_dsconv_fwd_split_act": [_P, _L, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _P],
    "smaat_pointwise_fwd_split_act": [_P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "smaat_pointwise_splitk_ws_floats": [_I, _I, _I, _I, _I],
    "smaat_pointwise_fwd_split_act_k": [_P, _L, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _P],
    "smaat_pointwise_splitk_slices": [_I, _I, _I, _I, _I, _I],
    "smaat_pointwise_fwd_split_k": [_P, _L, _P, _P, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    # ---- two-term fp16 split (three fp16 MFMAs per product; operand maxima from the producing kernels) ----
    "smaat_dw3x3_fwd_amax": [_P, _L, _P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _P],
    "smaat_bn_bwd_apply_amax": [_P, _L, _P, _P, _L, _P, _P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _P],
    "smaat_split_planes_h_bytes": [_I, _I],
    "smaat_split_planes_h_pieces": [_I, _I],
    "smaat_split_planes_h": [_P, _I, _I, _P, _I, _P],
    "smaat_weight_planes_multi_h": [_P, _I, _I, _I, _P],
    "smaat_pointwise_fwd_split_h": [_P, _L, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _P],
    "smaat_pointwise_fwd_split_k_h": [_P, _L, _P, _P, _P, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "smaat_pointwise_wgrad_h": [_P, _L, _P, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "smaat_dsconv_fwd_rows_amax": [_P, _L, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "smaat_dsconv_wgrad_split_h": [_P, _L, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "smaat_dsconv_fwd_rows_h": [_P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "smaat_pointwise_fwd_h2_proto": [_P, _L, _L, _P, _L, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "smaat_adam_max_tensors": [],
    "smaat_adam_block_elems": [],
    "smaat_adam_step": [_P, _P, _P, _P, _I, _I, _D, _D, _D, _D, _D, _D, _I, _P],
    "smaat_cbam_apply_amax": [_P, _L, _P, _P, _P, _L, _P, _I, _I, _I, _P],
    "smaat_upsample2x_fwd_amax": [_P, _L, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    # ---- mixed precision (bf16 activation storage) ----
    "smaat_bf16_planes": [_P, _I, _I, _P, _I, _P],
    "smaat_pointwise_fwd_bf16": [_P, _L, _P, _P, _P, _L, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "smaat_pointwise_wgrad_bf16": [_P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _I, _P],
    "smaat_dw3x3_fwd_t": [_P, _I, _L, _P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _I, _I, _P],
    "smaat_dw3x3_bwd_t": [_P, _I, _L, _P, _P, _P, _I, _L, _P, _P, _I, _L, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "smaat_affine_act_t": [_P, _I, _L, _P, _P, _P, _I, _L, _I, _I, _I, _I, _P],
    "smaat_bn_bwd_reduce_t": [_P, _I, _L, _P, _I, _L, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P],
    "smaat_bn_bwd_apply_t": [_P, _I, _L, _P, _I, _L, _P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _I, _P, _P],
    "smaat_outconv1_fwd_t": [_P, _I, _L, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P],
    "smaat_channel_sum_t": [_P, _I, _L, _I, _I, _I, _P, _P, _P],
    "smaat_maxpool2_fwd_t": [_P, _L, _P, _L, _I, _I, _I, _I, _I, _P],
    "smaat_maxpool2_bwd_t": [_P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "smaat_upsample2x_fwd_t": [_P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "smaat_upsample2x_bwd_t": [_P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "smaat_cbam_chpool_t": [_P, _L, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P, _I, _P],
    "smaat_dwconv_fwd_any": [_P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "smaat_dwconv_bwd_any": [_P, _L, _P, _L, _P, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "smaat_cbam_chpool_pool_t": [_P, _L, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _P, _P, _P, _I, _P],
    "smaat_cbam_sppool_t": [_P, _L, _P, _I, _I, _I, _P, _I, _P],
    "smaat_cbam_apply_t": [_P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    "smaat_cbam_bwd_gate_t": [_P, _L, _P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P],
    "smaat_cbam_bwd_main_t": [_P, _L, _P, _L, _P, _P, _P, _P, _I, _I, _I, _P, _L, _P, _I, _P],
    "smaat_cbam_bwd_final_t": [_P, _L, _P, _P, _P, _I, _I, _I, _I, _P],
    "smaat_cbam_bwd_final_pool_t": [_P, _L, _P, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _I, _P],
    "smaat_cbam_bwd3_ok": [_P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I],
    "smaat_cbam_bwd_gate_ds_t": [_P, _L, _P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _P],
    "smaat_cbam_sppool_idx_t": [_P, _L, _P, _I, _I, _I, _P, _P, _I, _P],
    "smaat_cbam_bwd_ds2_t": [_P, _L, _P, _P, _I, _I, _I, _P, _I, _P],
    "smaat_cbam_bwd_apply_t": [_P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P, _L, _I, _P],
    "smaat_precip_metrics_ws_bytes": [_L],
    "smaat_precip_metrics_update": [_P, _P, _L, _I, _F, _F, _I, _P, _P, _P, _P],
}

_instance = None
# Set to True ONLY by the CPU test-suite when it injects its emulation backend
# (tests/emu_backend.py).  Product code never touches it.
_ALLOW_HOST_POINTERS = False


class SmaatHipError(RuntimeError):
    pass


class _Lib:
    def __init__(self, path):
        self._dll = ctypes.CDLL(path)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(self._dll, name)  # AttributeError if the symbol is missing
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
            setattr(self, name, fn)
        if self.smaat_abi_version() != 1:
            raise SmaatHipError("libsmaat_hip.so ABI version mismatch")


def build(force=False, verbose=False):
    """Compile every .hip source for gfx950 into smaat_unet_amd/libsmaat_hip.so (in-tree)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=not verbose)
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise SmaatHipError("hipcc build failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout[-2000:])
    return LIB_PATH


def get():
    """The loaded library; raises loudly if it has not been built (no fallback)."""
    global _instance
    if _instance is None:
        if not os.path.exists(LIB_PATH):
            raise SmaatHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). smaat_unet_amd has no CPU fallback.")
        _instance = _Lib(LIB_PATH)
    return _instance


def check(rc, what):
    if rc != 0:
        raise SmaatHipError(f"{what} failed with code {rc}")


# --------------------------------------------------------------------------------------
# optional per-entry-point timing (HIP events on the launch stream) used by bench.py
# --------------------------------------------------------------------------------------
def _w_dsconv_fwd(a):
    n, cin, kpl, cout, h, w = a[12:18]
    return 2.0 * n * cin * kpl * cout * h * w, 4.0 * n * (cin + cout) * h * w


def _w_pointwise_fwd(a):
    n, cin, m, h, w = a[7:12]
    return 2.0 * n * cin * m * h * w, 4.0 * n * (cin + m) * h * w


def _w_dsconv_wgrad(a):
    n, cin, kpl, cout, h, w = a[10:16]
    return 2.0 * n * cin * kpl * cout * h * w, 4.0 * n * (cin + cout) * h * w


def _w_pointwise_wgrad(a):
    n, cin, m, h, w = a[6:11]
    return 2.0 * n * cin * m * h * w, 4.0 * n * (cin + m) * h * w


def _w_dw_bwd(a):
    n, cin, kpl, h, w = a[10:15]
    return 38.0 * n * cin * kpl * h * w, 4.0 * n * (cin * kpl + 2 * cin) * h * w


def _w_pw_split(a):
    n, cin, m, h, w = a[7:12]
    return 2.0 * n * cin * m * h * w, 4.0 * n * (cin + m) * h * w


def _es(dt):
    return 2.0 if dt == 1 else 4.0


WORK_MODELS = {
    # mixed precision: algorithmic bytes with the element sizes actually passed
    "smaat_pointwise_fwd_bf16": lambda a: (2.0 * a[8] * a[9] * a[10] * a[11] * a[12],
                                           a[8] * (2.0 * a[9] + _es(a[6]) * a[10]) * a[11] * a[12]),
    "smaat_pointwise_wgrad_bf16": lambda a: (2.0 * a[6] * a[7] * a[8] * a[9] * a[10], 2.0 * a[6] * (a[7] + a[8]) * a[9] * a[10]),
    "smaat_dw3x3_fwd_t": lambda a: (18.0 * a[10] * a[11] * a[12] * a[13] * a[14],
                                    a[10] * a[11] * (_es(a[1]) + _es(a[8]) * a[12]) * a[13] * a[14]),
    "smaat_dw3x3_bwd_t": lambda a: (38.0 * a[18] * a[19] * a[20] * a[21] * a[22],
                                    a[18] * a[19] * (_es(a[6]) * a[20] + _es(a[1]) + (_es(a[10]) if a[9] else 0.0)) * a[21] * a[22]),
    "smaat_affine_act_t": lambda a: (2.0 * a[8] * a[9] * a[10], (_es(a[1]) + _es(a[6])) * a[8] * a[9] * a[10]),
    "smaat_bn_bwd_reduce_t": lambda a: (6.0 * a[11] * a[12] * a[13],
                                        (_es(a[4]) * a[12] + _es(a[1]) * (1 if a[15] else a[12])) * a[11] * a[13]),
    "smaat_bn_bwd_apply_t": lambda a: (8.0 * a[14] * a[15] * a[16],
                                       ((_es(a[4]) + _es(a[12])) * a[15] + _es(a[1]) * (1 if a[18] else a[15])) * a[14] * a[16]),
    "smaat_dw3x3_bwd_bnred": lambda a: (38.0 * a[15] * a[16] * a[17] * a[18] * a[19],
                                        4.0 * a[15] * (a[16] * a[17] + 2 * a[16]) * a[18] * a[19]),
    "smaat_pointwise_fwd_split": _w_pw_split,
    "smaat_pointwise_fwd_split_h": lambda a: (2.0 * a[8] * a[9] * a[10] * a[11] * a[12], 4.0 * a[8] * (a[9] + a[10]) * a[11] * a[12]),
    "smaat_pointwise_fwd_split_k_h": lambda a: (2.0 * a[10] * a[11] * a[12] * a[13] * a[14],
                                                4.0 * a[10] * (a[11] + a[12]) * a[13] * a[14]),
    "smaat_pointwise_wgrad_h": lambda a: (2.0 * a[8] * a[9] * a[10] * a[11] * a[12], 4.0 * a[8] * (a[9] + a[10]) * a[11] * a[12]),
    "smaat_dw3x3_fwd_amax": lambda a: (18.0 * a[9] * a[10] * a[11] * a[12] * a[13],
                                       4.0 * a[9] * a[10] * (1 + a[11]) * a[12] * a[13]),
    "smaat_bn_bwd_apply_amax": lambda a: (8.0 * a[13] * a[14] * a[15], (12.0 if not a[2] else 8.0) * a[13] * a[14] * a[15]),
    "smaat_dw3x3_fwd": lambda a: (18.0 * a[8] * a[9] * a[10] * a[11] * a[12],
                                  4.0 * a[8] * a[9] * (1 + a[10]) * a[11] * a[12]),
    "smaat_dsconv_fwd": _w_dsconv_fwd,
    "smaat_dsconv_fwd_split": _w_dsconv_fwd,
    "smaat_dsconv_fwd_rows": lambda a: (2.0 * a[13] * a[14] * a[15] * a[16] * a[17] * a[18],
                                        a[13] * (_es(a[1]) * a[14] + _es(a[10]) * a[16]) * a[17] * a[18]),
    "smaat_pointwise_fwd": _w_pointwise_fwd,
    "smaat_dsconv_wgrad": _w_dsconv_wgrad,
    "smaat_dsconv_wgrad_split": _w_dsconv_wgrad,
    "smaat_dsconv_wgrad_split_h": lambda a: (2.0 * a[12] * a[13] * a[14] * a[15] * a[16] * a[17],
                                             4.0 * a[12] * (a[13] + a[15]) * a[16] * a[17]),
    "smaat_dsconv_bwd_rows_h": lambda a: (2.0 * a[17] * a[18] * a[19] * a[20] * a[21] * a[22],
                                          4.0 * a[17] * (a[20] + 2 * a[18]) * a[21] * a[22]),  # dz + x read, dx written
    "smaat_dsconv_fwd_rows_amax": lambda a: (2.0 * a[12] * a[13] * a[14] * a[15] * a[16] * a[17],
                                             4.0 * a[12] * (a[13] + a[15]) * a[16] * a[17]),
    "smaat_dsconv_fwd_rows_h": lambda a: (2.0 * a[18] * a[19] * a[20] * a[21] * a[22] * a[23],
                                          4.0 * a[18] * (a[19] + a[21]) * a[22] * a[23]),
    "smaat_dsconv_wgrad_split_t": lambda a: (2.0 * a[12] * a[13] * a[14] * a[15] * a[16] * a[17],
                                             a[12] * (_es(a[1]) * a[13] + _es(a[8]) * a[15]) * a[16] * a[17]),
    "smaat_pointwise_wgrad": _w_pointwise_wgrad,
    "smaat_dw3x3_bwd": _w_dw_bwd,
    "smaat_affine_act": lambda a: (2.0 * a[6] * a[7] * a[8], 8.0 * a[6] * a[7] * a[8]),
    "smaat_bn_bwd_reduce": lambda a: (6.0 * a[9] * a[10] * a[11], 8.0 * a[9] * a[10] * a[11]),
    "smaat_bn_bwd_apply": lambda a: (8.0 * a[11] * a[12] * a[13], 12.0 * a[11] * a[12] * a[13]),
}


class Profiler:
    """Wraps every entry point of the loaded library with a pair of events on the current
    stream.  `summary()` synchronises and returns name -> dict(calls, ms, flop, bytes)."""

    def __init__(self):
        import torch
        self.torch = torch
        self.lib = get()
        self.records = []
        self._orig = {}
        for name in SIGNATURES:
            fn = getattr(self.lib, name)
            if name.endswith(("_slots", "_splits", "_blocks", "_version", "_ws_rows", "_enabled", "_mode", "_bytes", "_ok",
                              "_slices", "_floats", "_pieces")):
                continue
            self._orig[name] = fn
            setattr(self.lib, name, self._wrap(name, fn))

    def _wrap(self, name, fn):
        torch = self.torch

        def wrapped(*args):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            self.records.append((name, e0, e1, args))
            return rc
        return wrapped

    def close(self):
        for name, fn in self._orig.items():
            setattr(self.lib, name, fn)

    def summary(self):
        self.torch.cuda.synchronize()
        out = {}
        for name, e0, e1, args in self.records:
            d = out.setdefault(name, dict(calls=0, ms=0.0, flop=0.0, bytes=0.0))
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            wm = WORK_MODELS.get(name)
            if wm is not None:
                f, b = wm(args)
                d["flop"] += f
                d["bytes"] += b
        self.records = []
        return out

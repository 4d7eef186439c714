#!/usr/bin/env python
"""Where does an iteration of the row-walking fused forward (csrc/dsrows.hip) go?  Times it on one layer with parts of the
iteration compiled out (DSR_DBG bits: 1 no LDS reads + MFMA, 2 no B-image writes, 4 no depthwise math, 8 no global loads,
16 no barrier, 32 no output stores, 64 no BatchNorm partials; results are wrong, only the time means something).  One
experiment library per setting (smaat_unet_amd/exp/libsmaat_hip_dsrdbg<bits>.so), one child process per library."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    sys.path.insert(0, ROOT)
    from smaat_unet_amd import _lib
    L = _lib.get()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    cin, cout, h = int(os.environ.get("DG_CIN", 64)), 64, 288
    k, p = cin * 2, h * h
    w_dw, b_dw = torch.randn(k, 9, device=dev) * 0.3, torch.randn(k, device=dev) * 0.1
    w_pw, b_pw = torch.randn(cout, k, device=dev) * 0.1, torch.randn(cout, device=dev) * 0.1
    out = []
    for bf in (1, 0):
        N = 64 if bf else 32
        dt = torch.bfloat16 if bf else torch.float32
        x = torch.randn(N, cin, h, h, device=dev).to(dt)
        z = torch.empty(N, cout, h, h, device=dev, dtype=dt)
        part = torch.empty(3, L.smaat_dsconv_rows_num_slots(N, h, h), cout, device=dev)
        if bf:
            pl = torch.empty(((k + 31) // 32 * 2, cout, 16), dtype=torch.int16, device=dev)
            assert L.smaat_bf16_planes(w_pw.data_ptr(), cout, k, pl.data_ptr(), 0, st) == 0
        else:
            pl = torch.empty(3 * cout * ((k + 15) // 16 * 16), dtype=torch.int16, device=dev)
            assert L.smaat_split_planes(w_pw.data_ptr(), cout, k, pl.data_ptr(), st) == 0

        def run():
            rc = L.smaat_dsconv_fwd_rows(x.data_ptr(), bf, cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(), pl.data_ptr(),
                                         b_pw.data_ptr(), z.data_ptr(), bf, cout * p, part.data_ptr(), N, cin, 2, cout, h, h, st)
            assert rc == 0, rc
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 10)
        del x, z, part
    print(f"dbg={int(os.environ.get('DSR_DBG_TAG', '0')):3d}  bf16 B=64 {out[0]:7.3f} ms   f32 B=32 {out[1]:7.3f} ms", flush=True)


if __name__ == "__main__":
    if os.environ.get("DSR_CHILD") == "1":
        child()
    else:
        for dbg in [int(v) for v in os.environ.get("DSR_DBGS", "0,1,2,4,8,16,32,64,7,40,96,103,127,0").split(",")]:
            env = dict(os.environ, DSR_CHILD="1", DSR_DBG_TAG=str(dbg))
            if dbg:
                env["SMAAT_LIB"] = os.path.join(ROOT, "smaat_unet_amd", "exp", f"libsmaat_hip_dsrdbg{dbg}.so")
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=300)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("dbg=")]
            print(lines[-1] if lines else f"dbg={dbg} FAILED rc={r.returncode} {r.stderr[-300:]}", flush=True)

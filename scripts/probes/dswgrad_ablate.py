#!/usr/bin/env python
"""Where does an iteration of the recompute weight gradient go?  Times csrc/dswgrad.hip on one layer with parts of the
iteration compiled out (DWG_DBG bits: 1 no LDS reads + MFMA, 2 no LDS writes, 4 no depthwise math, 8 no global loads,
16 no barrier; the results are wrong, only the time means something).  One experiment library per setting
(smaat_unet_amd/exp/libsmaat_hip_dwgdbg<bits>.so: `hipcc -DDWG_DBG=<bits> -c dswgrad.hip`, linked with the other objects),
one child process per library (SMAAT_LIB).  The switches are compile-time on purpose: as run-time flags they changed the
code of the normal path."""
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
    dw = torch.empty(cout, k, device=dev)
    out = []
    for bf in (1, 0):
        N = 64 if bf else 32
        dt = torch.bfloat16 if bf else torch.float32
        x = torch.randn(N, cin, h, h, device=dev).to(dt)
        dz = torch.randn(N, cout, h, h, device=dev).to(dt)
        ws = torch.empty(L.smaat_dsconv_wgrad_split_num_splits(N, cin, cout, h, h), cout, k, device=dev)

        def run():
            if bf:
                rc = L.smaat_dsconv_wgrad_split_t(x.data_ptr(), 1, cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(), dz.data_ptr(),
                                                  1, cout * p, ws.data_ptr(), dw.data_ptr(), N, cin, 2, cout, h, h, st)
            else:
                rc = L.smaat_dsconv_wgrad_split(x.data_ptr(), cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(), dz.data_ptr(),
                                                cout * p, ws.data_ptr(), dw.data_ptr(), N, cin, 2, cout, h, h, st)
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
        del x, dz, ws
    print(f"dbg={int(os.environ.get('DWG_DBG_TAG', '0')):2d}  bf16 B=64 {out[0]:7.3f} ms   f32 B=32 {out[1]:7.3f} ms", flush=True)


if __name__ == "__main__":
    if os.environ.get("DWG_CHILD") == "1":
        child()
    else:
        for dbg in [int(v) for v in os.environ.get("DWG_DBGS", "0,1,2,4,8,16,7,12,15,31,0").split(",")]:
            env = dict(os.environ, DWG_CHILD="1", DWG_DBG_TAG=str(dbg))
            if dbg:
                env["SMAAT_LIB"] = os.path.join(ROOT, "smaat_unet_amd", "exp", f"libsmaat_hip_dwgdbg{dbg}.so")
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=300)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("dbg=")]
            print(lines[-1] if lines else f"dbg={dbg} FAILED rc={r.returncode} {r.stderr[-300:]}", flush=True)

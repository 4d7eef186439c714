#!/usr/bin/env python
"""Round 6: where does a step of k_dsconv_bwd_rows go?  Times smaat_dsconv_bwd_rows_h on inc.1 / up4.0 of BASELINE configs[1]
(batch 32, 288 x 288) with the normal library and with the ablation builds (-DDBW_DBG=<bits>, smaat_unet_amd/exp/), beside the two
kernels it replaces.  One subprocess per library (SMAAT_LIB is read at import)."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from smaat_unet_amd import _lib
    from tests.test_gpu_kernels import P, stream
    from tests.test_gpu_f16_split import _publish, _h_image
    L, dev = _lib.get(), torch.device("cuda:0")
    out = {}
    for name, (N, Cin, Cout, H, W, aff) in {"inc.1": (32, 64, 64, 288, 288, True), "up4.0": (32, 128, 64, 288, 288, False)}.items():
        K = 2 * Cin
        g = torch.Generator(device="cpu").manual_seed(1)
        x = torch.randn(N, Cin, H, W, generator=g).to(dev)
        dz = (torch.randn(N, Cout, H, W, generator=g) * 1e-3).to(dev)
        w_pw, w_dw = (torch.randn(Cout, K, generator=g) * 0.2).to(dev), (torch.randn(K, 9, generator=g) * 0.3).to(dev)
        adz, pl_t = _publish(dz), _h_image(L, dev, w_pw, transposed=True)
        sc = sh = mean = invstd = rp = None
        rows = L.smaat_dsconv_bwd_rows_num_rows(N, Cin, H, W)
        if aff:
            sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.3
            mean, invstd = torch.randn(Cin, device=dev) * 0.1, torch.rand(Cin, device=dev) + 0.5
            rp = torch.empty(2, rows, Cin, device=dev)
        ws, dx = torch.empty(rows, K, 10, device=dev), torch.empty(N, Cin, H, W, device=dev)
        dw, db = torch.empty(K, 9, device=dev), torch.empty(K, device=dev)

        def fused():
            assert L.smaat_dsconv_bwd_rows_h(P(x), Cin * H * W, P(sc), P(sh), P(mean), P(invstd), P(dz), Cout * H * W, P(adz), P(pl_t), P(w_dw),
                                             P(dx), Cin * H * W, P(ws), P(dw), P(db), P(rp), N, Cin, 2, Cout, H, W, stream(dev)) == 0

        def timeit(fn, n=20):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return round(e0.elapsed_time(e1) / n, 4)
        r = {"fused_ms": timeit(fused)}
        if os.environ.get("DBW_REF") == "1":
            dy = torch.empty(N, K, H, W, device=dev)
            rows1 = L.smaat_dw3x3_bwd_ws_rows(N, Cin, H, W)
            ws1 = torch.empty(rows1, K, 10, device=dev)
            rp1 = torch.empty(2, rows1 - 1, Cin, device=dev)

            def dgrad():
                assert L.smaat_pointwise_fwd_split_h(P(dz), Cout * H * W, P(adz), P(pl_t), None, P(dy), K * H * W, None, N, Cout, K, H, W, stream(dev)) == 0

            def dwb():
                if aff:
                    assert L.smaat_dw3x3_bwd_bnred(P(x), Cin * H * W, P(sc), P(sh), P(dy), K * H * W, P(w_dw), P(dx), Cin * H * W, P(ws1), P(dw), P(db),
                                                   P(mean), P(invstd), P(rp1), N, Cin, 2, H, W, stream(dev)) == 0
                else:
                    assert L.smaat_dw3x3_bwd(P(x), Cin * H * W, P(dy), K * H * W, P(w_dw), P(dx), Cin * H * W, P(ws1), P(dw), P(db), N, Cin, 2, H, W, stream(dev)) == 0
            r["dgrad_ms"], r["dw_bwd_ms"] = timeit(dgrad), timeit(dwb)
        out[name] = r
    print("RESULT " + json.dumps(out))
    sys.exit(0)
libs = [None] + sorted(glob.glob(os.path.join(ROOT, "smaat_unet_amd", "exp", "libsmaat_hip_dbwdbg*.so")), key=lambda p: int(p.split("dbwdbg")[1][:-3]))
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["SMAAT_LIB"] = lib
    else:
        env["DBW_REF"] = "1"
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
    line = next((l for l in p.stdout.splitlines() if l.startswith("RESULT ")), None)
    tag = "normal build" if not lib else "DBW_DBG=" + lib.split("dbwdbg")[1][:-3]
    print(f"{tag:16s}", line[7:] if line else "FAILED " + p.stderr[-300:])

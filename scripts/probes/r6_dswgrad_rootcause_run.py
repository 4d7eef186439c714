#!/usr/bin/env python
"""Round 6, VERDICT r5 weak #1: which change cures the call-to-call nondeterminism of k_dsconv_wgrad_split<NT=2, AFF, scalar>?
For every experiment library of scripts/probes/r6_dswgrad_rootcause_build.sh (a subprocess each: SMAAT_LIB is read at import):
the AFF recompute weight gradient on the two-term fp16 split, REPS calls on identical inputs, rel-L2 of dW against fp64 and
whether the calls are bit-identical.  Shapes: the round-5 probe (2 x 64ch x 32x32 -> 4 one-item workgroups, walks of 18 rows)
and a longer walk (4 x 64ch x 96x64)."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REPS = int(os.environ.get("RC_REPS", "24"))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from smaat_unet_amd import _lib
    from tests.test_gpu_kernels import P, T, rel, rnd, stream
    from tests.test_gpu_f16_split import _publish
    from oracle import smaat_oracle as O
    L, dev = _lib.get(), torch.device("cuda:0")
    out = {}
    for (N, Cin, Cout, H, W) in ((2, 64, 64, 32, 32), (4, 64, 64, 96, 64)):
        K = 2 * Cin
        xn = rnd(1, N, Cin, H, W)
        w_dwn, b_dwn = rnd(2, K, 9, scale=0.3), rnd(3, K, scale=0.3)
        scn = np.random.default_rng(6).uniform(0.5, 1.5, Cin).astype(np.float32)
        shn = rnd(7, Cin, scale=0.3)
        x, w_dw, b_dw, sc, sh = T(xn, dev), T(w_dwn, dev), T(b_dwn, dev), T(scn, dev), T(shn, dev)
        xa = np.maximum(xn * scn[None, :, None, None] + shn[None, :, None, None], 0)
        y64 = O.dw3x3_fwd(xa.astype(np.float64), w_dwn.astype(np.float64).reshape(K, 1, 3, 3), b_dwn.astype(np.float64), 2)
        dzn = rnd(8, N, Cout, H, W) * 1e-3
        dz = T(dzn, dev)
        ref = np.einsum("nmp,nkp->mk", dzn.astype(np.float64).reshape(N, Cout, -1), y64.reshape(N, K, -1))
        ay = _publish(torch.from_numpy(y64.astype(np.float32))).to(dev)
        adz = _publish(dz)
        ws = torch.empty((L.smaat_dsconv_wgrad_split_num_splits(N, Cin, Cout, H, W), Cout, K), device=dev)
        errs, outs = [], []
        for _ in range(REPS):
            dw = torch.empty((Cout, K), device=dev)
            assert L.smaat_dsconv_wgrad_split_h(P(x), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(ay), P(dz), Cout * H * W, P(adz), P(ws),
                                                P(dw), N, Cin, 2, Cout, H, W, stream(dev)) == 0
            torch.cuda.synchronize()
            outs.append(dw.cpu())
            errs.append(float(rel(outs[-1].numpy(), ref)))
        out[f"{N}x{Cin}x{H}x{W}"] = dict(rel_min=min(errs), rel_max=max(errs), distinct=len({o.numpy().tobytes() for o in outs}),
                                         wrong_calls=sum(e > 1e-5 for e in errs), calls=REPS)
    print("RESULT " + json.dumps(out))
    sys.exit(0)

libs = sorted(glob.glob(os.path.join(ROOT, "smaat_unet_amd", "exp", "libsmaat_hip_rc_*.so")))
print(f"{len(libs)} experiment libraries, {REPS} calls each")
for lib in libs:
    tag = os.path.basename(lib)[len("libsmaat_hip_rc_"):-3]
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, SMAAT_LIB=lib), capture_output=True, text=True, timeout=600)
    line = next((l for l in p.stdout.splitlines() if l.startswith("RESULT ")), None)
    if line is None:
        print(f"{tag:24s} FAILED rc={p.returncode} {p.stderr[-400:]}")
        continue
    for shape, r in json.loads(line[7:]).items():
        print(f"{tag:24s} {shape:14s} rel vs fp64 {r['rel_min']:.3e} .. {r['rel_max']:.3e}   wrong calls {r['wrong_calls']:2d}/{r['calls']}   distinct results {r['distinct']}")

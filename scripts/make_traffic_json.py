#!/usr/bin/env python
"""profiles/<round>/<tag>/rocprof_summary.txt (scripts/prof_round.sh) -> profiles/hbm_traffic.json:
HBM bytes per launch of each kernel from the PMC passes, (FETCH_SIZE x 2 [gfx950 correction, see
MI355X_MICROARCH.md] + WRITE_SIZE) / dispatches.  bench.py quotes these as `roofline.traffic`."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r1", "r1r", "rocprof_summary.txt")
fetch, write = {}, {}
for line in open(src):
    m = re.match(r"(.+?)\s+FETCH_SIZE ([\d.]+) MB over (\d+) disp \(x2 corrected ([\d.]+) MB\)", line)
    if m:
        fetch[m.group(1).strip()] = (float(m.group(4)) * 1e6, int(m.group(3)))
    m = re.match(r"(.+?)\s+WRITE_SIZE ([\d.]+) MB over (\d+) disp", line)
    if m:
        write[m.group(1).strip()] = (float(m.group(2)) * 1e6, int(m.group(3)))
sys.path.insert(0, ROOT)
from bench import csrc_sha16  # noqa: E402
sha = sys.argv[2] if len(sys.argv) > 2 else csrc_sha16()  # the build the counters were collected on
out = {"source": os.path.relpath(src, ROOT), "csrc_sha16": sha, "unit": "bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) / dispatches",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, (0.0, 0)), write.get(k, (0.0, 0))
    nd = max(f[1], w[1])
    if nd:
        name = re.sub(r"^void\s+", "", k)
        out["kernels"][name] = {"bytes_per_launch": round((f[0] + w[0]) / nd), "dispatches": nd,
                                "fetch_bytes": round(f[0]), "write_bytes": round(w[0])}
dst = os.path.join(ROOT, "profiles", "hbm_traffic.json")
json.dump(out, open(dst, "w"), indent=1)
print(dst, len(out["kernels"]), "kernels")

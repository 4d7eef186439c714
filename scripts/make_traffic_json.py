#!/usr/bin/env python
"""profiles/<round>/<tag>/rocprof_summary.txt (scripts/prof_round.sh) -> profiles/hbm_traffic.json:
HBM bytes per launch of each kernel from the PMC passes, (FETCH_SIZE x 2 [gfx950 correction, see
MI355X_MICROARCH.md] + WRITE_SIZE) / dispatches.  bench.py quotes these as `roofline.traffic`."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# usage: make_traffic_json.py summary.txt [more_summaries.txt ...]   (e.g. the f32 run and the mixed-precision run of one
# build: their kernels have different names -- element-type template arguments -- and share one JSON)
srcs = [a for a in sys.argv[1:] if not a.startswith("sha=")] or [os.path.join(ROOT, "profiles", "r1", "r1r", "rocprof_summary.txt")]
src = srcs[0]
fetch, write = {}, {}
for line in [ln for f in srcs for ln in open(f)]:
    m = re.match(r"(.+?)\s+FETCH_SIZE ([\d.]+) MB over (\d+) disp \(x2 corrected ([\d.]+) MB\)", line)
    if m:
        fetch[m.group(1).strip()] = (float(m.group(4)) * 1e6, int(m.group(3)))
    m = re.match(r"(.+?)\s+WRITE_SIZE ([\d.]+) MB over (\d+) disp", line)
    if m:
        write[m.group(1).strip()] = (float(m.group(2)) * 1e6, int(m.group(3)))
sys.path.insert(0, ROOT)
from bench import csrc_sha16  # noqa: E402
sha = next((a[4:] for a in sys.argv[1:] if a.startswith("sha=")), None) or csrc_sha16()  # the build the counters were collected on
out = {"source": ", ".join(os.path.relpath(os.path.abspath(f), ROOT) for f in srcs), "csrc_sha16": sha, "unit": "bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) / dispatches",
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

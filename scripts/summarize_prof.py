#!/usr/bin/env python
"""Condense the rocprofv3 output of scripts/prof_round.sh into small text/JSON summaries:
per-kernel time (from *_kernel_stats.csv) and per-kernel mean counter values per dispatch
(from *_counter_collection.csv).  FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports
half the bytes of wide coalesced reads); both raw and corrected values are printed."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    return name[:90]


def main():
    out = sys.argv[1]
    res = {}
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        print("== kernel stats:", os.path.relpath(f, out))
        rows = list(csv.DictReader(open(f)))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        res["kernel_stats"] = []
        for r in rows[:40]:
            e = dict(name=short(r["Name"]), calls=int(r["Calls"]), total_ms=float(r["TotalDurationNs"]) / 1e6,
                     avg_us=float(r["AverageNs"]) / 1e3, pct=float(r["Percentage"]))
            res["kernel_stats"].append(e)
            print(f'{e["pct"]:6.2f}%  calls {e["calls"]:5d}  total {e["total_ms"]:9.3f} ms  avg {e["avg_us"]:9.2f} us  {e["name"]}')
        print(f"total kernel time {tot / 1e6:.3f} ms")
    for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            print("== counters:", os.path.relpath(f, out))
            acc = defaultdict(lambda: defaultdict(float))
            cnt = defaultdict(lambda: defaultdict(int))
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                c = r["Counter_Name"]
                acc[k][c] += float(r["Counter_Value"])
                cnt[k][c] += 1
            tab = {}
            for k in acc:
                tab[k] = {c: dict(sum=acc[k][c], dispatches=cnt[k][c]) for c in acc[k]}
            res[os.path.basename(d)] = tab
            order = sorted(acc, key=lambda k: -max(acc[k].values()))
            for k in order[:30]:
                parts = []
                for c in acc[k]:
                    v = acc[k][c]
                    if c in ("FETCH_SIZE", "WRITE_SIZE"):
                        mb = v * 1024 / 1e6  # counters are in KiB
                        extra = f" (x2 corrected {2 * mb:.1f} MB)" if c == "FETCH_SIZE" else ""
                        parts.append(f"{c} {mb:.1f} MB over {cnt[k][c]} disp{extra}")
                    else:
                        parts.append(f"{c} {v:.4g}")
                print(f"{k[:70]:70s} " + "; ".join(parts))
    json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Static check of the gfx950 ISA hipcc generates for every kernel of the library (no GPU needed).

Flags the two code-generation patterns that cost the most on this path:
  * DRAIN  - an innermost loop that waits with `s_waitcnt vmcnt(0)` and issues fewer than 3 vector-memory
             loads per such wait: every iteration exposes the full memory latency with (almost) nothing
             else in flight (streaming loops that batch >= 3 loads per drain are not reported);
  * SERIAL - global/buffer stores that directly follow an `s_waitcnt vmcnt(0)`: each store waits for the
             previous one to be acknowledged (a load, a spill reload or a divergent branch sits between them).
Usage:  python scripts/asm_lint.py [file.hip ...]      (default: every smaat_unet_amd/csrc/*.hip)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def compile_asm(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


def demangle(names):
    try:
        p = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "-n"], input="\n".join(names), capture_output=True, text=True)
        return dict(zip(names, p.stdout.split("\n")))
    except OSError:
        return {n: n for n in names}


def analyse(text):
    funcs, name = {}, None
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+|k_\w+):", line)
        if m:
            name = m.group(1)
            funcs[name] = []
        elif name is not None:
            funcs[name].append(line)
            if "s_endpgm" in line:
                name = None
    rows = []
    for fn, lines in funcs.items():
        # loops = [label line index, back-edge line index] pairs: a branch to an earlier label
        labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
        loops = []
        for i, l in enumerate(lines):
            m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        drains = 0
        for a, b in loops:
            if any(a < a2 and b2 < b for a2, b2 in loops):  # not innermost
                continue
            body = lines[a:b]
            nload = sum(1 for l in body if re.search(r"\b(global|buffer|flat)_load", l))
            ndrain = sum(1 for l in body if "s_waitcnt vmcnt(0)" in l)
            if nload and ndrain and nload < 3 * ndrain:
                drains += 1
        serial = 0
        nstore = 0
        for i, l in enumerate(lines):
            if re.search(r"\b(global|buffer)_store", l):
                nstore += 1
                if any("s_waitcnt vmcnt(0)" in x for x in lines[max(0, i - 4):i]):
                    serial += 1
        rows.append((fn, len(loops), drains, nstore, serial))
    return rows


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "smaat_unet_amd", "csrc", "*.hip")))
    for f in files:
        rows = analyse(compile_asm(f))
        names = demangle([r[0] for r in rows])
        print(f"== {os.path.relpath(f, ROOT)}")
        for fn, nloops, drains, nstore, serial in rows:
            flags = []
            if drains:
                flags.append(f"DRAIN x{drains}")
            if serial > 2:
                flags.append(f"SERIAL {serial}/{nstore} stores")
            if flags:
                print(f"  {names[fn][:110]:110s} loops={nloops:2d}  " + ", ".join(flags))


if __name__ == "__main__":
    main()

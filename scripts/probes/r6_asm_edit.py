#!/usr/bin/env python3
"""ISA edits for scripts/probes/r6_asm_variant.sh: argv = in.s out.s tag.  The tag names the edit:
  same            no change (control: the assemble-and-bundle pipeline reproduces the compiler's own object)
  vgpr136_<sub>   kernels whose mangled name contains <sub>: .amdhsa_next_free_vgpr / .amdhsa_accum_offset -> 136 (three waves
                  per SIMD, the allocation of the failing build, code unchanged)
  drain_<sub>     s_waitcnt vmcnt(0) after every inline-asm load of those kernels (nothing stays in flight)
  drainset_<sub>  s_waitcnt vmcnt(0) after every THIRD inline-asm load (= after a whole set: x row, edge, dz)
  nopmov_<sub>    s_nop 1 in front of every v_mov_b64 of those kernels (two wait states between a 32-bit VALU write of one half of
                  a register pair and the 64-bit move that reads the pair)
  nopsA_/nopsB_/nopsC_<sub>   s_nop 3 after every VALU instruction of ONE region of each pipeline slot (a slot = counted wait ..
                  s_barrier): A = the window update (wait .. first v_pk_fma_f32), B = depthwise + operand split + LDS writes
                  (.. last ds_write), C = advance + issue + the copies in front of the barrier.  Which region is timing-sensitive?
  splitmov_<sub>  every v_mov_b64 vD[a:a+1], vS[b:b+1] replaced by two v_mov_b32 (no 64-bit move left; same wait states as the
                  original sequence otherwise)
"""
import re
import sys

src, dst, tag = sys.argv[1:4]
kind, _, sub = tag.partition("_")
sub = {"scalar": "ILi2ELb1ELb0EffLb0", "packed": "ILi2ELb1ELb1EffLb0"}.get(sub.split("_")[-1], sub)
lines = open(src).read().split("\n")
out, cur, in_kernel_meta, nload = [], None, None, 0
for ln in lines:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        cur, nload = m.group(1), 0
    m = re.match(r"\s*\.amdhsa_kernel (\S+)", ln)
    if m:
        in_kernel_meta = m.group(1)
    if ".end_amdhsa_kernel" in ln:
        in_kernel_meta = None
    if kind == "vgpr136" and in_kernel_meta and sub in in_kernel_meta:
        ln = re.sub(r"(\.amdhsa_next_free_vgpr|\.amdhsa_accum_offset) \d+", r"\1 136", ln)
    mm = re.match(r"(\s*)v_mov_b64_e32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\]\s*$", ln.split(";")[0].rstrip())
    if mm and cur and sub in cur and kind == "nopmov":
        out.append("\ts_nop 1")
    if mm and cur and sub in cur and kind == "splitmov":
        d0, s0 = int(mm.group(2)), int(mm.group(4))
        # (order: a pair may overlap its source shifted by one register)
        if d0 == s0 + 1:
            out += [f"\tv_mov_b32_e32 v{d0 + 1}, v{s0 + 1}", f"\tv_mov_b32_e32 v{d0}, v{s0}"]
        else:
            out += [f"\tv_mov_b32_e32 v{d0}, v{s0}", f"\tv_mov_b32_e32 v{d0 + 1}, v{s0 + 1}"]
        continue
    out.append(ln)
    if kind in ("drain", "drainset") and cur and sub in cur and ln.strip().startswith(";;#ASMEND") and len(out) >= 2 and "global_load" in out[-2]:
        nload += 1
        if kind == "drain" or nload % 3 == 0:
            out.append("\ts_waitcnt vmcnt(0)")
QUART = None
if re.match(r"nopsA[1-4]$", kind):  # a quarter of region A's VALU instructions (second bisection step)
    QUART, kind = int(kind[-1]) - 1, "nopsA"
NOPV = None
m_ = re.match(r"nopv(\d+)$", kind)
if m_:  # as nopat, but counted from the FIRST VALU instruction of region A (nopat12..30 turned out to sit in the ~40 scalar
    NOPV, kind = int(m_.group(1)), "nopsA"  # instructions of item bookkeeping that precede the window update: uninformative)
NOPAT = None
m_ = re.match(r"nopat(\d+)$", kind)
if m_:  # 16 wait states in front of the k-th instruction of region A of every slot (third bisection step: the curing
    NOPAT, kind = int(m_.group(1)), "nopsA"  # positions are the interval between the two instructions of the hazard)
ONLY = None
if kind in ("nopsDPP", "nopsEXEC"):  # s_nop 3 before and after every DPP instruction / after every write of EXEC, region A only
    ONLY, kind = kind, "nopsA"
if kind in ("nopsA", "nopsB", "nopsC"):
    res, cur, region = [], None, None
    body = out
    # per kernel: slot boundaries from the listing itself
    i = 0
    while i < len(body):
        ln = body[i]
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur, region = m.group(1), None
        t = ln.split(";")[0].strip()
        if cur and sub in cur:
            if re.match(r"s_waitcnt vmcnt\([1-9]\d*\)", t):
                # look ahead to the slot's end and its landmarks
                j = i
                while j < len(body) and "s_barrier" not in body[j]:
                    j += 1
                idx_pk = next((k for k in range(i, j) if "v_pk_fma_f32" in body[k]), j)
                idx_ds = max([k for k in range(i, j) if re.match(r"\s*ds_write", body[k])] or [i])
                lo, hi = {"nopsA": (i, idx_pk), "nopsB": (idx_pk, idx_ds + 1), "nopsC": (idx_ds + 1, j)}[kind]
                valu = [k for k in range(lo, hi) if re.match(r"\s*v_", body[k])]
                if QUART is not None:
                    n4 = (len(valu) + 3) // 4
                    valu = valu[QUART * n4:(QUART + 1) * n4]
                valu = set(valu)
                if NOPV is not None:
                    real = [k for k in range(lo, hi) if re.match(r"\s*[vs]_", body[k])]
                    first_v = next((q for q, k in enumerate(real) if re.match(r"\s*v_", body[k])), 0)
                    for k in range(i, j):
                        if first_v + NOPV < len(real) and k == real[first_v + NOPV]:
                            res += ["\ts_nop 7", "\ts_nop 7"]
                        res.append(body[k])
                    i = j
                    continue
                if NOPAT is not None:
                    real = [k for k in range(lo, hi) if re.match(r"\s*[vs]_", body[k])]
                    for k in range(i, j):
                        if NOPAT < len(real) and k == real[NOPAT]:
                            res += ["\ts_nop 7", "\ts_nop 7"]
                        res.append(body[k])
                    i = j
                    continue
                for k in range(i, j):
                    t2 = body[k].split(";")[0]
                    if ONLY == "nopsDPP":
                        if lo <= k < hi and "_dpp" in t2:
                            res += ["\ts_nop 3", body[k], "\ts_nop 3"]
                        else:
                            res.append(body[k])
                        continue
                    if ONLY == "nopsEXEC":
                        res.append(body[k])
                        if lo <= k < hi and re.match(r"\s*s_\w+\s+(exec|s\[\d+:\d+\], vcc)", t2) and ("exec" in t2):
                            res.append("\ts_nop 3")
                        continue
                    res.append(body[k])
                    if k in valu:
                        res.append("\ts_nop 3")
                i = j
                continue
        res.append(ln)
        i += 1
    out = res
open(dst, "w").write("\n".join(out))

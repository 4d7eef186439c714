#!/usr/bin/env python
"""CFG-aware hazard lint of the gfx950 ISA hipcc generates for the kernels that issue their global loads from INLINE ASM
(no GPU needed).  hipcc does not know that an `asm volatile("global_load_dwordx4 %0, ..." : "=v"(r))` is asynchronous: to the
compiler the destination holds its value as soon as the statement has executed, and nothing stops it from copying, spilling
or re-using that register before the hand-counted `s_waitcnt vmcnt(N)` that really completes the load.  Whether it does is
a matter of register allocation -- it did, in round 5, for one instantiation of k_dsconv_wgrad_split (profiles/r6/).

The lint walks the control-flow graph of every kernel (labels, s_branch / s_cbranch_*, back-edges included) to a fixpoint
and tracks, per vector register, how many vector-memory operations have been issued since the load that targets it (the
vmcnt queue position; loads and stores return in order on gfx9).  `s_waitcnt vmcnt(N)` completes every entry at position
>= N.  At a join the state of a register is the LEAST complete of its predecessors'.  Reported:

  INFLIGHT  an instruction reads or writes a VGPR/AGPR that is the destination of a load not yet covered by a wait
            (a v_mov copying a prefetched register, a VALU op on it, a second load re-targeting it, a spill of it);
  SGPRHAZ   a VALU instruction writes an SGPR (v_readfirstlane / v_readlane / v_cmp) and a vector-memory instruction uses
            that SGPR as its scalar base fewer than 5 wait states later -- a gfx9 hazard LLVM's hazard recogniser resolves for
            its own VMEM instructions and does NOT look for inside inline asm;
  RETARGET  informational: a load whose destination is still the destination of an earlier load in flight (harmless: loads
            return in order, the later data lands last);
  UNDERWAIT informational: a counted wait whose N exceeds the number of loads in the queue on some path (harmless).

Usage:  python scripts/isa_hazards.py [file.hip | file.s ...]      (default: the row-walking / split GEMM sources)
        python scripts/isa_hazards.py --kernel <substring> file.s   restrict the report
        python scripts/isa_hazards.py --steady ...                  assume no slot of a software pipeline is ever skipped: the
                                                                    arms of `if (more) { wait; commit; issue }` that bypass the
                                                                    asm loads are removed from the graph (for kernels whose tail
                                                                    is conditional; see conditional_issue_edges)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ASM_LOAD_SOURCES = ("dswgrad.hip", "dsrows.hip", "splitmma.hip", "dsconv_split.hip", "dsbwd.hip", "pwgemm.hip", "dwrows.hip", "uprows.hip", "bf16gemm.hip")
VMCNT_MAX = 63
SGPR_WAIT_STATES = 5  # VALU writes SGPR -> VMEM reads that SGPR (gfx90a / gfx940 family hazard table)

_REG = re.compile(r"\b([vas])(\d+)\b|\b([vas])\[(\d+):(\d+)\]")


def compile_asm(src, extra=()):
    if src.endswith(".s"):
        return open(src).read()
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", out, *extra]
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


def split_functions(text):
    """{mangled name: [instruction / label lines]} for every kernel of an assembly listing."""
    funcs, name = {}, None
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+|k_\w+):", line)
        if m:
            name = m.group(1)
            funcs[name] = []
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            name = None
            continue
        funcs[name].append(line)
    return funcs


class Ins:
    __slots__ = ("idx", "op", "text", "vregs", "sregs", "dst_v", "dst_s", "asm")

    def __init__(self, idx, text, in_asm):
        self.idx, self.text, self.asm = idx, text, in_asm
        body = text.split(";")[0].strip()
        parts = body.split(None, 1)
        self.op = parts[0]
        ops = [o.strip() for o in (parts[1].split(",") if len(parts) > 1 else [])]
        self.vregs, self.sregs = set(), set()
        per_op = []
        for o in ops:
            regs = set()
            for m in _REG.finditer(o):
                if m.group(1):
                    regs.add((m.group(1), int(m.group(2))))
                else:
                    regs.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
            if re.search(r"\bvcc\b", o):
                regs.update({("s", 106), ("s", 107)})
            per_op.append(regs)
            self.vregs.update(r for r in regs if r[0] in "va")
            self.sregs.update(r for r in regs if r[0] == "s")
        first = per_op[0] if per_op else set()
        self.dst_v = {r for r in first if r[0] in "va"}
        self.dst_s = {r for r in first if r[0] == "s"}
        if re.match(r"v_pk_(fma|mul|add)_f32", self.op):
            # packed f32 sources are register PAIRS of which op_sel (low result lane) / op_sel_hi (high lane) pick one register
            # each; hipcc pairs a value with an unrelated neighbour register and selects the value twice ("op_sel_hi:[1,0,1]"
            # = source 1 is a scalar broadcast).  A register that is named but selected by neither lane is not read.
            nsrc = len(per_op) - 1
            sel = [[0] * nsrc, [1] * nsrc]
            for k, key in enumerate(("op_sel", "op_sel_hi")):
                m = re.search(key + r":\[([01,]+)\]", body)
                if m:
                    vals = [int(v) for v in m.group(1).split(",")]
                    sel[k][:len(vals)] = vals
            self.vregs = set(self.dst_v)
            for i in range(nsrc):
                o = ops[1 + i].split()[0] if ops[1 + i] else ""
                m = re.match(r"([va])\[(\d+):(\d+)\]", o)
                if m:
                    lo = int(m.group(2))
                    self.vregs.update({(m.group(1), lo + sel[0][i]), (m.group(1), lo + sel[1][i])})
                else:
                    self.vregs.update(r for r in per_op[1 + i] if r[0] in "va")

    @property
    def is_vmem(self):
        return bool(re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", self.op)) or self.op.startswith("tbuffer_")

    @property
    def is_vmem_load_to_vgpr(self):
        if not re.match(r"(global|buffer|flat|scratch)_load", self.op) and not self.op.startswith("tbuffer_load"):
            return False
        return "_lds_" not in self.op and not re.search(r"\blds\b", self.text.split(";")[0])

    @property
    def vmcnt(self):
        if self.op != "s_waitcnt":
            return None
        m = re.search(r"vmcnt\((\d+)\)", self.text)
        if m:
            return int(m.group(1))
        m = re.match(r"\s*s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s*$", self.text.split(";")[0])
        if m:  # raw immediate: vmcnt = bits [3:0] | bits [15:14] << 4
            v = int(m.group(1), 0)
            return (v & 15) | ((v >> 14) & 3) << 4
        return None

    @property
    def wait_states(self):
        if self.op == "s_nop":
            m = re.search(r"s_nop\s+(\d+)", self.text)
            return 1 + (int(m.group(1)) if m else 0)
        return 1

    @property
    def valu_writes_sgpr(self):
        if not self.op.startswith("v_"):
            return set()
        if self.op.startswith(("v_readfirstlane", "v_readlane")):
            return set(self.dst_s)
        if self.op.startswith(("v_cmp", "v_cmpx")) or re.match(r"v_(add|sub|subrev)c?_co_", self.op) or self.op.startswith("v_div_scale") or self.op.startswith("v_mad_u64") or self.op.startswith("v_mad_i64"):
            out = set(self.dst_s)
            if self.op.endswith("_e32") or (not self.dst_s and self.op.startswith("v_cmp")):
                out |= {("s", 106), ("s", 107)}
            return out
        return set()


def parse(lines):
    """-> (instructions, blocks, succ): blocks = list of (first, last+1) instruction index ranges."""
    ins, labels, in_asm = [], {}, False
    for ln in lines:
        s = ln.strip()
        if not s:
            continue
        if s.startswith((";;#ASMSTART", ";APP")):
            in_asm = True
            continue
        if s.startswith((";;#ASMEND", ";NO_APP")):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        ins.append(Ins(len(ins), s, in_asm))
    starts = {0} | set(labels.values())
    for i, x in enumerate(ins):
        if x.op.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc", "s_swappc")) and i + 1 < len(ins):
            starts.add(i + 1)
    starts = sorted(s for s in starts if s < len(ins))
    blocks = [(s, e) for s, e in zip(starts, starts[1:] + [len(ins)])]
    bidx = {s: k for k, (s, _) in enumerate(blocks)}
    succ = []
    for k, (s, e) in enumerate(blocks):
        last = ins[e - 1]
        out = []
        m = re.search(r"(\.LBB\d+_\d+)", last.text)
        if last.op == "s_branch":
            if m and m.group(1) in labels and labels[m.group(1)] in bidx:
                out.append(bidx[labels[m.group(1)]])
        elif last.op.startswith("s_cbranch"):
            if m and m.group(1) in labels and labels[m.group(1)] in bidx:
                out.append(bidx[labels[m.group(1)]])
            if k + 1 < len(blocks):
                out.append(k + 1)
        elif last.op in ("s_endpgm",) or last.op.startswith(("s_setpc", "s_swappc")):
            pass
        elif k + 1 < len(blocks):
            out.append(k + 1)
        succ.append(out)
    return ins, blocks, succ


def _join(states):
    out = {}
    for st in states:
        for r, p in st.items():
            out[r] = min(out.get(r, 1 << 30), p)
    return out


def _src_vregs(x):
    """vector registers an instruction READS (all operands but the first; the first too when it is not a pure destination)"""
    body = x.text.split(";")[0].strip().split(None, 1)
    ops = body[1].split(",") if len(body) > 1 else []
    out = set()
    for o in ops[1:]:
        for m in _REG.finditer(o):
            if m.group(1):
                if m.group(1) in "va":
                    out.add((m.group(1), int(m.group(2))))
            elif m.group(3) in "va":
                out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def _transfer(ins, s, e, vm_in, sg_in, report):
    """vm: {vector register: vmcnt queue position}; sg: {sgpr: wait states since a VALU wrote it}"""
    vm, sg = dict(vm_in), dict(sg_in)
    for x in ins[s:e]:
        n = x.vmcnt
        if n is not None:
            if report is not None and vm and n > 0 and max(vm.values()) < n - 1 and x.asm:
                report.append(("UNDERWAIT", x.idx, x.text, f"deepest entry at position {max(vm.values())}"))
            vm = {r: p for r, p in vm.items() if p < n}
        touched = x.vregs & vm.keys()
        if touched and report is not None:
            # a LOAD whose destination is still the target of an earlier load is harmless: loads return in order, the later
            # data lands last and nobody reads the register in between (hipcc does this when it reuses a register it believes
            # free, and on the infeasible then-AND-else paths of a two-armed issue)
            retarget = x.is_vmem_load_to_vgpr and touched <= x.dst_v and not (touched & _src_vregs(x))
            regs = ",".join(f"{t}{i}" for t, i in sorted(touched))
            report.append(("RETARGET" if retarget else "INFLIGHT", x.idx, x.text,
                           f"{regs} {'re-targeted by a load' if retarget else 'touched'} with its load at queue position {min(vm[r] for r in touched)}"))
        if x.is_vmem:
            used = {r for r in x.sregs if r in sg}
            if used and report is not None:
                regs = ",".join(f"s{i}" for _, i in sorted(used))
                report.append(("SGPRHAZ", x.idx, x.text, f"{regs} written by a VALU instruction {min(sg[r] for r in used)} wait state(s) earlier (needs {SGPR_WAIT_STATES})"))
            vm = {r: min(p + 1, VMCNT_MAX + 1) for r, p in vm.items()}
            if x.is_vmem_load_to_vgpr:
                for r in x.dst_v:
                    vm[r] = 0
        # (a non-load instruction writing a pending register leaves it pending: the load will still overwrite it)
        ws = x.wait_states
        sg = {r: a + ws for r, a in sg.items() if a + ws < SGPR_WAIT_STATES and r not in x.dst_s}
        for r in x.valu_writes_sgpr:
            sg[r] = 0
    return vm, sg


def _ipdom(nblocks, succ):
    """immediate post-dominator of every block (virtual exit = nblocks); iterative set algorithm on the reverse graph"""
    exit_ = nblocks
    full = set(range(nblocks + 1))
    pdom = [set(full) for _ in range(nblocks)] + [{exit_}]
    sx = [list(o) if o else [exit_] for o in succ]
    changed = True
    while changed:
        changed = False
        for k in range(nblocks - 1, -1, -1):
            new = set(full)
            for t in sx[k]:
                new &= pdom[t]
            new = new | {k}
            if new != pdom[k]:
                pdom[k], changed = new, True
    out = []
    for k in range(nblocks):
        cand = pdom[k] - {k}
        # the immediate one is post-dominated by every other candidate
        imm = next((c for c in cand if all(o in pdom[c] for o in cand)), exit_)
        out.append(imm)
    return out


def conditional_issue_edges(ins, blocks, succ):
    """Edges (b, arm) of two-armed branches where only the OTHER arm (up to the branches' merge point) contains inline-asm
    loads: `if (more work) { wait; commit; issue }` written in the source.  Taking such an edge means a slot of the software
    pipeline is skipped; the kernels only ever skip at the very end of a walk (and then never commit again), which a
    path-insensitive analysis cannot know -- `steady` mode removes these edges and so proves the steady state."""
    ipd = _ipdom(len(blocks), succ)
    has_asm = [any(x.asm and x.is_vmem_load_to_vgpr for x in ins[s:e]) for s, e in blocks]

    def region(start, stop, avoid):
        seen, work = set(), [start]
        while work:
            k = work.pop()
            if k in seen or k == stop or k == avoid or k >= len(blocks):
                continue
            seen.add(k)
            work.extend(succ[k])
        return seen

    out = set()
    for b, o in enumerate(succ):
        if len(o) != 2 or o[0] == o[1]:
            continue
        m = ipd[b]
        arms = [region(x, m, b) for x in o]
        loads = [any(has_asm[k] for k in arm) for arm in arms]
        if loads[0] != loads[1]:
            out.add((b, o[1] if loads[0] else o[0]))
    return out


def analyse_function(lines, steady=False):
    ins, blocks, succ = parse(lines)
    if not ins:
        return [], 0
    if steady:
        drop = conditional_issue_edges(ins, blocks, succ)
        succ = [[t for t in o if (k, t) not in drop] for k, o in enumerate(succ)]
    pred = [[] for _ in blocks]
    for k, out in enumerate(succ):
        for t in out:
            pred[t].append(k)
    vm_out, sg_out = [None] * len(blocks), [None] * len(blocks)
    work = [0]
    while work:
        k = work.pop()
        ps = [p for p in pred[k] if vm_out[p] is not None]
        vm_in = _join([vm_out[p] for p in ps]) if ps else {}
        sg_in = _join([sg_out[p] for p in ps]) if ps else {}
        vm, sg = _transfer(ins, *blocks[k], vm_in, sg_in, None)
        if vm != vm_out[k] or sg != sg_out[k]:
            vm_out[k], sg_out[k] = vm, sg
            work.extend(succ[k])
    report = []
    for k, (s, e) in enumerate(blocks):
        if vm_out[k] is None:
            continue  # unreachable
        ps = [p for p in pred[k] if vm_out[p] is not None]
        _transfer(ins, s, e, _join([vm_out[p] for p in ps]) if ps else {}, _join([sg_out[p] for p in ps]) if ps else {}, report)
    nasm = sum(1 for x in ins if x.asm and x.is_vmem_load_to_vgpr)
    return report, nasm


def analyse_text(text, kernel=None, steady=False):
    """-> {mangled kernel name: (findings, number of inline-asm loads)} for the kernels that have inline-asm loads"""
    out = {}
    for fn, lines in split_functions(text).items():
        if kernel and kernel not in fn:
            continue
        rep, nasm = analyse_function(lines, steady)
        if nasm or kernel:
            out[fn] = (rep, nasm)
    return out


def main():
    args = sys.argv[1:]
    kernel = None
    steady = "--steady" in args
    if steady:
        args.remove("--steady")
    if "--kernel" in args:
        i = args.index("--kernel")
        kernel = args[i + 1]
        del args[i:i + 2]
    files = args or [os.path.join(ROOT, "smaat_unet_amd", "csrc", f) for f in ASM_LOAD_SOURCES]
    bad = 0
    for f in files:
        res = analyse_text(compile_asm(f), kernel, steady)
        names = demangle(list(res))
        print(f"== {os.path.relpath(f, ROOT) if f.startswith(ROOT) else f}: {len(res)} kernels with inline-asm loads")
        for fn, (rep, nasm) in res.items():
            hard = [r for r in rep if r[0] in ("INFLIGHT", "SGPRHAZ")]
            bad += len(hard)
            print(f"  {names[fn][:120]:120s} asm loads={nasm:3d}  {'OK' if not hard else 'HAZARDS: %d' % len(hard)}")
            for kind, idx, text, why in rep:
                if kind not in ("UNDERWAIT", "RETARGET") or "-v" in sys.argv:
                    print(f"      {kind:9s} #{idx:5d}  {text.split(';')[0].strip():70s} {why}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

"""CPU (hipcc cross-compiles gfx950 without a GPU): the generated ISA of the bf16-split GEMM family must keep
the two properties that were worth 30-40 % of those kernels:
  * no loop of the producer waves drains the vector-memory counter (s_waitcnt vmcnt(0)) while loads are in
    flight -- the prefetch depth is what the explicit counted waits say, not one chunk;
  * the store tail of a tile is not serialised (no s_waitcnt vmcnt(0) between consecutive stores).
Uses scripts/asm_lint.py, the same check that produced profiles/r1/r1s/asm_lint.txt."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="hipcc not available")
def test_split_gemm_isa_has_no_drained_loops_or_serialised_stores():
    spec = importlib.util.spec_from_file_location("asm_lint", os.path.join(ROOT, "scripts", "asm_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    rows = lint.analyse(lint.compile_asm(os.path.join(ROOT, "smaat_unet_amd", "csrc", "splitmma.hip")))
    seen = 0
    for fn, nloops, drains, nstore, serial in rows:
        if "k_pw_split_p" in fn or "k_wgrad_split" in fn:
            seen += 1
            assert drains == 0, (fn, "a producer/consumer loop waits with vmcnt(0) with < 3 loads in flight")
            assert serial <= 2, (fn, f"{serial}/{nstore} stores follow an s_waitcnt vmcnt(0)")
    assert seen >= 8  # every instantiation of the two kernels was inspected


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="hipcc not available")
def test_scalar_base_loads_really_have_a_scalar_base():
    """Round 4: inline-asm loads of the form `global_load_* vdst, voffset, sbase` take their base through an "s" constraint,
    which does NOT make a value uniform -- when hipcc keeps the pointer in VGPRs (it did for a base formed with a select
    on a bool) it prints a VGPR pair into the scalar slot, the assembler accepts it, and the kernel faults on the GPU.
    Every such load / store in the recompute weight gradient and the split GEMMs must name an SGPR pair."""
    import re
    import subprocess
    import tempfile
    for src in ("dswgrad.hip", "splitmma.hip", "dsconv_split.hip"):
        with tempfile.NamedTemporaryFile(suffix=".s") as f:
            subprocess.run([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S",
                            "--cuda-device-only", os.path.join(ROOT, "smaat_unet_amd", "csrc", src), "-o", f.name],
                           check=True, capture_output=True)
            asm = open(f.name).read()
        bad = [ln.strip() for ln in asm.splitlines()
               if re.match(r"\s*(global|buffer)_(load|store)", ln) and re.search(r",\s*v\d+,\s*v\[\d+:\d+\]", ln)]
        assert not bad, (src, bad[:3])
        if src == "dswgrad.hip":
            assert len(re.findall(r"global_load_dwordx?4? v\[?\d+[:\d\]]*, v\d+, s\[\d+:\d+\]", asm)) >= 24


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="hipcc not available")
def test_row_walking_kernels_do_not_drain_their_prefetch_queue():
    """Round 4: the row-walking kernels (dswgrad.hip, dsrows.hip) keep PD rows of inline-asm loads in flight behind counted
    `s_waitcnt vmcnt(N)`.  hipcc does not see those loads -- but a compiler-visible load (the depthwise weights, the BatchNorm
    coefficients) that was still pending at the loop entry made it emit `s_waitcnt vmcnt(0)` at the value's first use INSIDE
    the loop, once per iteration, in all but two instantiations: the queue was emptied every chunk.  The kernels now use those
    values once before the first asm load; no loop of theirs may contain a full drain -- which also means no instantiation may
    spill (scratch reloads wait with vmcnt(0)): the f32 two-channel build of the fused forward did until its third weight
    plane moved to LDS and its store addresses to scalar row pointers.
    (Round 6: loops laid out BEFORE the function's first inline-asm load are not counted -- the two-term fused forward bounds
    |y| in a prologue whose loops read the depthwise weights with ordinary loads; nothing is in flight there to be drained.)"""
    import re
    import subprocess
    import tempfile
    for src in ("dswgrad.hip", "dsrows.hip"):
        with tempfile.NamedTemporaryFile(suffix=".s") as f:
            subprocess.run([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S",
                            "--cuda-device-only", os.path.join(ROOT, "smaat_unet_amd", "csrc", src), "-o", f.name],
                           check=True, capture_output=True)
            asm = open(f.name).read()
        fn, inloop, drains, seen, in_asm, asm_load = None, False, {}, 0, False, False
        for ln in asm.splitlines():
            m = re.match(r"^(_Z\w+):", ln)
            if m:
                fn, inloop, in_asm, asm_load = m.group(1), False, False, False
                drains[fn] = 0
                continue
            if "#ASMSTART" in ln:
                in_asm = True
            elif "#ASMEND" in ln:
                in_asm = False
            elif in_asm and re.search(r"\b(global|buffer)_load_", ln):
                asm_load = True
            if ln.startswith(".Lfunc_end"):
                fn = None
            if fn is None:
                continue
            if re.match(r"^\.LBB\d+_\d+:", ln) or re.match(r"^; %bb\.\d+:", ln):
                inloop = "in Loop" in ln or "Loop Header" in ln
            if inloop and asm_load and "s_waitcnt vmcnt(0)" in ln:
                drains[fn] += 1
        for fn, n in drains.items():
            if "k_dsconv_wgrad_split" in fn or "k_dsconv_rows_fwd" in fn:
                seen += 1
                assert n == 0, (fn, f"{n} full drains of the vector-memory counter inside a loop")
        assert seen >= 12, (src, seen)


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="hipcc not available")
def test_attention_backward_loops_keep_their_prefetch():
    """Round 5: the channel loops of the three-pass attention backward (cbam.hip) are hand-pipelined -- the loads of the next two
    channels are issued before the current two are processed.  The first form of k_cbam_bwd_apply_v4 issued a load BETWEEN the
    stores of a trip, hipcc answered with `s_waitcnt vmcnt(0)` in the middle of the loop and the kernel ran at 3.4 TB/s instead of
    5.  In the f32 instantiations that carry the step: no drained loop, and no vector load between the first and
    the last store of the apply kernel's loop body."""
    import re
    spec = importlib.util.spec_from_file_location("asm_lint", os.path.join(ROOT, "scripts", "asm_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    asm = lint.compile_asm(os.path.join(ROOT, "smaat_unet_amd", "csrc", "cbam.hip"))
    seen = 0
    for fn, nloops, drains, nstore, serial in lint.analyse(asm):
        if ("k_cbam_bwd_gate_ds_v4If" in fn or "k_cbam_bwd_ds2_v4If" in fn or "k_cbam_bwd_apply_v4IfLb1" in fn):
            seen += 1
            assert drains == 0, fn
    assert seen == 3
    name = [m for m in re.findall(r"^(_Z\w*k_cbam_bwd_apply_v4IfLb1\w*):", asm, re.M)][0]
    body = asm[asm.index("\n" + name + ":"):]
    body = body[:body.index("s_endpgm")]
    loop = body[body.rindex(".LBB", 0, body.rindex("s_cbranch_scc")):body.rindex("s_cbranch_scc")]  # the last (main) loop
    ops = [ln.split()[0] for ln in loop.splitlines() if re.match(r"\s*(global_load|buffer_load|buffer_store|global_store)", ln)]
    stores = [i for i, o in enumerate(ops) if "store" in o]
    assert len(stores) == 4, ops  # two channels x two rows
    assert not any("load" in o for o in ops[stores[0]:stores[-1]]), ops


# ------------------------------------------------------------------------------------------------ round 6: CFG-aware hazard lint
def _hazards():
    spec = importlib.util.spec_from_file_location("isa_hazards", os.path.join(ROOT, "scripts", "isa_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="hipcc not available")
@pytest.mark.parametrize("src,min_kernels", [("dswgrad.hip", 14), ("dsrows.hip", 16), ("splitmma.hip", 13), ("dsconv_split.hip", 24), ("dsbwd.hip", 2)])
def test_no_instruction_touches_a_prefetched_register_before_its_wait(src, min_kernels):
    """Round 6: hipcc believes the destination of an inline-asm `global_load` is valid as soon as the
    statement has executed -- under register pressure it spills it (a scratch store of a register whose load has not landed,
    reloaded later as if it held the data), copies it or lends it to a temporary.  The round-6 form of the row-walking and split
    GEMM kernels (walks padded to the unroll depth, every slot unconditional, one load form per prefetch) makes the proof
    path-insensitive: on EVERY path of the control-flow graph, back-edges included, no instruction may read or write (other
    than re-load) a vector register between the inline-asm load that targets it and the counted s_waitcnt that covers it, and
    no vector-memory instruction may take a scalar base a VALU instruction wrote fewer than 5 wait states earlier.
    EVERY instantiation in the library is checked: a hipcc upgrade (or an edit) that moves one into the state of round 5's
    k_dsconv_wgrad_split<2, AFF, scalar> fails here, on the CPU, instead of silently on the GPU."""
    H = _hazards()
    res = H.analyse_text(H.compile_asm(os.path.join(ROOT, "smaat_unet_amd", "csrc", src)))
    assert len(res) >= min_kernels, (src, len(res))
    bad = {fn: [r for r in rep if r[0] in ("INFLIGHT", "SGPRHAZ")] for fn, (rep, _) in res.items()}
    bad = {fn: r for fn, r in bad.items() if r}
    assert not bad, {fn: [(k, t.split(";")[0].strip(), why) for k, _, t, why in r[:3]] for fn, r in bad.items()}


def test_hazard_lint_finds_the_planted_hazards():
    """The lint itself, on hand-written listings: a copy of a prefetched register over a loop back-edge, a spill right after the
    issue, a counted wait one load short on ONE of two paths, a VALU-written scalar base -- and the clean forms of each."""
    H = _hazards()

    def run(body):
        rep, _ = H.analyse_function(body.strip().split("\n"))
        return [r[0] for r in rep if r[0] in ("INFLIGHT", "SGPRHAZ")]

    load = ";;#ASMSTART\nglobal_load_dwordx4 v[4:7], v1, s[2:3]\n;;#ASMEND"
    wait0 = ";;#ASMSTART\ns_waitcnt vmcnt(0)\n;;#ASMEND"
    assert run(f"{load}\n{wait0}\nv_add_f32_e32 v8, v4, v5\ns_endpgm") == []
    assert run(f"{load}\nv_mov_b32_e32 v9, v4\n{wait0}\ns_endpgm") == ["INFLIGHT"]                     # copy before the wait
    assert run(f"{load}\nscratch_store_dwordx4 off, v[4:7], off\n{wait0}\ns_endpgm") == ["INFLIGHT"]    # spill of a prefetched set
    assert run(f"{load}\nglobal_load_dwordx4 v[4:7], v1, s[2:3]\n{wait0}\nv_mov_b32_e32 v9, v4\ns_endpgm") == []  # re-load: harmless
    # software pipeline, two sets, counted wait vmcnt(1): set A is complete when one younger load is outstanding
    loop_ok = (".LBB0_1:\n;;#ASMSTART\ns_waitcnt vmcnt(1)\n;;#ASMEND\nv_mov_b32_e32 v20, v4\n;;#ASMSTART\nglobal_load_dword v4, v1, s[2:3]\n;;#ASMEND\n"
               ";;#ASMSTART\ns_waitcnt vmcnt(1)\n;;#ASMEND\nv_mov_b32_e32 v21, v5\n;;#ASMSTART\nglobal_load_dword v5, v1, s[2:3]\n;;#ASMEND\n"
               "s_cbranch_scc1 .LBB0_1\ns_endpgm")
    pro = ";;#ASMSTART\nglobal_load_dword v4, v1, s[2:3]\n;;#ASMEND\n;;#ASMSTART\nglobal_load_dword v5, v1, s[2:3]\n;;#ASMEND\n"
    assert run(pro + loop_ok) == []
    # the same loop with a phi copy of the set issued LAST placed on the back-edge: only the graph walk sees it
    assert "INFLIGHT" in run(pro + loop_ok.replace("s_cbranch_scc1 .LBB0_1", "v_mov_b32_e32 v30, v5\ns_cbranch_scc1 .LBB0_1"))
    # a slot whose issue can be skipped: on the skipping path the next wait is one load short
    skip = pro + (".LBB0_1:\n;;#ASMSTART\ns_waitcnt vmcnt(1)\n;;#ASMEND\nv_mov_b32_e32 v20, v4\ns_cbranch_scc0 .LBB0_2\n"
                  ";;#ASMSTART\nglobal_load_dword v4, v1, s[2:3]\n;;#ASMEND\n.LBB0_2:\n;;#ASMSTART\ns_waitcnt vmcnt(1)\n;;#ASMEND\nv_mov_b32_e32 v21, v5\n"
                  ";;#ASMSTART\nglobal_load_dword v5, v1, s[2:3]\n;;#ASMEND\ns_cbranch_scc1 .LBB0_1\ns_endpgm")
    assert "INFLIGHT" in run(skip)
    # VALU writes the scalar base of a vector-memory instruction: 5 wait states
    assert run("v_readfirstlane_b32 s2, v1\nv_readfirstlane_b32 s3, v2\nglobal_load_dword v4, v1, s[2:3]\ns_waitcnt vmcnt(0)\ns_endpgm") == ["SGPRHAZ"]
    assert run("v_readfirstlane_b32 s2, v1\nv_readfirstlane_b32 s3, v2\ns_nop 4\nglobal_load_dword v4, v1, s[2:3]\ns_waitcnt vmcnt(0)\ns_endpgm") == []
    # packed f32 operands: a register that is named but selected by neither lane is not read
    assert run(f"{load}\nv_pk_fma_f32 v[10:11], v[12:13], v[3:4], v[14:15] op_sel_hi:[1,0,1]\n{wait0}\ns_endpgm") == []
    assert run(f"{load}\nv_pk_fma_f32 v[10:11], v[12:13], v[3:4], v[14:15]\n{wait0}\ns_endpgm") == ["INFLIGHT"]

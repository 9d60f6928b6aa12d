"""scripts/isa_lint.py on the CPU: the hand-scheduled gfx950 kernels of the built library satisfy their ISA-level rules, and the rules BITE
(both round-4 miscompiles of k_wconv4 were found by wrong results on metered GPU time; a compiler bump can break any of
these silently).  Doctored listings: a deleted wait-state fence, a uniform branch around epilogue stores and a spill inside the K loop
are each reported."""
import importlib.util
import os
import re
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lint():
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "scripts", "isa_lint.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    if not os.path.exists(os.path.join(m.CSRC, "conv_wino4_f16x3.o")):
        import sys
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return m


@pytest.fixture(scope="module")
def wconv4_listing(lint):
    with tempfile.TemporaryDirectory() as tmp:
        dis, notes = lint.disassemble(os.path.join(lint.CSRC, "conv_wino4_f16x3.o"), tmp)
    # one kernel only (the no-residual instantiation), as text lines
    lines, keep = [], False
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
        if m:
            keep = m.group(1) == "_Z8k_wconv4ILi0EEv6WConvP"
        if keep:
            lines.append(ln)
    return lines, notes


def spec_of(lint, name):
    return [s for s in lint.SPECS if s["name"] == name][0]


def test_built_library_passes(lint):
    assert lint.run() == []


def test_clean_listing_passes(lint, wconv4_listing):
    lines, notes = wconv4_listing
    res = lint.lint_listing(spec_of(lint, "k_wconv4"), "\n".join(lines), notes)
    assert res == {"_Z8k_wconv4ILi0EEv6WConvP": []}


def test_deleted_fence_is_caught(lint, wconv4_listing):
    """w4_fence (s_nop 15 / s_nop 3 tied to the accumulators) deleted once: the fence count is off (and, where a compiler copy follows the
    K loop's last MFMAs, the hazard rule fires too)."""
    lines, notes = wconv4_listing
    idx = [k for k in range(len(lines) - 1) if re.match(r"^\s+s_nop 15\b", lines[k]) and re.match(r"^\s+s_nop 3\b", lines[k + 1])]
    assert len(idx) == spec_of(lint, "k_wconv4")["fences"]
    doctored = lines[:idx[-1]] + lines[idx[-1] + 2:]
    errs = lint.lint_listing(spec_of(lint, "k_wconv4"), "\n".join(doctored), notes)["_Z8k_wconv4ILi0EEv6WConvP"]
    assert any("fence" in e for e in errs), errs


def test_branch_around_epilogue_stores_is_caught(lint, wconv4_listing):
    """A uniform branch over four of the always-issued epilogue stores (the shortest DMA-to-wait path of the clean kernel holds 66: the 64
    output stores plus the column sums): a path from the next tile's prefetch DMA to the tile-start wait then issues 62 stores, one fewer
    than s_waitcnt vmcnt(63) leaves in flight."""
    lines, notes = wconv4_listing
    ins = lint.parse_disassembly("\n".join(lines))["_Z8k_wconv4ILi0EEv6WConvP"]
    stores = [k for k, i in enumerate(ins) if i.op.startswith("buffer_store")]
    # the kernel holds several epilogue variants (whole tiles / masked tiles): in EVERY scalar-branch-free run of at least eight stores,
    # branch from the instruction in front of its third store to behind its sixth
    scalar_br = [k for k, i in enumerate(ins) if i.target is not None and not i.op.startswith("s_cbranch_exec")]
    groups = [[stores[0]]]
    for k in stores[1:]:
        if any(groups[-1][-1] < q < k for q in scalar_br):
            groups.append([k])
        else:
            groups[-1].append(k)
    doctored = list(lines)
    n_doctored = 0
    for g in groups:
        if len(g) < 8:
            continue
        src, dst = ins[g[2] - 1], ins[g[5] + 1]
        imm = (dst.addr - src.addr - 4) // 4
        assert 0 < imm < 0x7fff and src.target is None
        pat = re.compile(r"//\s*0*%X:" % src.addr, re.I)
        doctored = [re.sub(r"^\s+\S.*?//", f"\ts_cbranch_scc1 {imm} //", ln) if pat.search(ln) else ln for ln in doctored]
        n_doctored += 1
    assert n_doctored >= 2 and doctored != lines
    errs = lint.lint_listing(spec_of(lint, "k_wconv4"), "\n".join(doctored), notes)["_Z8k_wconv4ILi0EEv6WConvP"]
    assert any("tile-start wait" in e for e in errs), errs


def test_spill_inside_k_loop_is_caught(lint, wconv4_listing):
    lines, notes = wconv4_listing
    ins = lint.parse_disassembly("\n".join(lines))["_Z8k_wconv4ILi0EEv6WConvP"]
    lo, hi, n = lint.innermost_mfma_loop(ins, "v_mfma_f32_16x16x32_f16")
    assert n == 864
    victim = [i for i in ins[lo:hi] if i.op == "s_nop" or i.op.startswith("s_mov") or i.op.startswith("v_add")][0]
    pat = re.compile(r"//\s*0*%X:" % victim.addr, re.I)
    doctored = [re.sub(r"^\s+\S.*?//", "\tscratch_load_dword v1, off, off //", ln) if pat.search(ln) else ln for ln in lines]
    errs = lint.lint_listing(spec_of(lint, "k_wconv4"), "\n".join(doctored), notes)["_Z8k_wconv4ILi0EEv6WConvP"]
    assert any("inside the K loop" in e for e in errs), errs
    # ... and an accumulator move there likewise
    doctored = [re.sub(r"^\s+\S.*?//", "\tv_accvgpr_read_b32 v1, a0 //", ln) if pat.search(ln) else ln for ln in lines]
    errs = lint.lint_listing(spec_of(lint, "k_wconv4"), "\n".join(doctored), notes)["_Z8k_wconv4ILi0EEv6WConvP"]
    assert any("inside the K loop" in e for e in errs), errs


def test_quantile_load_block_without_its_wait_is_caught(lint):
    """k_quantile_mask_multi issues its V4 + 1 loads AND their s_waitcnt vmcnt(0) in one asm statement; an instruction between the last
    load and the wait (a compiler copy of a destination register would sit there) and a split block are each reported."""
    with tempfile.TemporaryDirectory() as tmp:
        dis, notes = lint.disassemble(os.path.join(lint.CSRC, "sampler.o"), tmp)
    lines, keep = [], False
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
        if m:
            keep = m.group(1).startswith("_Z21k_quantile_mask_multiILi6E")
        if keep:
            lines.append(ln)
    spec = spec_of(lint, "k_quantile_mask_multi")
    res = lint.lint_listing(spec, "\n".join(lines), notes)
    assert list(res.values()) == [[]]
    loads = [k for k, ln in enumerate(lines) if "global_load_dwordx4" in ln]
    last = [k for k in loads if k + 1 not in loads][0]                   # the last load of the first run
    assert "s_waitcnt" in lines[last + 1] and "vmcnt(0)" in lines[last + 1]
    addr = re.search(r"//\s*([0-9A-Fa-f]+):", lines[last]).group(1)
    moved = lines[:last + 1] + [f"\tv_mov_b32_e32 v1, v2                                        // {addr}: 00000000"] + lines[last + 1:]
    errs = list(lint.lint_listing(spec, "\n".join(moved), notes).values())[0]
    assert any("instead of its s_waitcnt vmcnt(0)" in e for e in errs), errs
    split = lines[:last] + [f"\tv_mov_b32_e32 v1, v2                                        // {addr}: 00000000"] + lines[last:]
    errs = list(lint.lint_listing(spec, "\n".join(split), notes).values())[0]
    assert any("load block of 6" in e for e in errs), errs


def test_scratch_reload_in_the_epilogue_is_caught(lint, wconv4_listing):
    """A spill reload after the K loop (round 5: __shfl_xor's hoisted lane id came back from scratch behind vmcnt(0) after the 64 stores of every
    k_wconv4 tile) is reported."""
    lines, notes = wconv4_listing
    last = max(k for k, ln in enumerate(lines) if "v_mfma_f32_16x16x32_f16" in ln)
    addr = re.search(r"//\s*([0-9A-Fa-f]+):", lines[last + 5]).group(1)
    doctored = lines[:last + 5] + [f"\tscratch_load_dword v144, off, off offset:32                // {addr}: DC504020"] + lines[last + 5:]
    errs = list(lint.lint_listing(spec_of(lint, "k_wconv4"), "\n".join(doctored), notes).values())[0]
    assert any("scratch_load instruction(s) after the last" in e for e in errs), errs


def test_performance_shaped_findings_warn_but_do_not_fail(lint, wconv4_listing):
    """register ceilings, MFMA counts per loop body and the set of wait immediates may move with a compiler update without
    anything being wrong - they are reported with the '[perf]' prefix and build() does not fail on them; a hazard still does."""
    lines, notes = wconv4_listing
    spec = dict(spec_of(lint, "k_wconv4"), regs=256, mfma=("v_mfma_f32_16x16x32_f16", {1}))
    res = lint.lint_listing(spec, "\n".join(lines), notes)
    errs = [e for v in res.values() for e in v]
    assert any("vgpr_count" in e and lint.PERF in e for e in errs), errs
    assert any("K loop holds" in e and lint.PERF in e for e in errs), errs
    assert lint.fatal(errs) == []
    mixed = errs + ["3 wait-state fences (s_nop 15 / s_nop 3), expected 12"]
    assert lint.fatal(mixed) == mixed[-1:]
    assert os.path.exists(os.path.join(lint.LLVM, "llvm-objdump"))          # resolved from the toolchain in use, not hard-coded

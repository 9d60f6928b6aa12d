#!/usr/bin/env python3
"""ISA lint of the hand-scheduled gfx950 kernels - runs WITHOUT a GPU.

    python scripts/isa_lint.py [--json profiles/rNN_isa_summary.json] [--objdir DIR] [-v]

The kernels below rest on properties of the EMITTED code that no source-level check sees and that, until now, only wrong results on the
GPU revealed (profiles/LOG.md: both k_wconv4 miscompiles of round 4): zero spill traffic and zero register shuffling inside the K loop,
an exact number of matrix-core instructions per loop body, the counted `s_waitcnt vmcnt(N)` immediates the source intends, enough
always-issued stores behind a tile's prologue DMA for the tile-start wait to be safe on EVERY path, and - for the inline-asm MFMAs of
k_wconv4, which the compiler cannot see - wait states between a matrix-core write and the next ordinary access of that register.
This script disassembles the gfx950 code objects embedded in csrc/*.o (llvm-objdump --offloading / -d, llvm-readelf --notes) and asserts,
per kernel symbol, the rules in SPECS.  `__graft_entry__.build()` runs it after every build; tests/test_isa_lint.py runs it on the CPU
and checks that the rules BITE (a deleted fence, a branch around an epilogue store and a spill inside the K loop are each caught on
doctored listings).  It also writes the per-kernel register / scratch table and the compiler version (profiles/rNN_isa_summary.json).
"""
import argparse
import heapq
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diffusion-uncertainty_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
PERF = "[perf] "          # prefix of the warn-only findings (register / MFMA counts, allowed wait immediates): see fatal()


def _llvm_bin():
    """Directory of llvm-objdump / llvm-readelf: $DUA_LLVM_BIN, else next to the hipcc in use (<rocm>/lib/llvm/bin), else on PATH
    (a ROCm point release or another install layout must not break build())."""
    cands = [os.environ.get("DUA_LLVM_BIN")]
    hip = shutil.which(HIPCC) or HIPCC
    if os.path.exists(hip):
        rocm = os.path.dirname(os.path.dirname(os.path.realpath(hip)))
        cands += [os.path.join(rocm, "lib", "llvm", "bin"), os.path.join(rocm, "llvm", "bin")]
    cands.append("/opt/rocm/lib/llvm/bin")
    w = shutil.which("llvm-objdump")
    if w:
        cands.append(os.path.dirname(w))
    for c in cands:
        if c and os.path.exists(os.path.join(c, "llvm-objdump")) and os.path.exists(os.path.join(c, "llvm-readelf")):
            return c
    raise FileNotFoundError("isa_lint: llvm-objdump / llvm-readelf not found (set DUA_LLVM_BIN to the directory that holds them)")


LLVM = _llvm_bin()


def fatal(failures):
    """The findings that fail a build: hazards, tail waits, spills / register shuffling inside the K loop, missing counted waits, fences,
    load blocks, missing objects / symbols.  Register ceilings, MFMA counts per loop body and the set of allowed wait immediates are
    performance-shaped (a compiler update may move them without breaking anything): reported, prefixed '[perf]', not fatal."""
    return [f for f in failures if PERF not in f]

# Per kernel family: `sym` regex over the mangled names of one object; `regs` = upper bound of vgpr_count (unified VGPR + AGPR file);
# `scratch` = upper bound of private_segment_fixed_size in bytes (spills OUTSIDE the K loop are tolerated up to the recorded amount);
# `mfma` = (mnemonic, instructions per K-loop body: every allowed value); `loop_vmcnt` = the immediates of the s_waitcnt vmcnt(N) allowed
# in the K loop / `loop_vmcnt_need` = those that must be there; `tail_wait` = N of the tile-start wait `s_waitcnt vmcnt(N)` outside the K
# loop: on every path from a DMA (`... lds`) to that wait at least N stores must issue; `hazard` = wait states required between an
# (inline-asm) MFMA write and any other instruction touching the register; `fences` = exact count of the `s_nop 15 / s_nop 3` blocks.
SPECS = [
    dict(name="k_wconv4", obj="conv_wino4_f16x3.o", sym=r"^_Z8k_wconv4ILi[0-3]EEv6WConvP$", regs=512, scratch=128,
         mfma=("v_mfma_f32_16x16x32_f16", {864}), loop_vmcnt={0, 4, 6, 8}, loop_vmcnt_need={6, 8}, tail_wait=63, hazard=18, fences=12, tail_scratch_loads=0),
    dict(name="k_wconv", obj="conv_wino_f16x3.o", sym=r"^_Z7k_wconv6WConvP$", regs=512, scratch=0,
         mfma=("v_mfma_f32_16x16x32_f16", None), loop_vmcnt=None, loop_vmcnt_need=set(), tail_wait=16),
    dict(name="k_wconv1", obj="conv_wino_f16x3.o", sym=r"^_Z8k_wconv16WConvP$", regs=512, scratch=0,
         mfma=("v_mfma_f32_16x16x32_f16", None), loop_vmcnt=None, loop_vmcnt_need=set(), tail_wait=16),
    # (k_hgemm: no tail_wait rule - its stream-K dump sits behind compiler-generated flag blocks whose correlated branches a path-insensitive
    # walk cannot follow; HG_TAIL_WAIT is covered by the static_assert on the store count in the source.  Round 5: this lint found a
    # reload of a spilled LDS address, behind an s_waitcnt vmcnt(0), inside the K loop of the 256-column GELU -> planes instantiation
    # (U-ViT fc1); its epilogue now requests the column scales / biases one pair ahead instead of all 64 registers up front.)
    dict(name="k_hgemm", obj="gemm_f16x3.o", sym=r"^_Z7k_hgemmILi[234]ELi[01]EEv6HGemmP$", regs=256, scratch=64,
         mfma=("v_mfma_f32_16x16x32_f16", {6 * 48, 6 * 72, 6 * 96}), loop_vmcnt={4, 6, 7}, loop_vmcnt_need=set(), tail_wait=None, tail_scratch_loads=2),
    dict(name="k_sgemm", obj="gemm_skip_f16x3.o", sym=r"^_Z7k_sgemmILi[24]ELb[01]EEv6SGemmP$", regs=256, scratch=0,
         mfma=("v_mfma_f32_16x16x32_f16", {48, 96}), loop_vmcnt={4, 6}, loop_vmcnt_need=set(), tail_wait=8, tail_scratch_loads=0),
    dict(name="k_wconv432", obj="conv_wino4_f32.o", sym=r"^_Z10k_wconv4326WConvP$", regs=256, scratch=128,
         mfma=("v_mfma_f32_16x16x4_f32", {9 * 64}), loop_vmcnt={2, 4, 6}, loop_vmcnt_need={2, 4, 6}, tail_wait=16),
    dict(name="k_wconv32", obj="conv_wino_f32.o", sym=r"^_Z9k_wconv326WConvP$", regs=256, scratch=192,
         mfma=("v_mfma_f32_16x16x4_f32", {6 * 128}), loop_vmcnt={4, 6, 8}, loop_vmcnt_need=set(), tail_wait=16),
    # (no matrix cores: 1024 threads = 128 registers, no scratch, and the asm statement that holds its V4 + 1 loads AND their wait)
    dict(name="k_quantile_mask_multi", obj="sampler.o", sym=r"^_Z21k_quantile_mask_multiILi[3468]EEv", regs=128, scratch=0, mfma=None,
         load_block={4, 5, 7, 9}),
    dict(name="k_xtile<f16x3>", obj="mfma_bf16x3.o", sym=r"^_Z7k_xtileILi0ELb[01]ELb0ELb0ELb1EEv6XTileP$", regs=256, scratch=0,
         mfma=("v_mfma_f32_32x32x16_f16", {48}), loop_vmcnt=set(range(8, 16)), loop_vmcnt_need=set(), tail_wait=None),
]

INSN = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
REG = re.compile(r"\b([av])\[(\d+):(\d+)\]|\b([av])(\d+)\b")


class Insn:
    __slots__ = ("addr", "op", "args", "target")

    def __init__(self, addr, op, args):
        self.addr, self.op, self.args, self.target = addr, op, args, None


def extract_code_object(obj, tmp):
    """The gfx950 code object embedded in a host .o (hipcc -c): llvm-objdump --offloading writes it next to the input, so work on a copy."""
    dst = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, dst)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", dst], check=True, capture_output=True, cwd=tmp)
    cands = [f for f in os.listdir(tmp) if f.startswith(os.path.basename(obj) + ".") and "gfx950" in f]
    if not cands:
        raise RuntimeError(f"no gfx950 code object in {obj}")
    return os.path.join(tmp, cands[0])


def parse_disassembly(text):
    """{symbol: [Insn]}; branch targets resolved to addresses (simm16 counts dwords from the next instruction)."""
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        m = INSN.match(line)
        if not m:
            continue
        cur.append(Insn(int(m.group(3), 16), m.group(1), m.group(2)))
    for insns in kernels.values():
        for k, i in enumerate(insns):
            if i.op.startswith("s_cbranch") or i.op == "s_branch":
                imm = int(i.args.split()[0])
                if imm >= 0x8000:
                    imm -= 0x10000
                i.target = i.addr + 4 + 4 * imm
    return kernels


def parse_metadata(notes):
    """{symbol: {vgpr_count, agpr_count, private_segment_fixed_size, sgpr_spill_count, vgpr_spill_count, sgpr_count}} from llvm-readelf --notes."""
    out, cur = {}, {}
    keys = ("agpr_count", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "sgpr_spill_count", "vgpr_spill_count", "group_segment_fixed_size")
    for line in notes.splitlines():
        s = line.strip()
        if s.startswith("- .agpr_count:") or s.startswith("- .args:"):
            if cur.get("name"):
                out[cur["name"]] = cur
            cur = {}
            s = s[2:]
        m = re.match(r"\.(\w+):\s+(\S+)", s)
        if m and (m.group(1) in keys):
            cur[m.group(1)] = int(m.group(2))
        elif m and m.group(1) == "name" and m.group(2).startswith("_Z"):
            cur["name"] = m.group(2)
    if cur.get("name"):
        out[cur["name"]] = cur
    return out


def regs_of(args):
    s = set()
    for m in REG.finditer(args):
        if m.group(1):
            s.update((m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            s.add((m.group(4), int(m.group(5))))
    return s


def innermost_mfma_loop(insns, mfma):
    """(lo, hi, n) instruction indices of the K loop: among the backward-branch regions that hold matrix-core instructions and contain no
    smaller such region (innermost loops), the one with the most of them (the steady-state body; tail code has shorter innermost loops)."""
    index = {i.addr: k for k, i in enumerate(insns)}
    pos = [k for k, i in enumerate(insns) if i.op == mfma]
    regions = []
    for k, i in enumerate(insns):
        if i.target is not None and i.target <= i.addr and i.target in index:
            lo = index[i.target]
            n = sum(1 for q in pos if lo <= q <= k)
            if n:
                regions.append((lo, k, n))
    inner = [r for r in regions if not any((o[0] >= r[0] and o[1] <= r[1] and (o[1] - o[0]) < (r[1] - r[0])) for o in regions)]
    return max(inner, key=lambda r: r[2]) if inner else None


def min_stores_before(insns, wait_idx, is_store, is_dma):
    """{DMA instruction index: fewest stores on any path from it to the wait} (Dijkstra over the reversed instruction graph; a path ends at
    the first DMA it meets)."""
    index = {i.addr: k for k, i in enumerate(insns)}
    preds = [[] for _ in insns]
    for k, i in enumerate(insns):
        uncond = i.op == "s_branch" or i.op == "s_endpgm"
        if not uncond and k + 1 < len(insns):
            preds[k + 1].append(k)
        # (exec-mask skips - s_cbranch_execz / execnz over a predicated region - are NOT followed: the compiler lowers `ok ? store(o) :
        # store(out of range)` to two complementary masked halves, each behind such a skip, and at run time at least one of them
        # issues; what the rule is after are SCALAR branches - uniform conditions - around stores)
        if i.target is not None and i.target in index and not i.op.startswith("s_cbranch_exec"):
            preds[index[i.target]].append(k)
    dist = {wait_idx: 0}
    heap = [(0, wait_idx)]
    found = {}
    while heap:
        d, k = heapq.heappop(heap)
        if d > dist.get(k, 1 << 30):
            continue
        for p in preds[k]:
            if is_dma(insns[p]):
                found[p] = min(found.get(p, 1 << 30), d)
                continue
            nd = d + (1 if is_store(insns[p]) else 0)
            if nd < dist.get(p, 1 << 30):
                dist[p] = nd
                heapq.heappush(heap, (nd, p))
    return found


def check_tail_wait(insns, w, N):
    """The tile-start wait `s_waitcnt vmcnt(N)` is safe when every DMA that can be in flight at it is followed by at least N stores.  The
    analysis is path-insensitive: the FIRST tile's prologue (one run of DMA instructions right in front of the wait) reaches this wait
    statically with no store in between, although its flag sends it to the vmcnt(0) twin at run time.  So exactly ONE run of DMA
    instructions may sit closer than N stores - with none at all; every other DMA (the K loop's, the prefetch of the next tile in front of the
    epilogue) must have N stores on every path.  A branch around epilogue stores, or around the whole epilogue, adds a second such run."""
    found = min_stores_before(insns, w, lambda i: i.op.startswith("buffer_store") or i.op.startswith("global_store"),
                              lambda i: i.op.startswith("buffer_load") and i.args.rstrip().endswith("lds"))
    short = sorted(k for k, d in found.items() if d < N)
    runs = []
    for k in short:
        if runs and k - runs[-1][-1] < 40:
            runs[-1].append(k)
        else:
            runs.append([k])
    errs = []
    if len(runs) > 1 or any(found[k] != 0 for r in runs for k in r):
        worst = min((found[k], k) for k in short if not (len(runs) == 1 and found[k] == 0)) if short else None
        bad = [r for r in runs if any(found[k] != 0 for k in r)] or runs[1:]
        d, k = min((found[k], k) for r in bad for k in r)
        errs.append(f"a DMA at {insns[k].addr:#x} reaches the tile-start wait vmcnt({N}) at {insns[w].addr:#x} behind only {d} store(s) on some path "
                    f"({len(runs)} runs of DMA instructions closer than {N} stores; one - the first tile's prologue, with none - is expected)")
    return errs


def check_load_block(insns, lengths):
    """The inline-asm load block of a kernel without a K loop (k_quantile_mask_multi): its global_load_dwordx4 run - the first run of two
    or more in the kernel - has one of the expected lengths and the very next instruction is the s_waitcnt vmcnt(0) of the same asm
    statement: nothing the compiler placed can read a destination register before its data has arrived."""
    k = 0
    while k < len(insns):
        if insns[k].op == "global_load_dwordx4":
            e = k
            while e + 1 < len(insns) and insns[e + 1].op == "global_load_dwordx4":
                e += 1
            if e > k:
                n = e - k + 1
                errs = [] if n in lengths else [f"load block of {n} global_load_dwordx4 at {insns[k].addr:#x}, expected one of {sorted(lengths)}"]
                nxt = insns[e + 1] if e + 1 < len(insns) else None
                if nxt is None or nxt.op != "s_waitcnt" or "vmcnt(0)" not in nxt.args:
                    errs.append(f"the load block at {insns[k].addr:#x} is followed by {nxt.op if nxt else 'nothing'} instead of its s_waitcnt vmcnt(0)")
                return errs
            k = e
        k += 1
    return ["no block of consecutive global_load_dwordx4 found"]


def lint_kernel(spec, sym, insns, meta):
    errs = []
    if meta:
        if meta.get("vgpr_count", 0) > spec["regs"]:
            errs.append(PERF + f"vgpr_count {meta.get('vgpr_count')} > {spec['regs']}")
        if meta.get("private_segment_fixed_size", 0) > spec["scratch"]:
            errs.append(PERF + f"scratch {meta.get('private_segment_fixed_size')} B per lane > {spec['scratch']} (new spills: check they are outside the K loop and raise the bound)")
    if spec["mfma"] is None:                # no matrix-core loop: only the rules named in the spec
        if spec.get("load_block"):
            errs += check_load_block(insns, spec["load_block"])
        return errs
    mfma_op, counts = spec["mfma"]
    loop = innermost_mfma_loop(insns, mfma_op)
    if loop is None:
        return errs + [f"no K loop with {mfma_op} found"]
    lo, hi, n = loop
    body = insns[lo:hi + 1]
    if counts is not None and n not in counts:
        errs.append(PERF + f"K loop holds {n} {mfma_op}, expected one of {sorted(counts)}")
    bad = [i for i in body if i.op.startswith("scratch_") or i.op.startswith("v_accvgpr")]
    if len(bad) > spec.get("max_loop_spills", 0):
        errs.append(f"{len(bad)} spill / accumulator-move instruction(s) inside the K loop (first: {bad[0].op} at {bad[0].addr:#x})")
    vm = [int(m.group(1)) for i in body if i.op == "s_waitcnt" for m in [re.search(r"vmcnt\((\d+)\)", i.args)] if m]
    if spec["loop_vmcnt"] is not None:
        extra = sorted(set(vm) - spec["loop_vmcnt"])
        if extra:
            errs.append(PERF + f"s_waitcnt vmcnt({extra}) inside the K loop (allowed: {sorted(spec['loop_vmcnt'])}) - a compiler-inserted wait drains the DMA queue")
    missing = sorted(spec["loop_vmcnt_need"] - set(vm))
    if missing:
        errs.append(f"the K loop lacks its counted waits vmcnt({missing})")
    if spec.get("tail_scratch_loads") is not None:
        # a reload from scratch after the K loop waits on vmcnt - which retires in order: it drains the next item's prologue DMA and every
        # output store in flight (round 5: a spilled lane id re-loaded after the 64 stores of every k_wconv4 tile)
        last = max(k for k, i in enumerate(insns) if i.op == mfma_op)
        n_sl = sum(1 for i in insns[last + 1:] if i.op.startswith("scratch_load"))
        if n_sl > spec["tail_scratch_loads"]:
            errs.append(f"{n_sl} scratch_load instruction(s) after the last {mfma_op} (allowed: {spec['tail_scratch_loads']}): a spill reload in the epilogue "
                        "is an s_waitcnt vmcnt that drains the stores in flight")
    if spec.get("tail_wait") is not None:
        N = spec["tail_wait"]
        waits = [k for k, i in enumerate(insns) if i.op == "s_waitcnt" and re.search(rf"vmcnt\({N}\)", i.args) and not (lo <= k <= hi)]
        if not waits:
            errs.append(f"tile-start wait s_waitcnt vmcnt({N}) not found outside the K loop")
        for w in waits:
            errs += check_tail_wait(insns, w, N)
    if spec.get("fences") is not None:
        nf = sum(1 for k in range(len(insns) - 1) if insns[k].op == "s_nop" and insns[k].args.startswith("15") and insns[k + 1].op == "s_nop"
                 and insns[k + 1].args.startswith("3"))
        if nf != spec["fences"]:
            errs.append(f"{nf} wait-state fences (s_nop 15 / s_nop 3), expected {spec['fences']}")
    if spec.get("hazard"):
        need = spec["hazard"]
        last = {}                       # register -> wait-state clock of the last MFMA write
        clock = 0
        worst = None
        for i in insns:
            if i.op == mfma_op:
                dst = regs_of(i.args.split(",")[0])
                clock += 1
                for r in dst:
                    last[r] = clock
                continue
            if i.op == "s_nop":
                clock += int(i.args.split()[0]) + 1
                continue
            touched = regs_of(i.args) if (i.op.startswith("v_") or i.op.startswith("ds_") or i.op.startswith("buffer_") or i.op.startswith("global_")
                                          or i.op.startswith("scratch_")) else ()
            for r in touched:
                if r in last and clock - last[r] < need:
                    gap = clock - last[r]
                    if worst is None or gap < worst[0]:
                        worst = (gap, i.op, i.addr, r)
            for r in touched:
                last.pop(r, None)
            clock += 1
        if worst:
            errs.append(f"{worst[1]} at {worst[2]:#x} touches {worst[3][0]}{worst[3][1]} {worst[0]} wait states after a matrix-core write (need {need}): "
                        "a compiler-generated copy next to an inline-asm MFMA")
    return errs


def lint_listing(spec, text, notes=""):
    """Lint one disassembly listing (used by the tests on doctored text)."""
    kernels, meta = parse_disassembly(text), parse_metadata(notes)
    out = {}
    for sym, insns in kernels.items():
        if re.match(spec["sym"], sym):
            out[sym] = lint_kernel(spec, sym, insns, meta.get(sym))
    return out


def disassemble(objpath, tmp):
    co = extract_code_object(objpath, tmp)
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    return dis, notes


def run(objdir=CSRC, verbose=False, json_path=None):
    failures, table = [], {}
    cache = {}
    with tempfile.TemporaryDirectory() as tmp:
        for spec in SPECS:
            obj = os.path.join(objdir, spec["obj"])
            if not os.path.exists(obj):
                failures.append(f"{spec['name']}: {obj} missing (build first)")
                continue
            if obj not in cache:
                sub = os.path.join(tmp, os.path.basename(obj) + ".d")
                os.makedirs(sub)
                cache[obj] = disassemble(obj, sub)
            dis, notes = cache[obj]
            res = lint_listing(spec, dis, notes)
            if not res:
                failures.append(f"{spec['name']}: no symbol matches {spec['sym']}")
            for sym, errs in res.items():
                for e in errs:
                    failures.append(f"{spec['name']} [{sym}]: {e}")
                if verbose:
                    print(("FAIL " if errs else "ok   ") + sym)
        for obj, (dis, notes) in cache.items():
            for sym, m in parse_metadata(notes).items():
                table[sym] = {k: v for k, v in m.items() if k != "name"}
        if json_path:                                    # the record covers EVERY translation unit, not only the linted ones
            for f in sorted(os.listdir(objdir)):
                obj = os.path.join(objdir, f)
                if not f.endswith(".o") or obj in cache:
                    continue
                sub = os.path.join(tmp, f + ".d")
                os.makedirs(sub)
                try:
                    co = extract_code_object(obj, sub)
                    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
                except Exception as e:                   # (an object without device code)
                    continue
                for sym, m in parse_metadata(notes).items():
                    table[sym] = {k: v for k, v in m.items() if k != "name"}
    if json_path:
        ver = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.strip().splitlines()
        with open(json_path, "w") as f:
            json.dump({"compiler": ver, "flags": "--offload-arch=gfx950 -O3 -std=c++17", "lint_failures": failures, "kernels": table}, f, indent=1, sort_keys=True)
    return failures


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objdir", default=CSRC)
    ap.add_argument("--json", default="")
    ap.add_argument("-v", action="store_true")
    a = ap.parse_args()
    f = run(a.objdir, a.v, a.json or None)
    for e in f:
        print("isa_lint:", e)
    bad = fatal(f)
    print(f"isa_lint: {'FAILED (' + str(len(bad)) + ' fatal)' if bad else 'ok'}" + (f", {len(f) - len(bad)} performance-shaped finding(s)" if len(f) > len(bad) else ""))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

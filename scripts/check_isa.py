#!/usr/bin/env python3
"""Build-time guard for the hand-kept memory-pipeline tricks.

The table writer, the reducer and the tile-streamer sampler depend on things the compiler's
model does not cover: stores and LDS exchanges issued from inline asm, load waits that count the
YOUNGER loads instead of draining (`s_waitcnt vmcnt(N)`, N > 0), non-temporal stores.  A ROCm
bump that re-serialises one of their inner loops would still pass every parity test -- and cost
20-40 %.  This script disassembles the gfx950 code objects inside libf360.so and fails when

  * a loop of sat_write_kernel that stores table rows contains `s_waitcnt vmcnt(0)`, or
    sat_reduce_kernel has no row-batch loop left that waits with counted vmcnt only;
  * a table store of sat_write_kernel lost its `nt` bit;
  * the row loop of sample_rect_stream_kernel lost its LDS-direct loads, waits with vmcnt(0), or
    no longer waits with a counted vmcnt at all;
  * any kernel contains v_ashr_pk_u8_i32 (hipcc 7.2 packs bytes wrongly around it).

The byte-packing rule and the sc1 bits of the strip walker's hand-off guard RESULTS and always
fail; the others guard speed: they are printed as warnings (so a toolchain change cannot take the
correctness tests down with the build) and fail only with --strict, which the CPU-tier test
tests/test_isa_guard.py passes.

    python scripts/check_isa.py [--strict] [path/to/libf360.so]      (exit 0 = rules hold)
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find_objdump():
    """llvm-objdump of the ROCm toolchain: $ROCM_PATH, the default prefix, then PATH."""
    for root in (os.environ.get("ROCM_PATH"), os.environ.get("ROCM_HOME"), "/opt/rocm"):
        if root:
            cand = os.path.join(root, "lib", "llvm", "bin", "llvm-objdump")
            if os.path.exists(cand):
                return cand
    return shutil.which("llvm-objdump")


OBJDUMP = find_objdump()


def code_objects(lib):
    tmp = tempfile.mkdtemp(prefix="f360_isa_")
    shutil.copy(lib, os.path.join(tmp, "lib.so"))
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, capture_output=True, check=True)
    return tmp, [os.path.join(tmp, f) for f in sorted(os.listdir(tmp)) if f.endswith("gfx950")]


def functions(path):
    """{name: [(addr, text, branch_target_or_None)]}"""
    out = subprocess.run([OBJDUMP, "-d", path], capture_output=True, text=True, check=True).stdout
    funcs, cur, base = {}, None, 0
    for line in out.splitlines():
        m = re.match(r"^([0-9a-f]{16}) <(.+)>:$", line)
        if m:
            base, cur = int(m.group(1), 16), m.group(2)
            funcs[cur] = []
            continue
        m = re.match(r"^\t(.+?)\s*// ([0-9A-F]{12}):", line)
        if not (m and cur):
            continue
        text, addr = m.group(1).strip(), int(m.group(2), 16)
        tgt = None
        if text.startswith(("s_cbranch", "s_branch")):
            t = re.search(r"<.+\+0x([0-9a-f]+)>\s*$", line)
            tgt = base + int(t.group(1), 16) if t else (base if line.rstrip().endswith(">") else None)
        funcs[cur].append((addr, text, tgt))
    return funcs


def loops(ins):
    """Instruction slices of every INNERMOST loop (a backward branch and everything up to its
    target, with no other loop inside)."""
    index = {a: i for i, (a, _, _) in enumerate(ins)}
    spans = [(index[tgt], i) for i, (a, _, tgt) in enumerate(ins)
             if tgt is not None and tgt <= a and tgt in index]
    inner = [(lo, hi) for lo, hi in spans
             if not any((l2, h2) != (lo, hi) and lo <= l2 and h2 <= hi for l2, h2 in spans)]
    return [ins[lo:hi + 1] for lo, hi in inner]


def all_loops(ins):
    """Instruction slices of EVERY loop (backward branch .. its target), outer ones included."""
    index = {a: i for i, (a, _, _) in enumerate(ins)}
    return [ins[index[tgt]:i + 1] for i, (a, _, tgt) in enumerate(ins)
            if tgt is not None and tgt <= a and tgt in index]


def regs_of(operand):
    """Register numbers an operand token names: 's[10:17]' -> {('s', 10) .. ('s', 17)}."""
    m = re.match(r"^([sv])\[(\d+):(\d+)\]$", operand)
    if m:
        return {(m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"^([sv])(\d+)$", operand)
    return {(m.group(1), int(m.group(2)))} if m else set()


def inflight_violations(ins):
    """LDS / scalar-memory loads whose destination registers are touched before a wait has covered
    them (the strip walker issues some of its loads and their waits from SEPARATE asm
    statements -- the plan words' s_load_dwordx8, the first look at a mailbox word -- so nothing
    but this check stops a future compiler from copying or spilling a register that is still in
    flight).  Straight-line model: loads enter a queue in program order; `s_waitcnt lgkmcnt(N)`
    retires all but the youngest N when only LDS operations are pending (they return in order) and
    everything when N is 0 (scalar loads return out of order, so any other count retires none of
    them); a branch target or a branch empties the queue conservatively (the compiler's and the asm
    statements' own waits sit in front of those).  Returns [(load text, offending text)]."""
    pending, bad = [], []   # pending: (kind 'lds' | 'smem', regs, text)
    targets = {t for _, _, t in ins if t is not None}
    for addr, text, tgt in ins:
        if addr in targets or tgt is not None:
            pending = []
        ops = [o.strip() for o in re.split(r"[ ,]+", text)[1:]]
        if text.startswith("s_waitcnt"):
            m = re.search(r"lgkmcnt\((\d+)\)", text)
            if m:
                n = int(m.group(1))
                if n == 0:
                    pending = []
                elif all(k == "lds" for k, _, _ in pending):
                    pending = pending[-n:] if n < len(pending) else pending
            continue
        touched = set()
        for o in ops:
            touched |= regs_of(o)
        for kind, regs, ltext in pending:
            if regs & touched:
                bad.append((ltext, text))
        if text.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle")) and ops:
            pending.append(("lds", regs_of(ops[0]), text))
        elif text.startswith(("s_load_", "s_buffer_load")) and ops:
            pending.append(("smem", regs_of(ops[0]), text))
        elif text.startswith(("ds_write", "ds_add", "ds_cmpst", "ds_wrxchg", "ds_max", "ds_min", "ds_or", "ds_and")):
            pending.append(("lds", set(), text))   # (counted by lgkmcnt, no destination)
        elif text.startswith("s_memtime") or text.startswith("s_memrealtime"):
            pending.append(("smem", regs_of(ops[0]) if ops else set(), text))
    return bad


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    strict = "--strict" in sys.argv[1:]
    lib = args[0] if args else os.path.join(REPO, "foveated-360-video_amd", "lib", "libf360.so")
    if not OBJDUMP:
        print("check_isa: skipped (no llvm-objdump under $ROCM_PATH, /opt/rocm or on PATH)")
        return 0
    tmp, objs = code_objects(lib)
    # `errors` guard results (a known miscompile); `perf` guards speed only: a ROCm bump or a
    # harmless scheduling change must not fail the build and with it every correctness test, so
    # build() prints them as warnings and only --strict (tests/test_isa_guard.py) fails on them
    errors, perf, seen = [], [], set()
    try:
        for obj in objs:
            for name, ins in functions(obj).items():
                texts = [t for _, t, _ in ins]
                # the table writer proper: RGB0 / planar sources (1, 2, 3), LDS-staged stores (1);
                # the byte-source fallback and the fused path's emit mode are built differently
                if re.search(r"sat_write_kernelILi[123]ELi1EE", name):
                    seen.add("writer")
                    stores = [t for t in texts if t.startswith("global_store_dwordx4")]
                    if any(" nt" not in t for t in stores):
                        perf.append(f"{name}: a table store lost its nt bit")
                    for lp in loops(ins):
                        lt = [t for _, t, _ in lp]
                        if any(t.startswith("global_store_dwordx") for t in lt) and \
                                any(t.startswith("s_waitcnt vmcnt(0)") for t in lt):
                            perf.append(f"{name}: s_waitcnt vmcnt(0) inside a storing loop "
                                        f"({len(lt)} instructions)")
                # the band writer's one pass (sat_band_fuse.hip): the table writer's rules -- nt
                # table stores, no drain inside a storing loop, no scratch -- and nt pixel stores
                if "sat_write_fuse_kernel" in name:
                    seen.add("one-pass writer")
                    if any(t.startswith(("scratch_", "s_swappc")) for t in texts):
                        perf.append(f"{name}: scratch memory or a call in the one-pass writer")
                    stores = [t for t in texts if t.startswith("global_store_dwordx4")]
                    if len(stores) < 24 or any(" nt" not in t for t in stores):
                        perf.append(f"{name}: a table store lost its nt bit")
                    if any(" nt" not in t for t in texts if t.startswith(("global_store_short", "global_store_byte"))):
                        perf.append(f"{name}: a pixel store lost its nt bit")
                    # the row loop = the smallest loop that holds a double batch's table stores
                    # (cold blocks placed behind the loop jump back into the prologue: such a
                    # "loop" spans the whole kernel and is not the one meant)
                    cand = [[t for _, t, _ in lp] for lp in all_loops(ins)]
                    cand = [lt for lt in cand if sum(t.startswith("global_store_dwordx4") for t in lt) >= 24]
                    if not cand:
                        perf.append(f"{name}: no row loop (24 table stores) found")
                    elif any(t.startswith("s_waitcnt vmcnt(0)") for t in min(cand, key=len)):
                        perf.append(f"{name}: s_waitcnt vmcnt(0) inside the row loop "
                                    f"({len(min(cand, key=len))} instructions)")
                    if not any(re.match(r"s_waitcnt vmcnt\([1-9]", t) for t in texts):
                        perf.append(f"{name}: no counted vmcnt wait left")
                    # hipcc 7.2 folds the wave-shift move of the one-column boxes into the
                    # subtraction that follows (v_subrev_u32_dpp ... wave_shr:1); that form gave
                    # wrong differences on gfx950 (parity tests, round 5); the kernel pins the move
                    if any(re.match(r"v_(sub|subrev|add)\w*_dpp .*wave_sh", t) for t in texts):
                        errors.append(f"{name}: arithmetic folded into a wave-shift DPP "
                                      f"(known-bad on gfx950; keep the v_mov_b32_dpp)")
                if re.search(r"sat_reduce_kernelILi[123]EE", name):
                    seen.add("reducer")
                    # the steady-state loop: a batch of >= 8 row loads, waits that count the
                    # younger loads, no drain (the short loop that flushes a band's row sums
                    # does drain, once per band)
                    steady = [lt for lt in ([t for _, t, _ in lp] for lp in loops(ins))
                              if sum(t.startswith("global_load_dword") for t in lt) >= 8 and
                              any(re.match(r"s_waitcnt vmcnt\([1-9]", t) for t in lt) and
                              not any(t.startswith("s_waitcnt vmcnt(0)") for t in lt)]
                    if not steady:
                        perf.append(f"{name}: no row-batch loop with counted waits only")
                # the read-once batched encoder (RGB0 and planar sources): no scratch memory, nt
                # table stores, write-through granule stores, an sc1 poll, and a steady-state
                # loop whose only drain is the slow path of a hand-off wait (it follows its own
                # sc1 load directly)
                # (every instantiation whose name starts with sat_walk: a new source id or depth
                # must not slip past the sc1 rules)
                if "sat_walk" in name:
                    seen.add("walker")
                    if any(t.startswith(("scratch_", "s_swappc")) for t in texts):
                        perf.append(f"{name}: scratch memory or a call in the strip walker")
                    # (the debug statistics behind the last table store are plain stores)
                    # (the debug statistics are the only other 16-byte stores: scalar base, plain)
                    nts = [i for i, t in enumerate(texts)
                           if t.startswith("global_store_dwordx4") and " nt" in t]
                    if len(nts) < 24 or any(t.startswith("global_store_dwordx4") and ", off" in t
                                            and " nt" not in t for t in texts):
                        perf.append(f"{name}: a table store lost its nt bit")
                    gran = [t for t in texts if t.startswith("global_store_dwordx2")]
                    if not gran or any(" sc1" not in t for t in gran):
                        errors.append(f"{name}: hand-off granule stores must be sc1 (write-through)")
                    polls = [t for t in texts if t.startswith("global_load_dwordx2")]
                    if len([t for t in polls if " sc1" in t]) < 2:
                        errors.append(f"{name}: hand-off polls must be sc1 loads")
                    # the steady-state loop = the smallest loop that holds a batch's 24 table
                    # stores AND the hand-off polls (the loop a strip finishes alone in after a
                    # timed-out hand-off has the stores but no poll: it may drain as it likes).
                    # Its only full drains are the slow paths', each directly behind its own
                    # load in one asm statement: the re-poll of a hand-off (sc1) and the source
                    # loads of a strip that finishes alone after a timed-out hand-off.
                    cand = [[t for _, t, _ in lp] for lp in all_loops(ins)]
                    cand = [lt for lt in cand
                            if sum(t.startswith("global_store_dwordx4") and " nt" in t for t in lt) >= 24
                            and any(t.startswith("global_load_dwordx2") and " sc1" in t for t in lt)]
                    if not cand:
                        perf.append(f"{name}: no steady-state loop (24 nt table stores + sc1 polls) found")
                    else:
                        body = min(cand, key=len)
                        # (a wait at the loop head, before the iteration has issued anything, is
                        # the wait for the oldest batch in flight whatever its immediate says)
                        first_vmem = next(i for i, t in enumerate(body) if t.startswith("global_"))
                        bad = [i for i, t in enumerate(body) if t.startswith("s_waitcnt vmcnt(0)")
                               and i > first_vmem
                               and not body[i - 1].startswith("global_load_")]
                        if bad:
                            perf.append(f"{name}: {len(bad)} s_waitcnt vmcnt(0) on the fast path "
                                        f"of the row loop")
                        if not any(re.match(r"s_waitcnt vmcnt\([1-9]", t) for t in body):
                            perf.append(f"{name}: the row loop has no counted vmcnt wait")
                # the one-pass encode + sample instantiation: its helper waves' row loops (the
                # loops that store reduced pixels) must never wait on the vector memory counter
                # -- the pixel stores are hidden from the compiler, and any vmcnt wait there
                # also waits for the stores of the row before (it cost 800 cycles per row when a
                # plan word was read with a vector load)
                if "sat_walk" in name or "sat_write_fuse_kernel" in name:
                    for ltext, utext in inflight_violations(ins)[:3]:
                        errors.append(f"{name}: `{utext}` touches a register `{ltext}` still has "
                                      f"in flight")
                if "sat_walk" in name and "Lb1E" in name:
                    seen.add("one-pass walker")
                    px = [[t for _, t, _ in lp] for lp in all_loops(ins)]
                    px = [lt for lt in px if any(t.startswith("global_store_short") for t in lt)
                          and not any(t.startswith("global_store_dwordx4") for t in lt)]
                    if not px:
                        perf.append(f"{name}: no helper row loop (pixel stores) found")
                    if any(" nt" not in t for lt in px for t in lt if t.startswith("global_store_short")):
                        perf.append(f"{name}: helper pixel stores lost their nt bit")
                    if any("vmcnt" in t for lt in px for t in lt):
                        perf.append(f"{name}: a vmcnt wait inside a helper row loop")
                # hipcc 7.2 miscompiles byte packing around this instruction (its upper half is
                # not zero on gfx950 but later ORs assume so): both times it appeared, the parity
                # tests failed; the kernels are written so that it is not selected
                if any(t.startswith("v_ashr_pk_u8_i32") for t in texts):
                    errors.append(f"{name}: v_ashr_pk_u8_i32 selected (known-bad byte packing)")
                if "sample_rect_stream_kernel" in name or "sample_rect_stream_batch_kernel" in name:
                    seen.add("batch streamer" if "batch" in name else "streamer")
                    row_loops = [[t for _, t, _ in lp] for lp in loops(ins)
                                 if any(t.startswith("global_load_lds_dwordx4") for _, t, _ in lp)]
                    if not row_loops:
                        perf.append(f"{name}: no loop with LDS-direct loads")
                    for lt in row_loops:
                        if any(t.startswith("s_waitcnt vmcnt(0)") for t in lt):
                            perf.append(f"{name}: s_waitcnt vmcnt(0) inside the row loop")
                        if not any(re.match(r"s_waitcnt vmcnt\([1-9]", t) for t in lt):
                            perf.append(f"{name}: the row loop has no counted vmcnt wait")
                    if not any(t.startswith("global_store_short") and " nt" in t for t in texts):
                        perf.append(f"{name}: pixel stores lost their nt bit")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for k in ("writer", "reducer", "streamer", "batch streamer", "one-pass walker", "one-pass writer"):
        if k not in seen:
            perf.append(f"no {k} kernel found in {lib}")
    # the library always exports f360_sat_encode_batch: without a recognised strip walker the
    # result-class sc1 rules above have checked nothing
    if "walker" not in seen:
        errors.append(f"no strip-walker kernel (sat_walk*) found in {lib}: the sc1 rules did not run")
    for p in perf:
        print(("check_isa: FAILED (perf rule): " if strict else "check_isa: warning (perf rule): ") + p)
    if errors:
        print("check_isa: FAILED\n  " + "\n  ".join(errors))
        return 1
    if strict and perf:
        return 1
    print(f"check_isa: ok ({len(objs)} gfx950 code objects; "
          + ("writer, reducer, strip-walker and streamer loops keep their counted waits, "
             "LDS-direct loads, nt and sc1 bits" if not perf else f"{len(perf)} perf warnings") + ")")
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""CPU tier: the built library still carries the hand-kept memory-pipeline patterns
(scripts/check_isa.py --strict).  build() runs the same script without --strict, where only the
rules that guard RESULTS fail; the speed rules fail here, in a test of their own, so that a
toolchain change that re-schedules a loop cannot take the correctness suite down with it."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "scripts"))


def test_isa_rules_hold_for_the_built_library():
    import check_isa
    lib = os.path.join(REPO, "foveated-360-video_amd", "lib", "libf360.so")
    if not os.path.exists(lib):
        pytest.skip("libf360.so not built")
    if not check_isa.find_objdump():
        pytest.skip("no llvm-objdump in this environment")
    out = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "check_isa.py"), "--strict",
                          lib], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_isa_rules_hold_where_the_kernels_run():
    """The same strict check in the GPU tier: the library the parity tests of this run load is the
    one whose hidden-store loops, counted waits, nt and sc1 bits are verified."""
    test_isa_rules_hold_for_the_built_library()


def test_build_honours_the_strict_switch(monkeypatch):
    """F360_ISA_STRICT=1 makes build() pass --strict to the guard (read from the source: a real
    build() is the driver's job)."""
    src = open(os.path.join(REPO, "__graft_entry__.py")).read()
    assert 'F360_ISA_STRICT' in src and '"--strict"' in src


def test_objdump_lookup_honours_rocm_path(tmp_path, monkeypatch):
    import check_isa
    fake = tmp_path / "lib" / "llvm" / "bin"
    fake.mkdir(parents=True)
    (fake / "llvm-objdump").write_text("#!/bin/sh\n")
    monkeypatch.setenv("ROCM_PATH", str(tmp_path))
    assert check_isa.find_objdump() == str(fake / "llvm-objdump")


def test_inflight_register_rule():
    """scripts/check_isa.py inflight_violations: a register a load still has in flight must not be
    read or written before a wait covers it (the strip walker issues loads and their waits from
    separate asm statements)."""
    import check_isa

    def prog(*texts):
        return [(4 * k, t, None) for k, t in enumerate(texts)]
    # the walker's plan words: load, unrelated work, wait, use -- fine
    assert check_isa.inflight_violations(prog(
        "s_load_dwordx8 s[8:15], s[0:1], 0x0", "v_add_u32_e32 v1, v2, v3", "s_waitcnt lgkmcnt(0)",
        "s_and_b32 s20, s9, 1")) == []
    # a copy of one of the words before the wait -- caught
    bad = check_isa.inflight_violations(prog(
        "s_load_dwordx8 s[8:15], s[0:1], 0x0", "s_mov_b32 s20, s9", "s_waitcnt lgkmcnt(0)"))
    assert len(bad) == 1 and "s_mov_b32" in bad[0][1]
    # LDS returns in order: lgkmcnt(1) covers the older of two reads, not the younger
    assert check_isa.inflight_violations(prog(
        "ds_read_b32 v4, v0", "ds_read_b32 v5, v1", "s_waitcnt lgkmcnt(1)", "v_mov_b32_e32 v6, v4")) == []
    bad = check_isa.inflight_violations(prog(
        "ds_read_b32 v4, v0", "ds_read_b32 v5, v1", "s_waitcnt lgkmcnt(1)", "v_mov_b32_e32 v6, v5"))
    assert len(bad) == 1
    # a counted wait retires no scalar load (they return out of order)
    bad = check_isa.inflight_violations(prog(
        "s_load_dword s4, s[0:1], 0x0", "ds_read_b32 v5, v1", "s_waitcnt lgkmcnt(1)", "s_add_u32 s5, s4, 1"))
    assert len(bad) == 1

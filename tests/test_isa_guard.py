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

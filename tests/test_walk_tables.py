"""CPU check of llcomp_amd/csrc/walk_tables.hpp -- the lookup tables with which the 2-D encoder's snapshot pass replays a
sample's effect on the eight adaptive states of its context (snapshot_kernels.hip: k_snap_walk).  The header is plain constexpr
C++: a small host program (tests/helpers/walk_tables_check.cpp) compares the table-driven step with the state machine replayed
over the reference's binarisation for every residual -510..510, every state and every slot."""
import os
import subprocess

from conftest import ROOT


def test_walk_tables_equal_the_state_machine_over_the_binarisation(tmp_path):
    exe = str(tmp_path / "walk_tables_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "llcomp_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "helpers", "walk_tables_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.split() == ["ok", str(1021 * 128 * 8)]

"""Guards for the hand-written gfx950 blocks of the slice kernels (csrc/enc_rows_asm.hpp, enc_sample_asm.inc,
dec_rows_asm.hpp): they pin registers by name and lean on what hipcc does around them, so every toolchain has to show that
they still produce what hipcc's own code produces from the same C++ (`make -C llcomp_amd/csrc guards`, built by
__graft_entry__.build()):

  libllcomp_mi_noasm.so   -DLLMI_ASM_ENC=0 -DLLMI_ASM_DEC=0: the same library without the blocks
  libllcomp_mi_ldsoff.so  a static LDS array displaces the model table from LDS address 0, where the blocks expect it:
                          the kernels must refuse to run (status bit kStInternal -> LLCOMP_MI_HIP_ERROR), not code garbage

Each library runs in a child process of its own (LLCOMP_MI_LIB; one HIP runtime and one build per process)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT, load_golden

CHILD = os.path.join(ROOT, "tests", "helpers", "guard_child.py")
LIBDIR = os.path.join(ROOT, "llcomp_amd")


def run_child(lib, what):
    path = os.path.join(LIBDIR, lib)
    if not os.path.exists(path):
        pytest.fail(f"{path} missing: run `make -C llcomp_amd/csrc guards` (or __graft_entry__.build())")
    env = dict(os.environ, LLCOMP_MI_LIB=path)
    r = subprocess.run([sys.executable, CHILD, what], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, f"{lib} {what}: rc {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_hand_written_blocks_equal_hipcc_code():
    """120 sliced containers (one-row slices planar and interleaved, 2-D tiles) of random shapes and the three 4K contents at
    the benchmarked slicing: containers AND decoded pixels of the product build == those of the build without the blocks;
    the 4K containers also against the golden hashes from the real reference."""
    asm = run_child("libllcomp_mi.so", "cases")
    plain = run_child("libllcomp_mi_noasm.so", "cases")
    assert asm.keys() == plain.keys() and len(asm) >= 100
    diff = [k for k in asm if asm[k] != plain[k]]
    assert not diff, f"hand-written blocks and hipcc's code disagree on {diff[:8]}"
    slc = load_golden("slice_payloads.json")["vectors"]
    for gen in ("g3", "mid", "nat"):
        v = [x for x in slc if x["w"] == 3840 and x["gen"] == gen and x["planar"] and x["tile_w"] == 480 and x["tile_h"] == 1]
        if v:  # (the benchmarked slicing is in the fixtures for the contents bench.py measures)
            assert asm[f"4k-{gen}-480x1p"][0] == v[0]["container_fnv1a64"] and asm[f"4k-{gen}-480x1p"][2] == v[0]["container_len"]


@pytest.mark.gpu
def test_model_table_off_lds_address_zero_is_refused():
    """The blocks address the model table with offsets relative to LDS address 0; a build in which something else sits
    there must not code a byte: the encoder reports an internal error (LLCOMP_MI_HIP_ERROR = 7)."""
    import llcomp_amd as mi

    got = run_child("libllcomp_mi_ldsoff.so", "refuse")
    assert got["status"] == mi.HIP_ERROR, got
    ok = run_child("libllcomp_mi.so", "refuse")
    assert ok["status"] == 0, ok

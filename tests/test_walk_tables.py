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


def test_geometry_selects_the_kernel_family(tmp_path):
    """llcomp_amd/csrc/geometry.hpp (host + device header): one-row slices keep three contexts on chip (kGeoRows = 1), a lone slice
    per wavefront keeps its table in LDS (kGeoLdsTable = 2; also chosen for up to 512 big slices), 2-D slices of at most 16384
    samples (four chunks of 4096: 128x128 planes, 64x64 interleaved RGB) that share a wavefront get the snapshot encoder
    (kGeoSnapshot = 16) unless the tuning hook forbids it, bigger 2-D slices (256x256 planes, 128x129) keep tables in HBM both ways; every family with tables in HBM and at most four channels per slice decodes through the
    bank cache in LDS (kGeoBankCache = 32) unless LLCOMP_MI_NOCACHE forbids it."""
    exe = str(tmp_path / "geometry_check")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "llcomp_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "helpers", "geometry_check.cpp")])
    rows = {ln.split()[0]: [int(x) for x in ln.split()[1:]] for ln in subprocess.check_output([exe], text=True).splitlines()}
    flags = {k: v[0] for k, v in rows.items()}
    assert flags == {"rows_4k_480x1": 1, "tiles_4k_64x64": 48, "tiles_4k_32x32_interleaved": 48, "tiles_4k_128x128": 48,
                     "tiles_4k_64x64_interleaved": 48, "tiles_4k_256x256": 32, "tiles_4k_128x129": 32,
                     "one_frame_256x256": 2, "legacy_bulk_512": 2, "lone_legacy": 2, "tiles_4k_64x64_nosnap": 32,
                     "rows_forced_general": 48, "nine_channels_one_row": 2, "tiles_4k_64x64_nocache": 16, "nine_channels_tiles": 16}
    assert rows["tiles_4k_64x64"][1:] == [6, 64, 97920, 4096]      # lane_shift, slices per wavefront, slices, samples per slice
    assert rows["legacy_bulk_512"][1:3] == [0, 1] and rows["one_frame_256x256"][1:3] == [0, 1]
    assert rows["rows_4k_480x1"][3] == 32 * 51840

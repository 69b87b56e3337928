"""Shared fixtures.  `-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol check.
`-m gpu`: parity of the HIP path (through the C ABI) against the oracle and the golden vectors."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-second CPU cases")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    import subprocess

    import orc as orc_mod

    if not os.path.exists(orc_mod.ORC_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liborc.so"])
    return orc_mod.Orc()


@pytest.fixture(scope="session")
def ref():
    import orc as orc_mod

    if not orc_mod.Ref.available():
        pytest.skip("oracle/_ref/libllcomp_ref.so not built (needs /root/reference)")
    return orc_mod.Ref()


def make_image(gen, w, h, c):
    import numpy as np
    import orc as orc_mod

    if gen == "const0":
        return np.zeros((h, w, c), np.uint8)
    if gen == "const255":
        return np.full((h, w, c), 255, np.uint8)
    if gen.startswith("g3@"):  # std::mt19937 noise with another seed (BASELINE config 5: frame i = seed 1234 + i)
        return orc_mod.GENERATORS["g3"](w, h, c, seed=int(gen[3:]))
    return orc_mod.GENERATORS[gen](w, h, c)


def fnv_hex(orc, b):
    import ctypes as C

    import numpy as np

    a = np.frombuffer(b, dtype=np.uint8)
    if a.size == 0:
        return "%016x" % 1469598103934665603
    return "%016x" % orc.lib.orc_fnv1a64(a.ctypes.data_as(C.POINTER(C.c_uint8)), a.size)

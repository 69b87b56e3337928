"""Live cross-check of the plain-C oracle against the real reference compiled in place
(oracle/_ref, only where /root/reference exists or the prebuilt .so travelled)."""
import numpy as np
import pytest


def test_state_machine_and_quantisers(orc, ref):
    for s in range(128):
        assert orc.lib.orc_state_p(s) == ref.lib.ref_state_p(s)
        for b in (0, 1):
            assert orc.lib.orc_state_next(s, b) == ref.lib.ref_state_next(s, b)
    for x in range(-700, 701):
        assert orc.lib.orc_quant11(x) == ref.lib.ref_quant11(x)
        assert orc.lib.orc_quant5(x) == ref.lib.ref_quant5(x)
    rng = np.random.default_rng(1)
    for a, b, c in rng.integers(-600, 600, size=(2000, 3)):
        assert orc.lib.orc_median(int(a), int(b), int(c)) == ref.lib.ref_median(int(a), int(b), int(c))
    for a in range(-3, 4):
        for b in range(-3, 4):
            for c in range(-3, 4):
                assert orc.lib.orc_median(a, b, c) == ref.lib.ref_median(a, b, c)


@pytest.mark.parametrize("seed", range(40))
def test_random_images_bit_exact(orc, ref, seed):
    rng = np.random.default_rng(seed)
    w, h, c = int(rng.integers(1, 68)), int(rng.integers(1, 68)), int(rng.integers(1, 5))
    kind = seed % 4
    if kind == 0:
        img = rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)
    elif kind == 1:
        img = np.full((h, w, c), int(rng.integers(0, 256)), np.uint8)
    elif kind == 2:
        y, x, k = np.meshgrid(np.arange(h), np.arange(w), np.arange(c), indexing="ij")
        img = ((x * 3 + y * 5 + k * 11 + rng.integers(-2, 3, size=(h, w, c))) & 0xFF).astype(np.uint8)
    else:
        y, x, k = np.meshgrid(np.arange(h), np.arange(w), np.arange(c), indexing="ij")
        img = (((x + y + k) & 1) * 255).astype(np.uint8)
    s = orc.compress_image(img)
    s2 = ref.o2_compress_image(img)
    assert s == s2
    s1 = ref.o1_compress_image(img, len(s2))
    if s1 is not None:
        assert s1 == s
    assert np.array_equal(orc.forward_rct(img), ref.o2_forward_rct(img))
    rc, px = orc.decompress(s)
    assert rc == 0 and np.array_equal(px, img)
    if c >= 3:
        rc, px = ref.o1_decompress_image(s)
        assert rc == 0 and np.array_equal(px, img)
    # planar plane streams == reference components on the int16 plane
    planes = orc.forward_rct(img)
    for k in range(c):
        pl = np.ascontiguousarray(planes[:, :, k:k + 1])
        assert orc.encode_samples(pl) == ref.o2_encode_samples(pl)


@pytest.mark.parametrize("seed", range(30))
def test_random_garbage_decodes_like_reference(orc, ref, seed):
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    c = int(rng.integers(3, 5))
    body = rng.integers(0, 256, size=int(rng.integers(0, 400)), dtype=np.uint8).tobytes()
    data = bytes([0x79, c, w, 0, h, 0]) + body
    rc_r, px_r = ref.o1_decompress_image(data)
    rc_o, px_o = orc.decompress(data)
    if rc_r == 0:
        assert rc_o == 0 and np.array_equal(px_o, px_r)
    else:
        assert rc_o == rc_r


@pytest.mark.parametrize("seed", range(16))
def test_small_model_random_images_bit_exact(orc, seed):
    """the LargeModel = false variant against the real header compiled with the constant flipped (oracle/Makefile)"""
    import os

    import orc as orc_mod

    if not os.path.exists(orc_mod.REF_SMALL_PATH):
        pytest.skip("oracle/_ref/libllcomp_ref_small.so not built (needs /root/reference)")
    small = orc_mod.Ref(orc_mod.REF_SMALL_PATH)
    assert small.lib.ref_large_model() == 0
    rng = np.random.default_rng(500 + seed)
    w, h, c = int(rng.integers(1, 68)), int(rng.integers(1, 68)), int(rng.integers(1, 5))
    if seed % 2:
        img = rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)
    else:
        y, x, k = np.meshgrid(np.arange(h), np.arange(w), np.arange(c), indexing="ij")
        img = ((x * 3 + y * 5 + k * 11 + rng.integers(-2, 3, size=(h, w, c))) & 0xFF).astype(np.uint8)
    orc.set_small_model(True)
    try:
        s = orc.compress_image(img)
        assert s == small.o2_compress_image(img)
        rc, px = orc.decompress(s)
        assert rc == 0 and np.array_equal(px, img)
        if c >= 3:
            rc, px = small.o1_decompress_image(s)
            assert rc == 0 and np.array_equal(px, img)
    finally:
        orc.set_small_model(False)

"""The CPU oracle against the fixtures produced by running the reference renderer
(tests/golden/make_golden.py).  Everything is compared bit for bit: the oracle repeats the
reference's float64 operations in the same order on the same libm."""
import numpy as np
import pytest

from conftest import flat_for, golden_names, load_golden


@pytest.mark.parametrize("name", golden_names())
def test_oracle_reproduces_reference(name, oracle_mod):
    g, meta = load_golden(name)
    flat = flat_for(meta["scene"])
    rng_mode = 1 if meta["rng"] == "serial" else 0
    want_nodes = "nodes" in g.files
    r = oracle_mod.render(flat, meta["options"], seed=meta["seed"], rng_mode=rng_mode, want_nodes=want_nodes)
    assert np.array_equal(r["rgb"], g["image_rgb"]), "image differs from the reference"
    assert np.array_equal(r["prim"][..., 0], g["prim_obj"])
    assert np.array_equal(r["prim"][..., 1], g["prim_face"])
    assert np.array_equal(r["prim_t"], g["prim_t"], equal_nan=True)
    st = r["stats"]
    assert st["rays_api"] == meta["rays"]
    assert st["queries"] == meta["queries"]
    assert st["tri_tests"] == meta["tri_tests"]
    assert st["sphere_tests"] == meta["sphere_tests"]
    assert st["aabb_tests"] == meta["aabb_tests"]
    if "radiance" in g.files:
        # mean over samples in the reference's summation order
        rad = g["radiance"]
        acc = np.zeros(rad.shape[:2] + (3,))
        for k in range(rad.shape[2]):
            acc = acc + rad[:, :, k]
        assert np.array_equal(acc / rad.shape[2], r["radiance"])
    if want_nodes:
        gn, on = g["nodes"], r["nodes"]
        # the reference traces (but never shades) children of an emitter hit by a primary ray when
        # options['lights'] is False; the oracle records shaded nodes only
        key = {(int(n["px"]), int(n["py"]), int(n["sample"]), int(n["heap"])): n for n in on}
        assert len(gn) == len(on)
        for row in gn:
            n = key[(int(row[0]), int(row[1]), int(row[2]), int(row[3]))]
            assert n["obj"] == row[4] and n["face"] == row[5] and n["direction"] == row[6]
            assert np.array_equal(n["point"], row[7:10])
            assert np.array_equal(n["ray_start"], row[10:13])
            assert np.array_equal(n["ray_dir"], row[13:16])


def test_known_image_digests():
    """SHA-1 prefixes of the images the reference renders for G0 and G1 (recorded when the fixtures were made)."""
    import hashlib
    for name, digest in [("g0_cornell_160x120_b0", "157f954245a5"), ("g1_cornell_160x120_b1_mt", "1caea18147a0")]:
        g, meta = load_golden(name)
        assert hashlib.sha1(g["image_rgb"].tobytes()).hexdigest().startswith(digest)
    g, meta = load_golden("g0_cornell_160x120_b0")
    assert meta["rays"] == 38994 and meta["tri_tests"] == 1402770 and meta["aabb_tests"] == 565221
    g, meta = load_golden("g1_cornell_160x120_b1_mt")
    assert meta["rays"] == 54812 and meta["queries"] == 66530


def test_rows_and_stride_do_not_change_pixels(oracle_mod):
    g, meta = load_golden("g2_cornell_80x60_b4_s2_keyed")
    flat = flat_for(meta["scene"])
    full = oracle_mod.render(flat, meta["options"], seed=meta["seed"])
    part = oracle_mod.render(flat, meta["options"], seed=meta["seed"], row0=17, nrows=9)
    assert np.array_equal(part["bgr"], full["bgr"][17:26])
    inter = oracle_mod.render(flat, meta["options"], seed=meta["seed"], row0=1, row_stride=4)
    assert np.array_equal(inter["bgr"], full["bgr"][1::4])
    assert np.array_equal(inter["radiance"], full["radiance"][1::4])

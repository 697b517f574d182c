"""The CPU restatement (oracle/rt_oracle.c, double build) against the golden vectors that the
reference's own compiled main.cc produced (oracle/gen_golden.py).  Bit-exact unless stated."""
import numpy as np
import pytest

from conftest import load_golden

PHILOX_KAT = [  # Random123 kat_vectors, philox4x32 10
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


def test_philox_known_answers(oracle):
    for ctr, key, want in PHILOX_KAT:
        assert oracle.philox(ctr, key) == want


def test_stored_scene_is_the_gxx_scene(rtiow_flat):
    # facts about the scene the g++ build of main.cc:37-85 produces
    assert rtiow_flat.n_prims == 487
    counts = np.bincount(rtiow_flat.mat_type, minlength=3)
    assert tuple(counts[:3]) == (394, 64, 29)
    np.testing.assert_allclose(rtiow_flat.prim_geom[1, :4], [-10.2952107, 0.2, -10.6450554, 0.2],
                               rtol=0, atol=1e-6)


def test_stored_scene_matches_reference_build(reference):
    from raytrace_amd import scene as S
    assert np.array_equal(reference.build_default_scene(), S.rtiow_records())


def test_camera_records(oracle):
    g = load_golden("func_vectors.npz")
    for cin, want in zip(g["cam_in"], g["cam_out"]):
        got = oracle.camera(cin[0:3], cin[3:6], cin[6:9], cin[9], cin[10], cin[11], cin[12])
        assert np.array_equal(got, want)


def test_get_ray(oracle):
    g = load_golden("func_vectors.npz")
    for (s, t, seed, pixel, sample), want in zip(g["get_ray_in"], g["get_ray_out"]):
        got, used = oracle.get_ray(g["get_ray_cam"], s, t, int(seed), int(pixel), int(sample), 2)
        assert np.array_equal(got, want)
        assert used >= 2 and used % 2 == 0


@pytest.mark.parametrize("which", ["in_unit_sphere", "unit_vector", "in_unit_disk"])
def test_random_helpers(oracle, which):
    g = load_golden("func_vectors.npz")
    for key, want, used in zip(g["samp_in"], g[f"samp_{which}"], g[f"samp_used_{which}"]):
        got, n = oracle.sample(which, *[int(k) for k in key])
        assert np.array_equal(got, want)
        assert n == used


def test_closest_hit_all_ray_kinds(oracle, rtiow_flat):
    g = load_golden("primary_rays.npz")
    got = oracle.trace(rtiow_flat, g["rays"].astype(np.float64), float(g["t_min"]))
    assert np.array_equal(got["id"], g["id"])
    for k in ("t", "p", "n", "front"):
        assert np.array_equal(got[k], g[k]), k
    # every kind of ray is represented and both hits and misses occur
    assert set(np.unique(g["kind"])) == {0, 1, 2, 3, 4}
    assert (g["id"] < 0).any() and (g["id"] >= 0).any() and len(g["id"]) >= 4096


def test_scatter_all_materials(oracle, rtiow_flat):
    g = load_golden("func_vectors.npz")
    seen = set()
    for cin, want in zip(g["scatter_in"], g["scatter_out"]):
        ray, p, n = cin[0:6], cin[6:9], cin[9:12]
        front, obj = int(cin[12]), int(cin[13])
        seed, pixel, sample, segment = (int(x) for x in cin[14:18])
        mat = int(rtiow_flat.prim_mat[obj])
        seen.add(int(rtiow_flat.mat_type[mat]))
        ok, att, so, sd, em = oracle.scatter(rtiow_flat, mat, ray, p, n, front, [0, 0], seed, pixel,
                                             sample, segment)
        assert ok == bool(want[0])
        assert np.array_equal(att, want[1:4])
        assert np.array_equal(so, want[4:7])
        assert np.array_equal(sd, want[7:10])
        assert not em.any()
    assert seen == {0, 1, 2}


def test_write_color(oracle):
    g = load_golden("func_vectors.npz")
    assert np.array_equal(oracle.write_color(g["wc_in"], int(g["wc_spp"])), g["wc_out"])


def test_ray_color_depth_limits(oracle, rtiow_flat):
    g = load_golden("func_vectors.npz")
    for ray, (depth, seed, pixel, sample), want, nrays in zip(g["rc_rays"], g["rc_in"], g["rc_out"],
                                                              g["rc_nrays"]):
        got, n = oracle.ray_color(rtiow_flat, ray, int(depth), int(seed), int(pixel), int(sample))
        assert np.array_equal(got, want)
        assert n == nrays
        if depth == 0:
            assert not got.any() and n == 0      # main.cc:13-14


@pytest.mark.parametrize("name", ["image_120x80_16spp.npz", "image_240x160_64spp.npz"])
def test_keyed_images_bit_exact(oracle, rtiow_flat, name):
    g = load_golden(name)
    c = g["cam"]
    cam = oracle.camera(c[0:3], c[3:6], c[6:9], c[9], c[10], c[11], c[12])
    W, H, spp = int(g["W"]), int(g["H"]), int(g["spp"])
    sums, rays = oracle.render(rtiow_flat, cam, W, H, spp, int(g["depth"]), int(g["seed"]), threads=8)
    assert rays == int(g["rays"])
    assert np.array_equal(sums, g["sums"])
    assert np.array_equal(oracle.write_color(sums.reshape(-1, 3), spp).reshape(H, W, 3), g["p3"])


def test_sample_window(oracle, rtiow_flat):
    g = load_golden("image_120x80_16spp.npz")
    h = load_golden("image_120x80_8spp_from8.npz")
    c = g["cam"]
    cam = oracle.camera(c[0:3], c[3:6], c[6:9], c[9], c[10], c[11], c[12])
    second, rays = oracle.render(rtiow_flat, cam, 120, 80, 8, 50, 1, sample_begin=8, threads=8)
    assert np.array_equal(second, h["sums"]) and rays == int(h["rays"])
    first, _ = oracle.render(rtiow_flat, cam, 120, 80, 8, 50, 1, sample_begin=0, threads=8)
    np.testing.assert_allclose(first + second, g["sums"], rtol=1e-13)


def test_float_build_tracks_double_build(oracle, oracle32, rtiow_flat):
    """Calibration of what single precision costs on the same random stream: the float build of
    the restatement against the golden double image, per-pixel RMSE of the mean radiance."""
    g = load_golden("image_120x80_16spp.npz")
    c = g["cam"]
    cam = oracle32.camera(c[0:3], c[3:6], c[6:9], c[9], c[10], c[11], c[12])
    sums, rays = oracle32.render(rtiow_flat, cam, 120, 80, 16, 50, 1, threads=8)
    rmse = np.sqrt(np.mean((sums / 16 - g["sums"] / 16) ** 2))
    # measured: rmse 0.0049 (an independent seed gives 0.042); the naive float quadratic of
    # cuda/sphere.h re-hits surfaces it starts on, so the float build traces ~10 % more rays
    assert rmse < 0.01
    assert abs(rays - int(g["rays"])) / int(g["rays"]) < 0.15

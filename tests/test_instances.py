"""Instance wrappers (cuda/sceneobject.h:31-109 rotate_y, with a translation) on the CPU: the
facade's composition of nested wrappers and the checker's treatment of them."""
import numpy as np

from raytrace_amd import scene as S


def _rot(angle):
    r = np.deg2rad(angle)
    c, s = np.cos(r), np.sin(r)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])          # cuda/sceneobject.h:98-99


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def test_nested_wrappers_compose_into_one_rotation_and_offset():
    m = S.lambertian((0.5, 0.5, 0.5))
    leaf = S.xy_rect(0, 1, 0, 1, 0, m)
    obj = S.translate(S.rotate_y(S.translate(S.rotate_y(leaf, 20), (1, 2, 3)), -65), (-4, 0.5, 9))
    flat = S.flatten(S.sceneobject_list([obj, S.sphere((0, 0, 0), 1, m)]))
    assert list(flat.prim_xform) == [0, -1] and flat.xform.shape == (1, 4)
    angle, off = flat.xform[0, 0], flat.xform[0, 1:]
    p = np.array([0.3, -0.2, 0.7])
    want = _rot(-65) @ (_rot(20) @ p + [1, 2, 3]) + [-4, 0.5, 9]
    np.testing.assert_allclose(_rot(angle) @ p + off, want, atol=1e-12)
    # the same transform on several leaves is stored once
    two = S.flatten(S.translate(S.box((0, 0, 0), (1, 1, 1), m), (5, 0, 0)))
    assert two.n_prims == 6 and two.xform.shape == (1, 4) and set(two.prim_xform) == {0}


def test_box_is_six_rects_in_the_reference_order():
    a, b = S.lambertian((0.1, 0.8, 0.1)), S.lambertian((0.8, 0.1, 0.1))
    flat = S.flatten(S.box((90, 248, 395), (110, 300, 415), a, b))            # cuda/main.cu:127-137
    assert list(flat.prim_type) == [S.RT_XY_RECT, S.RT_XY_RECT, S.RT_XZ_RECT, S.RT_XZ_RECT, S.RT_YZ_RECT, S.RT_YZ_RECT]
    np.testing.assert_array_equal(flat.prim_geom[:, :5], [
        [90, 110, 248, 300, 415], [90, 110, 248, 300, 395],
        [90, 110, 395, 415, 300], [90, 110, 395, 415, 248],
        [248, 300, 395, 415, 110], [248, 300, 395, 415, 90]])
    assert list(flat.prim_mat) == [0, 1, 0, 1, 0, 1]


def test_checker_sphere_instance_is_the_moved_sphere(oracle):
    m = S.lambertian((0.5, 0.5, 0.5))
    c, T, ang = np.array([3.0, 1.0, -2.0]), np.array([-1.0, 0.5, 4.0]), 40.0
    inst = S.flatten(S.translate(S.rotate_y(S.sphere(c, 1.5, m), ang), T))
    moved = S.flatten(S.sphere(_rot(ang) @ c + T, 1.5, m))
    rng = np.random.default_rng(2)
    centre = _rot(ang) @ c + T
    o = centre + 6 * _unit(rng.normal(size=(500, 3)))
    rays = np.concatenate([o, centre - o + 1.2 * rng.normal(size=(500, 3))], 1)
    a, b = oracle.trace(inst, rays), oracle.trace(moved, rays)
    assert (a["id"] == b["id"]).all() and (a["id"] == 0).sum() > 200
    hit = a["id"] == 0
    np.testing.assert_allclose(a["t"][hit], b["t"][hit], rtol=1e-12)
    np.testing.assert_allclose(a["p"][hit], b["p"][hit], atol=1e-11)
    np.testing.assert_allclose(a["n"][hit], b["n"][hit], atol=1e-12)
    assert (a["front"] == b["front"]).all()
    # texture coordinates follow the instance: they are those of the normal turned back
    n_obj = (a["n"][hit] * np.where(a["front"][hit, None] == 1, 1, -1)) @ _rot(ang)   # R^T n
    u = (np.arctan2(-n_obj[:, 2], n_obj[:, 0]) + np.pi) / (2 * np.pi)
    v = np.arccos(np.clip(-n_obj[:, 1], -1, 1)) / np.pi
    np.testing.assert_allclose(a["uv"][hit], np.stack([u, v], 1), atol=1e-9)


def test_moving_the_whole_world_moves_every_hit(oracle):
    """Wrapping the room and turning the rays the same way is a change of coordinates: same
    objects hit at the same distances, sides kept (a ray that starts inside the glass sphere
    still leaves through a back face)."""
    world = S.cornell_boxes_scene()
    ang, T = 33.0, np.array([120.0, -40.0, 77.0])
    plain = S.flatten(world, background=S.RT_BG_SOLID)
    moved = S.flatten(S.translate(S.rotate_y(world, ang), T), background=S.RT_BG_SOLID)
    assert moved.xform.shape[0] == 3          # room, and the two boxes' own transforms on top of it
    rng = np.random.default_rng(6)
    o = rng.uniform(5, 550, (4000, 3))
    glass = _rot(-18) @ np.array([82.5, 215, 82.5]) + [130, 0, 65]
    o[:300] = glass + 20 * rng.uniform(-1, 1, (300, 3))      # inside the r = 50 sphere
    d = _unit(rng.normal(size=(4000, 3))) * rng.uniform(0.3, 2, (4000, 1))
    a = oracle.trace(plain, np.concatenate([o, d], 1))
    R = _rot(ang)
    b = oracle.trace(moved, np.concatenate([o @ R.T + T, d @ R.T], 1))
    assert (a["id"] == b["id"]).mean() > 0.999               # an edge-on ray may flip between two faces
    ok = (a["id"] == b["id"]) & (a["id"] >= 0)
    np.testing.assert_allclose(a["t"][ok], b["t"][ok], rtol=1e-9)
    np.testing.assert_allclose(a["p"][ok] @ R.T + T, b["p"][ok], atol=1e-8)
    np.testing.assert_allclose(a["n"][ok] @ R.T, b["n"][ok], atol=1e-12)
    assert (a["front"][ok] == b["front"][ok]).all()
    np.testing.assert_allclose(a["uv"][ok], b["uv"][ok], atol=1e-9)
    sphere_id = plain.n_prims - 1
    inside = a["id"][:300] == sphere_id
    assert inside.sum() > 250 and (a["front"][:300][inside] == 0).all()
    assert set(range(6, 18)) <= set(np.unique(a["id"]))      # every face of both boxes is seen


def test_instances_render_on_the_checker(oracle):
    flat = S.flatten(S.cornell_boxes_scene(), background=S.RT_BG_SOLID)
    cam = oracle.camera(**S.cornell_camera(1.5).desc())
    sums, rays = oracle.render(flat, cam, 48, 32, 8, 20, 4, threads=8)
    assert rays > 48 * 32 * 8 * 2 and sums.max() > 0 and np.isfinite(sums).all()

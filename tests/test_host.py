"""Host-side logic and the C-ABI surface, without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from raytrace_amd import api
    lib = api.lib()
    header = open(os.path.join(ROOT, "include", "rt.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z_]+)\s*\(", header))
    declared -= {"rt_error"}
    assert declared == set(api.EXPORTS), declared ^ set(api.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert b"gfx950" in lib.rt_version()


def test_struct_sizes_match_header_layout():
    from raytrace_amd import api
    assert C.sizeof(api.Hit) == 44 and api.HIT_DTYPE.itemsize == 44
    # (the 56-byte layout without the tuning fields and the 64-byte one without stack_lds_entries are accepted too)
    assert C.sizeof(api.RenderParams) == 72 and api.RenderParams.traversal_slice.offset == 56 and api.RenderParams.stack_lds_entries.offset == 64
    assert api.RenderParams.traversal_exit_share.offset == 68          # (in what was padding behind stack_lds_entries)
    assert C.sizeof(api.CameraDesc) == 13 * 8


def test_no_gpu_is_an_error_not_a_fallback():
    import torch
    from raytrace_amd import api
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.RtError) as e:
        api.Context(0)
    assert e.value.code == api.RT_ERR_NO_DEVICE


def test_shard_rows_matches_python_row_map():
    from raytrace_amd import api, dist
    for H, tr, n in ((1080, 8, 8), (1080, 8, 3), (80, 8, 3), (77, 16, 4), (10, 3, 2)):
        total = 0
        for g in range(n):
            rows = dist.shard_row_map(H, tr, g, n)
            assert api.Context.shard_rows(H, (tr, g, n)) == len(rows)
            total += len(rows)
        assert total == H
        allrows = np.sort(np.concatenate([dist.shard_row_map(H, tr, g, n) for g in range(n)]))
        assert np.array_equal(allrows, np.arange(H))
    assert api.Context.shard_rows(1080) == 1080
    with pytest.raises(api.RtError):
        api.Context.shard_rows(100, (8, 3, 3))


def test_flatten_keeps_list_order_and_dedups_materials():
    from raytrace_amd import scene as S
    white = S.lambertian((.73, .73, .73))
    chk = S.checker_texture((0.2, 0.3, 0.1), (0.9, 0.9, 0.9))
    img = S.image_texture(4, 2, np.arange(24, dtype=np.uint8).reshape(2, 4, 3))
    w = S.sceneobject_list()
    w.add(S.sphere((0, 1, 2), 3, white))
    w.add(S.xy_rect(0, 1, 2, 3, 4, white))
    w.add(S.bvh_node(S.sceneobject_list([S.sphere((1, 1, 1), 1, S.lambertian(chk)),
                                         S.sphere((2, 2, 2), 1, S.lambertian(img))])))
    w.add(S.yz_rect(0, 1, 2, 3, 4, S.diffuse_light((7, 7, 7))))
    w.add(S.xz_rect(0, 1, 2, 3, 4, S.metal((1, 1, 1), 3.0)))
    f = S.flatten(w, background=S.RT_BG_SOLID)
    assert list(f.prim_type) == [S.RT_SPHERE, S.RT_XY_RECT, S.RT_SPHERE, S.RT_SPHERE, S.RT_YZ_RECT, S.RT_XZ_RECT]
    assert list(f.prim_mat) == [0, 0, 1, 2, 3, 4]
    assert list(f.mat_type) == [S.RT_LAMBERTIAN, S.RT_LAMBERTIAN, S.RT_LAMBERTIAN, S.RT_DIFFUSE_LIGHT, S.RT_METAL]
    assert f.mat_param[4, 3] == 1.0                       # fuzz clamp, material.h:40
    assert list(f.tex_type) == [S.RT_TEX_SOLID, S.RT_TEX_SOLID, S.RT_TEX_CHECKER, S.RT_TEX_IMAGE]
    assert (f.tex_a[2], f.tex_b[2]) == (0, 1) and (f.tex_a[3], f.tex_b[3]) == (4, 2)
    assert f.texels.size == 24 and list(f.mat_tex) == [-1, 2, 3, -1, -1]
    assert np.array_equal(f.prim_geom[1, :5], [0, 1, 2, 3, 4])


def test_camera_path_matches_float_restatement(oracle):
    """update_scene (cuda/main.cu:220-279) in the Python host mirror against the C restatement;
    the CUDA original cannot be compiled here, so this pair is parity-unpinned beyond review."""
    from raytrace_amd import scene as S
    cam = S.camera((450, 278, 200), (0, 278, 555), (0, 1, 0), 40.0, 16 / 9, 0.0, 10.0)
    state = np.array([450, 278, 200, 0, 0, 0], np.float32)
    for frame in range(0, 600, 1):
        S.update_scene(cam, frame)
        state = oracle.camera_path_step(frame, state)
        if frame % 50 == 49 or frame < 3:
            np.testing.assert_allclose(cam.look_from, state[:3], rtol=3e-6)
            np.testing.assert_allclose(cam.angles, state[3:], rtol=0, atol=1e-6)
    # 199 yaw steps out (frames 1..199), 200 back (200..399); 100 tilt steps each way
    np.testing.assert_allclose(cam.angles, (0, -0.125, 0), atol=1e-4)


def test_p3_writer_layout():
    from raytrace_amd import ppm
    img = np.zeros((2, 3, 3), np.uint8)
    img[0, 0] = (1, 2, 3)          # bottom-left
    img[1, 2] = (7, 8, 9)          # top-right
    text = ppm.p3_bytes(img).decode().split("\n")
    assert text[:3] == ["P3", "3 2", "255"]
    assert text[3 + 2] == "7 8 9" and text[3 + 3] == "1 2 3"     # top row first (main.cc:111)
    assert ppm.frame_name(12) == "image_12_low.ppm"


def test_cpp_facade_flattens_instances_like_the_python_facade(tmp_path):
    """include/rt_scene.hpp and raytrace_amd/scene.py are two front ends to one descriptor."""
    import subprocess
    from raytrace_amd import scene as S
    exe = tmp_path / "flat_dump"
    subprocess.run(["g++", "-std=c++17", "-O0", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "flat_dump.cpp"), "-o", str(exe)], check=True)
    lines = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines()
    world = S.cornell_boxes_scene()
    for w in (world, S.translate(S.rotate_y(world, 33), (120, -40, 77))):
        flat = S.flatten(w, background=S.RT_BG_SOLID)
        head = lines.pop(0).split()
        assert head == ["scene", str(flat.n_prims), str(flat.xform.shape[0])]
        for i in range(flat.n_prims):
            v = lines.pop(0).split()
            assert int(v[0]) == flat.prim_type[i] and int(v[6]) == flat.prim_mat[i] and int(v[7]) == flat.prim_xform[i]
            np.testing.assert_array_equal([float(x) for x in v[1:6]], flat.prim_geom[i, :5])
        for i in range(flat.xform.shape[0]):
            np.testing.assert_allclose([float(x) for x in lines.pop(0).split()], flat.xform[i], rtol=0, atol=1e-12)
    assert not lines


def test_cpp_facade_walks_the_camera_path_like_the_python_facade(tmp_path):
    """cuda/camera.h:29-117 and cuda/main.cu:206-279 in include/rt_scene.hpp against
    raytrace_amd/scene.py: the same 600 camera states, to the last bit of their floats."""
    import subprocess
    from raytrace_amd import scene as S
    exe = tmp_path / "camera_path_dump"
    subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "camera_path_dump.cpp"), "-o", str(exe)], check=True)
    lines = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines()
    assert len(lines) == 601
    cam = S.camera((450, 278, 200), (0, 278, 555), (0, 1, 0), 40.0, 16 / 9, 0.0, 10.0, angles=(0, 0, 0))
    moved = 0
    for frame in range(600):
        S.update_scene(cam, frame)
        v = lines[frame].split()
        assert int(v[0]) == frame
        got = np.array([float(x) for x in v[1:]]).astype(np.float32)          # nine digits identify a float
        want = np.array([*cam.compute_camera_scene()["lookfrom"], *cam.angles])
        assert np.array_equal(want, want.astype(np.float32))                  # the path is walked in single precision
        assert np.array_equal(got, want.astype(np.float32)), (frame, got, want)
        moved += frame > 0 and lines[frame].split()[1:4] != lines[frame - 1].split()[1:4]
    assert moved == 599
    c2 = S.rtiow_camera(1.5)
    c2.rotate_camera_z(33.0)
    c2.rotate_camera_x(-12.5)
    c2.translate_camera(0.1, 0.2, 0.3)
    c2.rotate_camera_y(400.0)
    got = np.array([float(x) for x in lines[600].split()[1:]]).astype(np.float32)
    assert np.array_equal(got, np.array([*c2.look_from, *c2.angles]).astype(np.float32)), (got, c2.look_from, c2.angles)


def test_ppm_files_of_the_library_match_the_text_layout(tmp_path):
    """rt_write_ppm (output_image, cuda/main.cu:358-383): the P3 file is p3_bytes' text (every digit
    count, rows top to bottom), the P6 file the same pixels in binary; bad arguments are errors."""
    from raytrace_amd import api, ppm
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
    img[0, 0] = (0, 9, 10)
    img[6, 4] = (99, 100, 255)
    ppm.write_p3(str(tmp_path / "a.ppm"), img)
    got = open(tmp_path / "a.ppm", "rb").read()
    want = ("P3\n5 7\n255\n" + "".join(f"{r} {g} {b}\n" for r, g, b in img[::-1].reshape(-1, 3).tolist())).encode()
    assert got == want == ppm.p3_bytes(img)
    ppm.write_p6(str(tmp_path / "b.ppm"), img)
    assert open(tmp_path / "b.ppm", "rb").read() == b"P6\n5 7\n255\n" + img[::-1].tobytes()
    with pytest.raises(api.RtError):
        api.write_ppm(str(tmp_path / "no_such_dir" / "c.ppm"), img)
    with pytest.raises(ValueError):
        api.write_ppm(str(tmp_path / "d.ppm"), img[..., :2])

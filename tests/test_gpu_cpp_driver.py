"""The C++ host path: examples/rtiow_main.cpp (the reference's main.cc:87-129 rewritten over the
facade of include/rt_scene.hpp) must write the same P3 image as the Python host path."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "rtiow")


def test_cpp_driver_matches_python_host(tmp_path, oracle):
    if not os.path.exists(EXE):
        pytest.fail("examples/rtiow is not built: run __graft_entry__.build()")
    from raytrace_amd import api, scene as S, ppm
    rec = S.rtiow_records()
    scene_txt = tmp_path / "scene.txt"
    with open(scene_txt, "w") as f:
        for o in rec:
            f.write(" ".join(repr(float(x)) for x in o) + "\n")
    W, H, spp = 120, 80, 16
    out = subprocess.run([EXE, str(scene_txt), str(W), str(H), str(spp)], capture_output=True, check=True, timeout=120)
    tokens = out.stdout.split()
    assert tokens[:4] == [b"P3", str(W).encode(), str(H).encode(), b"255"]
    got = np.array(tokens[4:], dtype=np.int32).reshape(H, W, 3)[::-1]        # back to bottom-up rows

    ctx = api.Context(0)
    ctx.upload_scene(S.flatten(S.world_from_records(rec)))
    ctx.set_camera((6, 3, -13), (0, 0, 0), (0, 1, 0), 20, float(np.float32(W / H)), 0.1, 10.0)
    img = ctx.render_host(api.Context.params(W, H, spp, 50, 1))
    want = oracle.write_color(img[..., :3].astype(np.float64).reshape(-1, 3), spp).reshape(H, W, 3)
    assert np.array_equal(got, want)
    # and the text the Python writer produces is byte-identical to the C++ driver's
    assert ppm.p3_bytes(want.astype(np.uint8)) == out.stdout
    # frame 450 of the camera path, walked by the C++ facade's update_scene and by the Python one
    out = subprocess.run([EXE, str(scene_txt), str(W), str(H), str(spp), "450"], capture_output=True, check=True, timeout=120)
    cam = S.rtiow_camera(float(np.float32(W / H)))
    for f in range(451):
        S.update_scene(cam, f)
    ctx.set_camera(**cam.desc())
    img = ctx.render_host(api.Context.params(W, H, spp, 50, 1))
    want = oracle.write_color(img[..., :3].astype(np.float64).reshape(-1, 3), spp).reshape(H, W, 3)
    assert ppm.p3_bytes(want.astype(np.uint8)) == out.stdout
    ctx.close()


def test_cpp_sequence_driver_writes_the_python_drivers_files(tmp_path):
    """examples/sequence_main.cpp (the frame loop of cuda/main.cu:557-591 in C++: update_scene per
    frame, two contexts in flight, rt_quantize, image_<frame>_low.ppm) against
    raytrace_amd.sequence.render_sequence: the same files, byte for byte, in P3 and in P6."""
    exe = os.path.join(ROOT, "examples", "sequence")
    if not os.path.exists(exe):
        pytest.fail("examples/sequence is not built: run __graft_entry__.build()")
    from raytrace_amd import api, scene as S, sequence
    rec = S.rtiow_records()
    scene_txt = tmp_path / "scene.txt"
    with open(scene_txt, "w") as f:
        for o in rec:
            f.write(" ".join(repr(float(x)) for x in o) + "\n")
    W, H, spp, first, n = 96, 54, 4, 148, 5                 # frames either side of the path's first change of motion at 150
    ctx = api.Context(0)
    ctx.upload_scene(S.flatten(S.world_from_records(rec)))
    for fmt in ("p3", "p6"):
        cpp_dir, py_dir = tmp_path / f"cpp_{fmt}", tmp_path / f"py_{fmt}"
        subprocess.run([exe, str(scene_txt), str(cpp_dir), str(W), str(H), str(spp), str(first), str(n), fmt],
                       capture_output=True, check=True, timeout=120)
        cam = S.rtiow_camera(float(np.float32(W / H)))
        done = sequence.render_sequence(ctx, cam, n, W, H, spp, str(py_dir), max_depth=25, seed=1, fmt=fmt, first_frame=first)
        assert [d[0] for d in done] == list(range(first, first + n))
        for frame in range(first, first + n):
            name = os.path.join("ppm_images", f"image_{frame}_low.ppm")
            got, want = open(cpp_dir / name, "rb").read(), open(py_dir / name, "rb").read()
            assert got == want, (fmt, frame)
        assert sorted(os.listdir(cpp_dir / "ppm_images")) == sorted(os.listdir(py_dir / "ppm_images"))
    # with the reconstruction stage between render and write_color: the bytes the Python binding's calls give
    import torch
    cpp_dir = tmp_path / "cpp_recon"
    subprocess.run([exe, str(scene_txt), str(cpp_dir), str(W), str(H), str(spp), str(first), "2", "p6", "recon"],
                   capture_output=True, check=True, timeout=120)
    cam = S.rtiow_camera(float(np.float32(W / H)))
    fb = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    guides = torch.zeros((2, H, W, 4), dtype=torch.float32, device="cuda")
    rgb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for frame in range(first + 2):
        S.update_scene(cam, frame)
        if frame < first:
            continue
        ctx.set_camera(**cam.desc())
        ctx.render(api.Context.params(W, H, spp, 25, 1 + frame), fb.data_ptr())
        ctx.render_guides(W, H, guides.data_ptr())
        ctx.reconstruct(fb.data_ptr(), guides.data_ptr(), W, H, fb.data_ptr())
        ctx.quantize(fb.data_ptr(), W * H, spp, rgb.data_ptr())
        ctx.stats()
        got = open(cpp_dir / "ppm_images" / f"image_{frame}_recon.ppm", "rb").read()
        img = rgb.cpu().numpy()
        assert got == f"P6\n{W} {H}\n255\n".encode("ascii") + np.ascontiguousarray(img[::-1]).tobytes(), frame
    ctx.close()

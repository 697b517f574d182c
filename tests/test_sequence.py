"""Frame writer and sequence driver (the reference's per-frame loop and helper scripts)."""
import io
import os

import numpy as np
import pytest


def test_png_encoder_round_trips():
    from PIL import Image
    from raytrace_amd import sequence
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (13, 21, 3), dtype=np.uint8)          # bottom-up rows
    back = np.asarray(Image.open(io.BytesIO(sequence.png_bytes(img))).convert("RGB"))
    assert np.array_equal(back, img[::-1])


def test_frame_paths_follow_the_reference_layout():
    from raytrace_amd import sequence
    assert sequence.frame_path("out", 7, "p3").endswith(os.path.join("ppm_images", "image_7_low.ppm"))
    assert sequence.frame_path("out", 7, "png", "high").endswith(os.path.join("png_images", "high", "image_7.png"))
    cmd, ran = sequence.encode_video("out")
    assert cmd[:5] == ["ffmpeg", "-r", "60", "-i", os.path.join("out", "png_images", "low", "image_%01d.png")]


@pytest.mark.gpu
def test_sequence_frames_on_gpu(tmp_path, oracle):
    from PIL import Image
    from raytrace_amd import api, scene as S, sequence
    ctx = api.Context(0)
    ctx.upload_scene(S.flatten(S.rtiow_scene()))
    W, H, spp = 96, 54, 4
    cam = S.rtiow_camera(W / H)
    done = sequence.render_sequence(ctx, cam, 3, W, H, spp, str(tmp_path), fmt="p3")
    assert [d[0] for d in done] == [0, 1, 2] and all(d[1] > W * H * spp for d in done)
    # frame 2 again through PNG, by a "second rank" that owns the even frames
    cam2 = S.rtiow_camera(W / H)
    done2 = sequence.render_sequence(ctx, cam2, 3, W, H, spp, str(tmp_path), fmt="png", rank=0, world=2)
    assert [d[0] for d in done2] == [0, 2]
    text = open(sequence.frame_path(str(tmp_path), 2, "p3")).read().split()
    assert text[:4] == ["P3", str(W), str(H), "255"]
    p3 = np.array(text[4:], dtype=np.uint8).reshape(H, W, 3)
    png = np.asarray(Image.open(sequence.frame_path(str(tmp_path), 2, "png")).convert("RGB"))
    assert np.array_equal(p3, png)
    # the camera moved between frames (0.125 degrees of yaw per frame) and frames differ
    f0 = np.array(open(sequence.frame_path(str(tmp_path), 0, "p3")).read().split()[4:], dtype=np.uint8)
    assert not np.array_equal(f0, p3.reshape(-1))
    # frame 2 equals an independent render with the camera advanced by hand, quantised by write_color
    cam3 = S.rtiow_camera(W / H)
    for f in range(3):
        S.update_scene(cam3, f)
    ctx.set_camera(**cam3.desc())
    img = ctx.render_host(api.Context.params(W, H, spp, 25, seed=1 + 2))
    want = oracle.write_color(img[..., :3].astype(np.float64).reshape(-1, 3), spp).reshape(H, W, 3)[::-1]
    assert np.array_equal(p3, want.astype(np.uint8))
    # two contexts on two streams, a frame launched before the one before it is written: same files
    import torch
    other = api.Context(0)
    stream = torch.cuda.Stream()
    other.set_stream(stream.cuda_stream)
    other.upload_scene(S.flatten(S.rtiow_scene()))
    cam4 = S.rtiow_camera(W / H)
    out2 = tmp_path / "overlapped"
    done3 = sequence.render_sequence(ctx, cam4, 3, W, H, spp, str(out2), fmt="p3", overlap_ctx=other)
    assert [(d[0], d[1]) for d in done3] == [(d[0], d[1]) for d in done]
    for f in range(3):
        assert open(sequence.frame_path(str(out2), f, "p3")).read() == open(sequence.frame_path(str(tmp_path), f, "p3")).read()
    other.close()
    ctx.close()


@pytest.mark.gpu
def test_low_high_pairs_are_the_files_of_two_sequences(tmp_path):
    """The low / high sample-count pair of every frame (cuda/main.cu:557-583, 1 and 20 samples there),
    rendered as two windows of the high frame's samples: the low file is the one a sequence at low_spp
    writes, the high file the one a sequence at high_spp writes, and the rays add up to the high
    sequence's."""
    from raytrace_amd import api, scene as S, sequence
    ctx = api.Context(0)
    ctx.upload_scene(S.flatten(S.rtiow_scene()))
    W, H, low, high = 96, 54, 1, 6
    pairs = sequence.render_pairs(ctx, S.rtiow_camera(W / H), 2, W, H, low, high, str(tmp_path / "pairs"), first_frame=1)
    lo = sequence.render_sequence(ctx, S.rtiow_camera(W / H), 2, W, H, low, str(tmp_path / "lo"), quality="low", first_frame=1)
    hi = sequence.render_sequence(ctx, S.rtiow_camera(W / H), 2, W, H, high, str(tmp_path / "hi"), quality="high", first_frame=1)
    assert [d[0] for d in pairs] == [1, 2]
    assert [d[1] for d in pairs] == [d[1] for d in hi]
    for f in (1, 2):
        for quality, single in (("low", "lo"), ("high", "hi")):
            got = open(sequence.frame_path(str(tmp_path / "pairs"), f, "p3", quality)).read()
            assert got == open(sequence.frame_path(str(tmp_path / single), f, "p3", quality)).read(), (f, quality)
    assert open(sequence.frame_path(str(tmp_path / "pairs"), 1, "p3", "low")).read() != \
        open(sequence.frame_path(str(tmp_path / "pairs"), 1, "p3", "high")).read()
    with pytest.raises(ValueError):
        sequence.render_pairs(ctx, S.rtiow_camera(W / H), 1, W, H, 4, 4, None)
    # with the reconstruction stage: the same low and high files, a third file per frame between them in quality
    rec = sequence.render_pairs(ctx, S.rtiow_camera(W / H), 2, 192, 108, 2, 128, str(tmp_path / "rec"), first_frame=1, fmt="p6", reconstruct=True)
    plain = sequence.render_pairs(ctx, S.rtiow_camera(W / H), 2, 192, 108, 2, 128, str(tmp_path / "plain"), first_frame=1, fmt="p6")
    for (f, rays, _, psnr_low, psnr_recon), (f2, rays2, _) in zip(rec, plain):
        assert (f, rays) == (f2, rays2)
        assert psnr_recon > psnr_low + 4.0, (psnr_low, psnr_recon)
        for quality in ("low", "high"):
            assert open(sequence.frame_path(str(tmp_path / "rec"), f, "p6", quality), "rb").read() == \
                open(sequence.frame_path(str(tmp_path / "plain"), f, "p6", quality), "rb").read()
        assert os.path.getsize(sequence.frame_path(str(tmp_path / "rec"), f, "p6", "recon")) == \
            os.path.getsize(sequence.frame_path(str(tmp_path / "rec"), f, "p6", "low"))
    ctx.close()

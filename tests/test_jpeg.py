"""Texture decoding (rt_decode_jpeg, the stand-in for load_texture's stbi_load, cuda/main.cu:393)
against bytes produced by the reference's own vendored decoder.

tests/golden/jpeg_cases.npz holds small JPEG files with the RGB8 that cuda/stb_image.h decodes
them to (oracle/gen_golden_jpeg.py); tests/golden/reference_textures.json holds digests of the
reference's texture files decoded the same way -- those files stay in the reference checkout, so
that check runs only where the checkout is present."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLD, load_golden

REF_TEXTURES = "/root/reference/cuda"


@pytest.fixture(scope="module")
def api():
    from raytrace_amd import api as A
    return A


def test_every_coding_mode_decodes_to_the_reference_bytes(api):
    z = load_golden("jpeg_cases.npz")
    names = sorted({k.split("/")[0] for k in z.files})
    assert len(names) >= 45
    kinds = set()
    for name in names:
        got = api.decode_jpeg(z[name + "/jpeg"].tobytes())
        want = z[name + "/rgb"]
        assert got.shape == want.shape, name
        assert np.array_equal(got, want), (name, int(np.abs(got.astype(int) - want).max()))
        kinds.add(name.split("_")[0])
    assert {"base444", "base420", "base422", "prog420", "prog444", "grey", "restart", "cmyk", "patched"} <= kinds


def test_file_entry_point_and_facade(api, tmp_path):
    from raytrace_amd import scene as S
    z = load_golden("jpeg_cases.npz")
    path = tmp_path / "t.jpg"
    path.write_bytes(z["base420_48x40/jpeg"].tobytes())
    tex = S.load_texture(str(path))
    assert (tex.texture_width, tex.texture_height) == (48, 40)
    assert np.array_equal(tex.texels, z["base420_48x40/rgb"])
    flat = S.flatten(S.sphere((0, 0, 0), 1, S.lambertian(tex)))
    assert flat.tex_type[0] == S.RT_TEX_IMAGE and flat.texels.size == 48 * 40 * 3


def test_bad_input_is_an_error(api, tmp_path):
    z = load_golden("jpeg_cases.npz")
    good = z["prog420_48x40/jpeg"].tobytes()
    for bad in (b"", b"\x89PNG\r\n\x1a\n" + bytes(64), good[:100], good[:2] + b"\xff\xc9" + good[4:]):
        with pytest.raises(api.RtError) as e:
            api.decode_jpeg(bad)
        assert e.value.code in (api.RT_ERR_INVALID, api.RT_ERR_UNSUPPORTED)
    with pytest.raises(api.RtError):
        api.load_jpeg_file(str(tmp_path / "missing.jpg"))
    # entropy data cut short: what has been decoded is kept (the rest of the picture is flat),
    # as in the reference decoder
    w = api.decode_jpeg(good[: len(good) // 2] + b"\xff\xd9")
    assert w.shape == (40, 48, 3)


@pytest.mark.skipif(not os.path.isdir(REF_TEXTURES), reason="the reference checkout is not on this machine")
def test_reference_texture_files(api):
    want = json.load(open(os.path.join(GOLD, "reference_textures.json")))
    assert len(want) == 11
    for rel, rec in want.items():
        got = api.load_jpeg_file(os.path.join(REF_TEXTURES, rel))
        assert got.shape == (rec["height"], rec["width"], 3), rel
        assert hashlib.sha256(got.tobytes()).hexdigest() == rec["sha256_rgb8"], rel

"""The host library's own JPEG reader (raytracing_amd/host/jpeg_decode.cpp) against libjpeg-turbo (through PIL) and against
pinned digests of the shipped textures.  The reader follows the IJG decoder's arithmetic (islow IDCT, fancy upsampling,
fixed-point colour conversion), so the comparison is exact, not within a tolerance."""
import hashlib
import io
import os

import numpy as np
import pytest

import raytracing_amd as ra

ASSETS = ra.default_assets_dir()
# sha256 of the decoded 8-bit RGB pixels; the same bytes libjpeg-turbo 3 (PIL 12) gives for these files
SHIPPED = {
    "earthmap.jpg": ((512, 1024, 3), "a8cdc92a168d554ddc693785d31f5e251063724f44099571d7fbce3b43d44c45"),
    "DamagedHelmet/glTF/Default_albedo.jpg": ((2048, 2048, 3), "3f314b261ac2327044bfc879019a30b01a49b85834ca9058414ed427b7ba154f"),
    "DamagedHelmet/glTF/Default_metalRoughness.jpg": ((2048, 2048, 3), "e7a49431ecbf7f2c8cfe9ead693f4e958c1f810a02077d5f8f46127d8c8d389e"),
}


@pytest.mark.parametrize("name", sorted(SHIPPED))
def test_shipped_textures_pinned(name):
    img = ra.decode_image(open(os.path.join(ASSETS, name), "rb").read())
    shape, digest = SHIPPED[name]
    assert img.shape == shape
    assert hashlib.sha256(img.tobytes()).hexdigest() == digest


def _synthetic(w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 120 * np.sin(xx / 7.0 + seed), 127 + 120 * np.cos(yy / 5.0), (xx * 3 + yy * 5) % 256], axis=2)
    noise = rng.integers(-30, 30, size=(h, w, 3))
    return np.clip(base + noise, 0, 255).astype(np.uint8)


CASES = [
    # (w, h, PIL save options)
    (64, 48, dict(quality=90, subsampling=0)),                       # 4:4:4 baseline
    (67, 45, dict(quality=75, subsampling=2)),                       # 4:2:0, sizes that are no multiple of the MCU
    (130, 31, dict(quality=60, subsampling=1)),                      # 4:2:2 (h2v1)
    (97, 83, dict(quality=85, subsampling=2, progressive=True)),     # progressive with subsampled chroma
    (40, 40, dict(quality=95, subsampling=0, progressive=True)),
    (3, 2, dict(quality=90, subsampling=2)),                         # chroma planes of two samples or fewer
    (256, 64, dict(quality=30, subsampling=2, optimize=True)),       # optimised Huffman tables
    (33, 17, dict(quality=100, subsampling=0)),
]


@pytest.mark.parametrize("w,h,opts", CASES)
def test_against_libjpeg(w, h, opts):
    Image = pytest.importorskip("PIL.Image")
    buf = io.BytesIO()
    Image.fromarray(_synthetic(w, h, w * 31 + h)).save(buf, format="JPEG", **opts)
    data = buf.getvalue()
    want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    got = ra.decode_image(data)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_greyscale_and_restart_markers():
    Image = pytest.importorskip("PIL.Image")
    g = _synthetic(75, 50, 3)[:, :, 0]
    buf = io.BytesIO(); Image.fromarray(g, mode="L").save(buf, format="JPEG", quality=80)
    got = ra.decode_image(buf.getvalue())
    want = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))
    assert np.array_equal(got, want)
    # restart intervals (DRI + RSTn markers between MCU rows)
    buf = io.BytesIO()
    try:
        Image.fromarray(_synthetic(90, 70, 4)).save(buf, format="JPEG", quality=80, subsampling=2, restart_marker_blocks=3)
    except TypeError:
        pytest.skip("this PIL cannot write restart markers")
    data = buf.getvalue()
    got = ra.decode_image(data)
    want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    assert np.array_equal(got, want)


def test_rejects_garbage():
    with pytest.raises(ValueError):
        ra.decode_image(b"\xff\xd8\xff\xe0 not a jpeg at all")
    with pytest.raises(ValueError):
        ra.decode_image(b"")
    data = open(os.path.join(ASSETS, "earthmap.jpg"), "rb").read()
    # a truncated stream decodes what is there or is refused; it must not crash
    try:
        ra.decode_image(data[: len(data) // 3])
    except ValueError:
        pass


def test_helmet_scene_loads_its_jpeg_twins():
    s = ra.HostScene("helmet")
    d = s.desc
    sizes = sorted((d.textures[i].width, d.textures[i].height) for i in range(d.n_textures) if d.textures[i].kind == 1)
    assert sizes == [(2048, 2048), (2048, 2048)]
    s.close()

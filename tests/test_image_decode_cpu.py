"""Host decoder (rt_decode_image; replaces ImageHelper::new_from_raw_img_flow,
/root/reference/retto-core/src/image_helper.rs:34-44) -- no GPU needed.

PNG / PNM / BMP are lossless: results must equal the source pixels converted by the `image` crate's
to_rgb8 rules (alpha dropped, grey replicated, (v + 128) / 257 for 16-bit samples).  Files come from
two independent writers: Pillow (an independent codec implementation) and the small PNG writer below, which
produces the cases Pillow cannot (Adam7, chosen filter types, 2/4-bit grey, 16-bit colour, split IDAT).
JPEG: compared with Pillow's libjpeg decode of the same file -- equal, because both follow the IJG
integer arithmetic (slow integer IDCT, triangle chroma upsampling, 16-bit fixed-point colour conversion);
sequential and progressive files.
"""
import io
import struct
import zlib

import numpy as np
import pytest

import retto_amd

PIL = pytest.importorskip("PIL.Image")
from PIL import Image  # noqa: E402


def _rng_img(h, w, c, seed, dtype=np.uint8):
    rng = np.random.default_rng(seed)
    # smooth + noise so that filters / DCT see realistic content
    yy, xx = np.mgrid[0:h, 0:w]
    base = (np.sin(xx / 7.0 + seed) + np.cos(yy / 5.0)) * 60 + 128
    img = base[..., None] + rng.normal(0, 25, (h, w, c))
    if dtype == np.uint16:
        return np.clip(img * 257, 0, 65535).astype(np.uint16)
    return np.clip(img, 0, 255).astype(np.uint8)


def _save(img, fmt, **kw):
    b = io.BytesIO()
    img.save(b, fmt, **kw)
    return b.getvalue()


# ---- test-side PNG writer -------------------------------------------------------------------------
def _chunk(t, body):
    return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)


def _filter_row(ft, row, prev, bpp):
    row = np.frombuffer(row, np.uint8).astype(np.int32)
    prev = np.frombuffer(prev, np.uint8).astype(np.int32)
    a = np.concatenate([np.zeros(bpp, np.int32), row[:-bpp]]) if len(row) > bpp else np.zeros_like(row)
    c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if len(row) > bpp else np.zeros_like(row)
    if ft == 0:
        out = row
    elif ft == 1:
        out = row - a
    elif ft == 2:
        out = row - prev
    elif ft == 3:
        out = row - ((a + prev) >> 1)
    else:
        p = a + prev - c
        pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
        pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        out = row - pred
    return bytes([ft]) + (out & 255).astype(np.uint8).tobytes()


def _pack_rows(samples, depth):
    """samples: [h, w*chans] ints -> list of packed row bytes"""
    rows = []
    for r in samples:
        if depth == 16:
            rows.append(r.astype(">u2").tobytes())
        elif depth == 8:
            rows.append(r.astype(np.uint8).tobytes())
        else:
            bits = np.zeros((len(r) * depth + 7) // 8 * 8, np.uint8)
            for b in range(depth):
                bits[b::depth][: len(r)] = (r >> (depth - 1 - b)) & 1
            rows.append(np.packbits(bits).tobytes())
    return rows


def write_png(samples, ctype, depth, interlace=False, filters=(0, 1, 2, 3, 4), plte=None, split=1, extra=()):
    """samples: [h, w, chans] ints (chans matching ctype).  Returns the file bytes."""
    h, w, ch = samples.shape
    bpp = max(1, ch * depth // 8)
    raw = b""
    passes = [(0, 0, 1, 1)] if not interlace else [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    k = 0
    for x0, y0, dx, dy in passes:
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        rows = _pack_rows(sub.reshape(sub.shape[0], -1), depth)
        prev = bytes(len(rows[0]))
        for r in rows:
            raw += _filter_row(filters[k % len(filters)], r, prev, bpp)
            prev = r
            k += 1
    z = zlib.compress(raw, 6)
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, int(interlace)))
    for t, body in extra:
        out += _chunk(t, body)
    if plte is not None:
        out += _chunk(b"PLTE", np.asarray(plte, np.uint8).tobytes())
    step = (len(z) + split - 1) // split
    for i in range(0, len(z), step):
        out += _chunk(b"IDAT", z[i:i + step])
    return out + _chunk(b"IEND", b"")


def _to8(v16):
    return ((v16.astype(np.uint32) + 128) // 257).astype(np.uint8)


# ---- PNG ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode,c", [("RGB", 3), ("RGBA", 4), ("L", 1), ("LA", 2)])
@pytest.mark.parametrize("size", [(37, 53), (1, 1), (64, 200)])
def test_png_from_pillow(mode, c, size):
    h, w = size
    a = _rng_img(h, w, c, seed=h * 7 + c)
    img = Image.fromarray(a if c > 1 else a[..., 0], mode)
    got = retto_amd.decode_image(_save(img, "PNG"))
    want = a[..., :3] if c >= 3 else np.repeat(a[..., :1], 3, axis=2)
    assert got.shape == (h, w, 3) and np.array_equal(got, want)


def test_png_palette_and_bilevel_from_pillow():
    a = _rng_img(40, 61, 3, seed=3)
    p = Image.fromarray(a, "RGB").quantize(17)
    assert np.array_equal(retto_amd.decode_image(_save(p, "PNG")), np.asarray(p.convert("RGB")))
    one = Image.fromarray(a[..., 0], "L").point(lambda v: 255 if v > 128 else 0).convert("1")
    assert np.array_equal(retto_amd.decode_image(_save(one, "PNG")), np.asarray(one.convert("RGB")))
    # optimised palette files use 1 / 2 / 4-bit indices
    p4 = Image.fromarray(a, "RGB").quantize(5)
    assert np.array_equal(retto_amd.decode_image(_save(p4, "PNG", optimize=True)), np.asarray(p4.convert("RGB")))


@pytest.mark.parametrize("interlace", [False, True])
@pytest.mark.parametrize("ctype,depth,ch", [(0, 1, 1), (0, 2, 1), (0, 4, 1), (0, 8, 1), (0, 16, 1), (2, 8, 3), (2, 16, 3),
                                            (3, 1, 1), (3, 2, 1), (3, 4, 1), (3, 8, 1), (4, 8, 2), (4, 16, 2), (6, 8, 4), (6, 16, 4)])
def test_png_every_colour_type_depth_filter_and_adam7(ctype, depth, ch, interlace):
    rng = np.random.default_rng(ctype * 100 + depth)
    for h, w in ((13, 19), (1, 5), (5, 1), (9, 8), (3, 3)):
        s = rng.integers(0, 1 << depth, (h, w, ch), dtype=np.int64)
        plte = rng.integers(0, 256, (1 << min(depth, 8), 3), dtype=np.uint8) if ctype == 3 else None
        data = write_png(s, ctype, depth, interlace=interlace, plte=plte, split=3,
                         extra=[(b"gAMA", struct.pack(">I", 45455)), (b"tEXt", b"k\0v")])
        got = retto_amd.decode_image(data)
        if ctype == 3:
            want = plte[s[..., 0]]
        else:
            v = s if ctype in (2, 6) else np.repeat(s[..., :1], 3, axis=2)
            v = v[..., :3]
            want = _to8(v) if depth == 16 else (v * (255 // ((1 << depth) - 1))).astype(np.uint8) if depth < 8 else v.astype(np.uint8)
        assert got.shape == (h, w, 3)
        assert np.array_equal(got, want), (ctype, depth, interlace, h, w)


def test_png_16_bit_rounding_rule():
    # image 0.25.6 maps u16 -> u8 as (v + 128) / 257: 0x0080 -> 0, 0x0081 -> 1, 0xff7f -> 254, 0xff80 -> 255
    vals = np.array([0, 0x80, 0x81, 0x100, 0x7fff, 0x8000, 0xff7e, 0xff7f, 0xffff], np.int64).reshape(1, -1, 1)
    got = retto_amd.decode_image(write_png(vals, 0, 16))[0, :, 0]
    assert got.tolist() == [0, 0, 1, 1, 127, 128, 254, 255, 255]


def test_png_errors():
    good = write_png(np.zeros((4, 4, 3), np.int64), 2, 8)
    with pytest.raises(retto_amd.ImageError):
        retto_amd.decode_image(good[:-20])                      # truncated
    bad_crc = bytearray(good); bad_crc[30] ^= 1
    with pytest.raises(retto_amd.ImageError, match="CRC"):
        retto_amd.decode_image(bytes(bad_crc))
    with pytest.raises(retto_amd.ImageError):
        retto_amd.decode_image(b"not an image at all")
    with pytest.raises(retto_amd.ImageError):
        retto_amd.decode_image(b"")
    # image data shorter than the header promises
    short = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 8, 8, 8, 2, 0, 0, 0)) + \
        _chunk(b"IDAT", zlib.compress(bytes(50))) + _chunk(b"IEND", b"")
    with pytest.raises(retto_amd.ImageError, match="ends early"):
        retto_amd.decode_image(short)
    with pytest.raises(retto_amd.ImageError):                   # depth 4 is not valid for RGB
        retto_amd.decode_image(b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 8, 8, 4, 2, 0, 0, 0)) + _chunk(b"IEND", b""))


# ---- PNM / BMP ------------------------------------------------------------------------------------
def test_pnm_and_bmp():
    a = _rng_img(23, 31, 3, seed=9)
    rgb, gray = Image.fromarray(a, "RGB"), Image.fromarray(a[..., 0], "L")
    assert np.array_equal(retto_amd.decode_image(_save(rgb, "PPM")), a)
    assert np.array_equal(retto_amd.decode_image(_save(gray, "PPM")), np.repeat(a[..., :1], 3, axis=2))
    assert np.array_equal(retto_amd.decode_image(_save(rgb, "BMP")), a)
    assert np.array_equal(retto_amd.decode_image(_save(gray, "BMP")), np.repeat(a[..., :1], 3, axis=2))
    ascii_ppm = ("P3\n# comment\n%d %d\n255\n" % (31, 23) + " ".join(str(v) for v in a.reshape(-1))).encode()
    assert np.array_equal(retto_amd.decode_image(ascii_ppm), a)
    p16 = b"P5\n3 1\n65535\n" + np.array([0x0081, 0x8000, 0xffff], ">u2").tobytes()
    assert retto_amd.decode_image(p16)[0, :, 0].tolist() == [1, 128, 255]
    with pytest.raises(retto_amd.ImageError):
        retto_amd.decode_image(b"P6\n4 4\n255\n" + bytes(10))


# ---- JPEG -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("subsampling", [0, 1, 2])
@pytest.mark.parametrize("size", [(64, 64), (37, 53), (17, 9), (8, 130), (1, 1), (3, 2)])
@pytest.mark.parametrize("quality", [35, 90])
def test_jpeg_matches_libjpeg(subsampling, size, quality):
    h, w = size
    a = _rng_img(h, w, 3, seed=h + w + subsampling)
    data = _save(Image.fromarray(a, "RGB"), "JPEG", quality=quality, subsampling=subsampling)
    want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    got = retto_amd.decode_image(data)
    assert got.shape == want.shape
    assert np.array_equal(got, want), int(np.abs(got.astype(int) - want).max())


def test_jpeg_grey_optimised_tables_and_restart_intervals():
    a = _rng_img(70, 90, 3, seed=4)
    g = Image.fromarray(a[..., 0], "L")
    for kw in ({"quality": 80}, {"quality": 60, "optimize": True}):
        data = _save(g, "JPEG", **kw)
        assert np.array_equal(retto_amd.decode_image(data), np.asarray(Image.open(io.BytesIO(data)).convert("RGB")))
    rgb = Image.fromarray(a, "RGB")
    for kw in ({"restart_marker_blocks": 3}, {"restart_marker_rows": 1}, {"optimize": True, "subsampling": 2}):
        try:
            data = _save(rgb, "JPEG", quality=75, **kw)
        except TypeError:
            continue
        assert np.array_equal(retto_amd.decode_image(data), np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))), kw


@pytest.mark.parametrize("subsampling", [0, 1, 2])
@pytest.mark.parametrize("size", [(64, 64), (37, 53), (8, 130), (1, 1), (200, 120)])
def test_progressive_jpeg_matches_libjpeg(subsampling, size):
    h, w = size
    a = _rng_img(h, w, 3, seed=h + 3 * w + subsampling)
    for kw in ({"quality": 85}, {"quality": 40, "optimize": True}, {"quality": 95, "restart_marker_blocks": 5}):
        data = _save(Image.fromarray(a, "RGB"), "JPEG", progressive=True, subsampling=subsampling, **kw)
        assert b"\xff\xc2" in data
        want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        got = retto_amd.decode_image(data)
        assert np.array_equal(got, want), (kw, int(np.abs(got.astype(int) - want).max()))
    g = _save(Image.fromarray(a[..., 0], "L"), "JPEG", progressive=True, quality=70)
    assert np.array_equal(retto_amd.decode_image(g), np.asarray(Image.open(io.BytesIO(g)).convert("RGB")))


def test_jpeg_unsupported_and_corrupt():
    a = _rng_img(32, 32, 3, seed=1)
    base = _save(Image.fromarray(a, "RGB"), "JPEG")
    with pytest.raises(retto_amd.ImageError):
        retto_amd.decode_image(base[:30])
    cmyk = _save(Image.fromarray(np.dstack([a, a[..., :1]]), "CMYK"), "JPEG")
    with pytest.raises(retto_amd.ImageError, match="CMYK"):
        retto_amd.decode_image(cmyk)


@pytest.mark.parametrize("head,name", [(b"GIF89a" + b"\0" * 32, "GIF"), (b"RIFF\x10\0\0\0WEBPVP8 " + b"\0" * 16, "WebP"),
                                       (b"II*\0" + b"\0" * 16, "TIFF"), (b"MM\0*" + b"\0" * 16, "TIFF"), (b"qoif" + b"\0" * 16, "QOI")])
def test_formats_the_reference_reads_but_this_decoder_does_not_are_named(head, name):
    """image 0.25.6 with default features (the reference's Cargo.toml:23) also reads GIF / WebP / TIFF / ...; here they fail
    with ImageError naming the format (the documented gap), never with a silent mis-decode."""
    with pytest.raises(retto_amd.ImageError) as e:
        retto_amd.decode_image(head)
    assert name in str(e.value)

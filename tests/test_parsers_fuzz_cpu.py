"""The two parsers of untrusted bytes -- image decoder (image_decode.cpp) and ONNX importer (onnx_import.cpp) --
under AddressSanitizer + UBSan on mutated inputs (tests/fuzz/fuzz_parsers.cpp; CPU only, sanitizers are not available
on the GPU pool).  Any crash, sanitizer report or exception other than RtError fails."""
import io
import os
import subprocess
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    d = tmp_path_factory.mktemp("fuzz")
    exe = str(d / "fuzz_parsers")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-D__HIP_PLATFORM_AMD__",
           "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "fuzz", "fuzz_parsers.cpp"),
           os.path.join(ROOT, "retto_amd", "csrc", "image_decode.cpp"), os.path.join(ROOT, "retto_amd", "csrc", "onnx_import.cpp"),
           "-L" + os.path.join(ROOT, "retto_amd"), "-lretto_hip", "-Wl,-rpath," + os.path.join(ROOT, "retto_amd"), "-lz", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-300:])
    return d, exe


def _run(exe, mode, iters, seed, files):
    r = subprocess.run([exe, mode, str(iters), str(seed)] + files, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    acc, rej = (int(v) for v in r.stdout.split()[0::2][:2])
    assert acc > 0 and rej > 0  # the mutations reach both outcomes
    return acc, rej


def test_image_decoder_survives_mutated_files(harness):
    from PIL import Image
    from test_image_decode_cpu import _rng_img, write_png
    d, exe = harness
    a = _rng_img(40, 56, 3, 1)
    rng = np.random.default_rng(0)
    files = []

    def add(name, data):
        p = d / name
        p.write_bytes(data)
        files.append(str(p))

    def pil(img, fmt, **kw):
        b = io.BytesIO(); img.save(b, fmt, **kw); return b.getvalue()

    add("a.png", pil(Image.fromarray(a), "PNG"))
    add("b.png", pil(Image.fromarray(a).quantize(9), "PNG"))
    add("c.png", write_png(rng.integers(0, 65536, (9, 11, 4)), 6, 16, interlace=True))
    add("d.png", write_png(rng.integers(0, 4, (9, 11, 1)), 0, 2, interlace=True))
    add("e.jpg", pil(Image.fromarray(a), "JPEG", quality=80))
    add("f.jpg", pil(Image.fromarray(a), "JPEG", quality=60, subsampling=2, restart_marker_blocks=2))
    add("g.jpg", pil(Image.fromarray(a[..., 0]), "JPEG", optimize=True))
    add("p.jpg", pil(Image.fromarray(a), "JPEG", quality=70, progressive=True, subsampling=2))
    add("q.jpg", pil(Image.fromarray(a[..., 0]), "JPEG", quality=90, progressive=True, restart_marker_blocks=3))
    add("h.bmp", pil(Image.fromarray(a), "BMP"))
    add("i.bmp", pil(Image.fromarray(a).quantize(16), "BMP"))
    add("j.ppm", pil(Image.fromarray(a), "PPM"))
    add("k.ppm", b"P3\n3 2\n255\n" + b" ".join(str(v).encode() for v in range(18)))
    for seed in (1, 2):
        _run(exe, "image", 40000, seed, files)


def test_onnx_importer_survives_mutated_files(harness):
    import retto_amd
    from retto_amd import synth
    from onnx_writer import build_model_onnx
    d, exe = harness
    files = []
    for style in (0, 1):
        p = d / ("cls%d.onnx" % style)
        p.write_bytes(build_model_onnx(retto_amd.model_manifest(retto_amd.MODEL_CLS), synth.cls_tensors(), seed=2, style=style))
        files.append(str(p))
    _run(exe, "onnx1", 6000, 1, files)

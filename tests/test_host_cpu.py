"""CPU tests of the host-side pieces: the C-ABI library loads and exports every symbol of
include/retto_hip.h, the pure host entry points agree with the oracle, the model-blob
format round-trips, and the product path fails loudly without a GPU (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import retto_amd
from oracle import ref_lib as R
from retto_amd import _lib, synth, workmodel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "retto_hip.h")).read()
    declared = sorted(set(re.findall(r"RT_API\s+[\w\s\*]+?\b(rt_\w+)\s*\(", hdr)))
    assert len(declared) >= 35
    assert sorted(_lib.EXPORTS) == declared, "retto_amd/_lib.py EXPORTS is out of sync with the header"
    for name in declared:
        assert hasattr(lib, name), name


def test_config_defaults_match_reference():
    # session.rs:28-39, det_processor.rs:75-93, cls_processor.rs:27-36, rec_processor.rs:130-134
    c = _lib.Config()
    _lib.load().rt_config_default(C.byref(c))
    assert (c.max_side_len, c.min_side_len) == (2000, 30)
    assert (c.det_limit_side_len, c.det_limit_type, c.det_min_mini_box_size, c.det_dilation) == (736, 0, 3, 1)
    assert list(c.det_mean) == [0.5] * 3 and list(c.det_std) == [0.5] * 3
    assert np.float32(c.det_scale) == np.float32(1.0) / np.float32(255.0)
    assert np.float32(c.det_thresh) == np.float32(0.3) and np.float32(c.det_box_thresh) == np.float32(0.5)
    assert np.float32(c.det_unclip_ratio) == np.float32(1.6)
    assert list(c.cls_image_shape) == [3, 48, 192] and c.cls_batch_num == 6 and np.float32(c.cls_thresh) == np.float32(0.9)
    assert list(c.rec_image_shape) == [3, 48, 320] and c.rec_batch_num == 6


def test_no_cpu_fallback():
    """Without a gfx950 device rt_create must fail with a backend error; with one it works."""
    try:
        s = retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
    except retto_amd.BackendError as e:
        assert "no CPU fallback" in str(e) or "gfx950" in str(e)
    else:
        s.close()


def test_create_rejects_bad_arguments():
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.rt_create(None, C.byref(h)) == 8
    c = _lib.Config(); lib.rt_config_default(C.byref(c)); c.max_boxes_per_page = 100000
    assert lib.rt_create(C.byref(c), C.byref(h)) == 8
    assert b"max_boxes_per_page" in lib.rt_last_error(None)


def test_host_geometry_matches_oracle():
    lib = _lib.load()
    rng = np.random.default_rng(0)
    for _ in range(200):
        c = rng.uniform(20, 400, 2); L = rng.uniform(4, 150); T = rng.uniform(2, 40); a = rng.uniform(-np.pi, np.pi)
        u = np.array([np.cos(a), np.sin(a)]); v = np.array([-u[1], u[0]])
        box = np.round(np.stack([c - L * u - T * v, c + L * u - T * v, c + L * u + T * v, c - L * u + T * v])).astype(np.float32)
        ws, hs = np.zeros(1, np.int32), np.zeros(1, np.int32)
        assert lib.rt_crop_dims(box.ctypes.data, 1, ws.ctypes.data, hs.ctypes.data) == 0
        ow, oh, _ = R.crop_dims(box)
        assert (int(ws[0]), int(hs[0])) == (ow, oh)
        b2 = box.reshape(8).copy()
        lib.rt_scale_and_clip(b2.ctypes.data, 1, 736.0, 736.0, 640.0, 640.0)
        assert np.array_equal(b2.reshape(4, 2), R.scale_and_clip(box, 736, 736, 640, 640))
    for ratio in (0.0, 320 / 48, 7.3, 19.99):
        assert lib.rt_resize_norm_width(48, 320, ratio) == R.lib().orc_resize_norm_width(48, 320, C.c_float(ratio))


def test_blob_roundtrip_and_shapes():
    t = synth.det_tensors(1)
    back = synth.unpack_blob(synth.pack_blob(t))
    assert set(back) == set(t)
    assert all(np.array_equal(back[k], t[k]) for k in t)
    n = lambda d: sum(v.size for v in d.values())
    # parameter counts reproduce the published model sizes (det ~4.7 MB, rec ~10.7 MB fp32)
    assert abs(n(t) * 4 / 1e6 - 4.7) < 0.1
    assert abs(n(synth.rec_tensors(2)) * 4 / 1e6 - 10.7) < 0.2
    d = synth.synth_dict().decode().splitlines()
    assert len(d) == synth.REC_CLASSES - 2 and len(set(d)) == len(d)


def test_work_model_matches_survey():
    det = workmodel.det_work([(960, 960)], phase=False)
    assert abs(sum(v["flops"] for v in det.values()) / 1e9 - 10.36) < 0.02      # 10.36 GFLOP (the reference graph's convs)
    # the launch series as executed since round 4 (nn_fpn.hip: convs over upsampled levels as phase / class convs): 7.59 GFLOP
    assert abs(sum(v["flops"] for v in workmodel.det_work([(960, 960)], phase=True).values()) / 1e9 - 7.59) < 0.02
    rec = workmodel.rec_work([320])
    assert abs(sum(v["flops"] for v in rec.values()) / 1e9 - 1.405) < 0.005     # 1.405 GFLOP per 320-wide line
    assert abs(sum(v["flops"] for v in workmodel.det_work([(736, 736)], phase=False).values()) / 1e9 - 6.09) < 0.02


def test_python_mirror_validates_config():
    cfg = retto_amd.synthetic_session_config(0)
    cfg.det_processor_config.dilation_kernel = np.ones((3, 3))
    with pytest.raises(retto_amd.InvalidArgument):
        retto_amd.RettoSession(cfg)
    cfg = retto_amd.RettoSessionConfig()
    with pytest.raises(retto_amd.ModelNotFoundError):
        retto_amd.RettoSession(cfg)


def test_decode_image_png():
    from PIL import Image
    import io
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (9, 13, 3), dtype=np.uint8)
    buf = io.BytesIO(); Image.fromarray(a).save(buf, format="PNG")
    assert np.array_equal(retto_amd.decode_image(buf.getvalue()), a)
    with pytest.raises(retto_amd.ImageError):
        retto_amd.decode_image(b"not an image")


def test_cli_parser_and_walk(tmp_path):
    from retto_amd import cli
    a = cli.build_parser().parse_args(["--images", str(tmp_path), "--batch", "4", "--synthetic"])
    assert a.batch == 4 and a.synthetic and a.rec_keys_path == "ppocr_keys_v1.txt"
    (tmp_path / "b").mkdir(); (tmp_path / "b" / "2.png").write_bytes(b"x"); (tmp_path / "a.png").write_bytes(b"y")
    assert [os.path.basename(p) for p in cli.walk_files(str(tmp_path))] == ["a.png", "2.png"]


def test_integration_doc_config_matches_header():
    """The Rust #[repr(C)] mirror in INTEGRATION.md and ctypes' Config list the fields of rt_config in the header's order."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "retto_hip.h")).read()
    body = re.search(r"typedef struct rt_config \{(.*?)\} rt_config;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(None, 1)[1]
        fields += [re.sub(r"\[.*?\]", "", n).strip() for n in names.split(",")]
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    rs = re.search(r"pub struct RtConfig \{(.*?)\n\}", doc, re.S).group(1)
    rs_fields = re.findall(r"(\w+)\s*:", rs)
    assert rs_fields == fields
    from retto_amd._lib import Config
    assert [f[0] for f in Config._fields_] == fields


def _parse_dict(data: bytes):
    lib = _lib.load()
    lib.rt_parse_dictionary.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                        C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
    out, ln, n = C.c_void_p(), C.c_size_t(), C.c_int()
    err = C.create_string_buffer(256)
    rc = lib.rt_parse_dictionary(data, len(data), C.byref(out), C.byref(ln), C.byref(n), err, 256)
    if rc != 0:
        return rc, err.value.decode()
    joined = C.string_at(out, ln.value).decode("utf-8")
    lib.rt_buffer_free.argtypes = [C.c_void_p]
    lib.rt_buffer_free(out)
    ents = joined.split("\n")
    assert len(ents) == n.value
    return 0, ents


def test_dictionary_follows_rust_from_utf8_lines_trim():
    """RecCharacter::new (rec_processor.rs:29-46): strict UTF-8, str::lines, str::trim over the Unicode
    White_Space set -- both the product loader and the oracle's."""
    from oracle.pipeline import load_dictionary
    raw = ("a\r\n" "　\n" "  b \t \n" "\n" "c\n" "一　二\n" "\x1cfs\n" "​zw\n" "last").encode("utf-8")
    want = ["blank", "a", "", "b", "", "c", "一　二", "\x1cfs", "​zw", "last", " "]
    # U+3000-only line -> "", inner U+3000 kept, U+001C is NOT White_Space (Python's strip() would remove it),
    # U+200B (zero width space) is not White_Space either; no empty entry after a trailing newline
    rc, ents = _parse_dict(raw)
    assert rc == 0 and ents == want
    assert load_dictionary(raw) == want
    rc, ents = _parse_dict(raw + b"\n")
    assert rc == 0 and ents == want and load_dictionary(raw + b"\n") == want
    assert _parse_dict(b"")[1] == ["blank", " "] == load_dictionary(b"")
    for bad in (b"ok\n\xc0\xaf\n",          # overlong '/'
                b"\xe0\x80\xaf",            # overlong 3-byte
                b"\xed\xa0\x80",            # surrogate U+D800
                b"\xf4\x90\x80\x80",        # > U+10FFFF
                b"\xf8\x88\x80\x80\x80",    # 5-byte form
                b"abc\xe4\xb8",             # truncated
                b"\x80"):                   # stray continuation
        rc, msg = _parse_dict(bad)
        assert rc == 5 and "UTF-8" in msg, bad   # RT_ERR_UTF8 <-> RettoError::Utf8Error
        with pytest.raises(UnicodeDecodeError):
            load_dictionary(bad)
    # the synthetic 6623-line stand-in still gives 6625 classes
    rc, ents = _parse_dict(synth.synth_dict())
    assert rc == 0 and len(ents) == 6625 and ents[0] == "blank" and ents[-1] == " "


def test_json_numbers_are_shortest_round_trip_like_serde_json():
    """serde_json writes f32 through ryu: shortest digits that parse back to the same f32."""
    lib = _lib.load()
    lib.rt_format_f32.argtypes = [C.c_float, C.c_char_p, C.c_size_t]

    def fmt(v):
        b = C.create_string_buffer(64)
        lib.rt_format_f32(float(v), b, 64)
        return b.value.decode()

    known = {0.9: "0.9", 1.0: "1.0", 123.0: "123.0", 0.5: "0.5", 1e-7: "1e-7", 1.234e-7: "1.234e-7", 3.4e38: "3.4e38",
             0.00001234: "0.00001234", 1e13: "1e13", 1e12: "1000000000000.0", 16777216.0: "16777216.0", -2.5: "-2.5",
             1.0 / 3.0: "0.33333334", 0.0: "0.0", 959.0: "959.0"}
    for v, s in known.items():
        assert fmt(v) == s, (v, fmt(v), s)
    assert fmt(float("nan")) == "null" and fmt(float("inf")) == "null"
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.random(2000, dtype=np.float32), (rng.standard_normal(2000) * 1e3).astype(np.float32),
                           rng.integers(0, 4000, 500).astype(np.float32),
                           np.exp(rng.uniform(-80, 80, 2000)).astype(np.float32)])
    for v in vals:
        s = fmt(v)
        assert np.float32(float(s)) == v, (v, s)                      # round trip
        digits = s.replace("-", "").split("e")[0].replace(".", "").strip("0")
        assert len(digits) <= 9
        if len(digits) > 1:                                           # no shorter digit string round-trips
            assert np.float32(float("%.*e" % (len(digits) - 2, float(v)))) != v, (v, s)


def test_rt_create_rejects_a_config_of_another_size():
    """rt_config carries its own size (set by rt_config_default): a host compiled against another version of the header gets
    RT_ERR_INVALID from rt_create instead of fields read past the end of its struct.  (Checked before any device is touched.)"""
    lib = _lib.load()
    c = _lib.Config(); lib.rt_config_default(C.byref(c))
    assert c.struct_size == C.sizeof(_lib.Config)
    c.struct_size -= 4
    out = C.c_void_p()
    lib.rt_create.argtypes = [C.POINTER(_lib.Config), C.POINTER(C.c_void_p)]
    assert lib.rt_create(C.byref(c), C.byref(out)) == 8 and not out.value   # RT_ERR_INVALID


def test_format_f32_ignores_the_locale():
    """rt_format_f32 (the f32 -> JSON text of rt_results_json) must not depend on LC_NUMERIC."""
    import locale
    lib = _lib.load()
    lib.rt_format_f32.restype = C.c_int
    lib.rt_format_f32.argtypes = [C.c_float, C.c_char_p, C.c_size_t]

    def fmt(v):
        b = C.create_string_buffer(64); lib.rt_format_f32(v, b, 64); return b.value.decode()
    want = {0.9: "0.9", 123.0: "123.0", 1.234e-7: "1.234e-7", 1e30: "1e30", 0.00001234: "0.00001234", -2.5: "-2.5"}
    assert {v: fmt(v) for v in want} == want
    old = locale.setlocale(locale.LC_NUMERIC)
    try:
        for name in ("de_DE.UTF-8", "fr_FR.UTF-8", "de_DE", "fr_FR"):
            try:
                locale.setlocale(locale.LC_NUMERIC, name)
                break
            except locale.Error:
                continue
        assert {v: fmt(v) for v in want} == want
    finally:
        locale.setlocale(locale.LC_NUMERIC, old)


# ---------------------------------------------------------------- the pin kit of INTEGRATION.md section 9
def _pin_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pin")


def test_pin_kit_names_in_the_doc_exist():
    """Every pin-kit file and every pin.json key that INTEGRATION.md section 9 tells a maintainer to compare against exists
    (round 2's checklist named keys the fixtures did not hold)."""
    import json
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 9. What a maintainer"):]
    table = sec[:sec.index("Ready to paste")]
    kit = json.load(open(os.path.join(_pin_dir(), "pin.json")))
    files = set(re.findall(r"`([a-z0-9_]+_\d+x\d+(?:x\d+)?\.(?:u8|f32))`", table))
    assert len(files) >= 12
    for f in files:
        assert os.path.exists(os.path.join(_pin_dir(), f)), f

    def resolve(node, path):
        """path like dbpost.*.contours[*].box_score_fast.bits; returns the list of leaves reached"""
        nodes = [node]
        for part in re.findall(r"[A-Za-z0-9_{},]+|\*|\[\*\]", path):
            nxt = []
            for n in nodes:
                if part in ("*", "[*]"):
                    nxt += list(n.values()) if isinstance(n, dict) else list(n)
                elif part.startswith("{"):
                    nxt += [n[k] for k in part.strip("{}").split(",")]
                else:
                    nxt += [n[part]] if not isinstance(n, list) else [m[part] for m in n if part in m]
            nodes = nxt
        return nodes
    keys = set(re.findall(r"`((?:thumbnail|dbpost|crops|reading_order|dictionary|json_f32)[A-Za-z0-9_.*\[\]{},]*)`", table))
    assert len(keys) >= 18
    for key in keys:
        if re.search(r"\.\{[a-z_,]+\}$", key):       # trailing {a,b,c}: each of them
            base, alts = key[:key.rindex(".{")], key[key.rindex(".{") + 2:-1].split(",")
            for a_ in alts:
                assert resolve(kit, base + "." + a_), (key, a_)
        else:
            assert resolve(kit, key), key
    # the Rust module reads exactly these top-level sections
    for k in ("thumbnail", "dbpost", "crops", "reading_order", "dictionary", "json_f32"):
        assert 'k["%s"]' % k in sec or 'kit()["%s"]' % k in sec, k


def test_pin_kit_matches_the_oracle_and_the_library(tmp_path):
    """The committed kit is what the oracle produces today (no drift), and the two rows the library implements on the host
    (dictionary parsing, f32 -> JSON text) give the kit's expectations."""
    import filecmp
    import importlib.util
    import json
    import struct
    spec = importlib.util.spec_from_file_location("make_pin_kit", os.path.join(os.path.dirname(_pin_dir()), "make_pin_kit.py"))
    mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    mk.PIN = str(tmp_path / "pin")
    mk.build()
    names = sorted(os.listdir(_pin_dir()))
    assert names == sorted(os.listdir(mk.PIN))
    for n in names:
        assert filecmp.cmp(os.path.join(_pin_dir(), n), os.path.join(mk.PIN, n), shallow=False), n
    kit = json.load(open(os.path.join(_pin_dir(), "pin.json")))
    rc, classes = _parse_dict(bytes.fromhex(kit["dictionary"]["bytes_hex"]))
    assert rc == 0 and classes == kit["dictionary"]["expected_classes"]
    lib = _lib.load()
    lib.rt_format_f32.argtypes = [C.c_float, C.c_char_p, C.c_size_t]
    for e in kit["json_f32"]:
        v = struct.unpack("<f", struct.pack("<I", int(e["bits"], 16)))[0]
        b = C.create_string_buffer(64); lib.rt_format_f32(v, b, 64)
        assert b.value.decode() == e["expected"]


def test_host_cpu_budget_follows_affinity_quota_and_local_world_size():
    """rt_host_cpu_budget(): affinity mask capped by the cgroup CPU quota, divided by LOCAL_WORLD_SIZE (what sizes the decode
    thread pool of rt_run_encoded_batch: eight ranks on a 16-CPU pod must not start 8 x 16 threads)."""
    import subprocess
    import sys
    from retto_amd import _lib
    from oracle.cpu_baseline import host_cpus as usable_cpus
    lib = _lib.load()
    env = {k: v for k, v in os.environ.items() if k != "LOCAL_WORLD_SIZE"}
    code = "from retto_amd import _lib; print(_lib.load().rt_host_cpu_budget())"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    one = int(subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, check=True).stdout)
    assert one == usable_cpus() >= 1
    eight = int(subprocess.run([sys.executable, "-c", code], env=dict(env, LOCAL_WORLD_SIZE="8"), cwd=root, capture_output=True, text=True, check=True).stdout)
    assert eight == max(1, one // 8)
    assert lib.rt_host_cpu_budget() >= 1


def test_upsampling_aware_conv_identities():
    """The algebra behind retto_amd/csrc/nn_fpn.hip, checked with torch on the CPU (the GPU tests check the kernels): a 3x3 "same"
    conv over a nearest-neighbour upsampled tensor equals (a) for factor 2, four 2 x 2 phase convs of the tensor itself with
    pre-summed taps (fpn_phase_weights), (b) for factor 4 / 8, a lookup by the row / column class (first, interior, last row of an
    upsampling block) of a class conv computed at the coarse resolution (fpn_class_weights) -- zero padding included; and (c) a
    bias-free 1x1 conv, a per-channel scale and a 3x3 conv compose into one 3x3 conv (fpn_compose)."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    z = torch.randn(1, 6, 5, 7, generator=g, dtype=torch.float64)
    w = torch.randn(4, 6, 3, 3, generator=g, dtype=torch.float64)

    def taps(phase, t, s):   # taps of the 3-tap axis that land on relative coarse row t - 1 + ... for an output row at `phase` of an s-block
        # returns {relative coarse row: [taps]} for phase in [0, s)
        out = {}
        for d in range(3):
            rel = (phase + d - 1) // s          # floor division: -1, 0 or +1
            out.setdefault(rel, []).append(d)
        return out

    for s in (2, 4, 8):
        ref = F.conv2d(F.interpolate(z, scale_factor=s, mode="nearest"), w, padding=1)
        H, W = z.shape[2:]
        got = torch.zeros_like(ref)
        zp = F.pad(z, (1, 1, 1, 1))
        for py in range(s):
            ty = taps(py, 0, s)
            for px in range(s):
                tx = taps(px, 0, s)
                acc = torch.zeros(1, 4, H, W, dtype=torch.float64)
                for ry, dys in ty.items():
                    for rx, dxs in tx.items():
                        weff = sum(w[:, :, dy, dx] for dy in dys for dx in dxs)          # pre-summed taps: [n, k]
                        src = zp[:, :, 1 + ry:1 + ry + H, 1 + rx:1 + rx + W]
                        acc += torch.einsum("nk,bkhw->bnhw", weff, src)
                got[:, :, py::s, px::s] = acc
                # the class of a phase: first (0), interior (1), last (2) -- only these three tap patterns exist per axis
                assert sorted(ty) == ([-1, 0] if py == 0 else [0, 1] if py == s - 1 else [0])
        assert torch.allclose(got, ref, atol=1e-12)
    # (c) lateral 1x1 (no bias) * scale -> 3x3 conv == 3x3 conv of the narrow tensor with composed weights
    x = torch.randn(1, 3, 6, 6, generator=g, dtype=torch.float64)
    wl = torch.randn(8, 3, 1, 1, generator=g, dtype=torch.float64)
    sc = torch.rand(8, generator=g, dtype=torch.float64) + 0.5
    wc = torch.randn(4, 8, 3, 3, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.conv2d(x, wl) * sc.view(1, 8, 1, 1), wc, padding=1)
    wcomp = torch.einsum("nmyx,m,mc->ncyx", wc, sc, wl[:, :, 0, 0])
    assert torch.allclose(F.conv2d(x, wcomp, padding=1), ref, atol=1e-12)

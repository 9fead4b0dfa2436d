"""Synthetic model artefacts in the RTWB weight-blob format.

The reference resolves three ONNX files and a dictionary through
``RettoWorkerModelSource::{Path,Blob,HuggingFace}``
(/root/reference/retto-core/src/worker.rs:18-56,
/root/reference/retto-core/src/worker/ort_worker.rs:58-110).  Neither the
``.onnx`` files nor ``ppocr_keys_v1.txt`` exist offline, so throughput and
parity runs use *seeded synthetic weights in the exact PP-OCRv4 layer shapes*
(SURVEY.md Appendix C).  This module only writes bytes; it never computes a
forward pass.

RTWB v1 container (little endian)::

    "RTWB" u32 version u32 n_tensors u32 reserved
    n x { u16 name_len, name, u8 ndim, u8 dtype(0=f32), u16 0,
          u32 dims[ndim], u64 offset, u64 nbytes }
    pad to 64 B, then the data section (offsets relative to its start, each
    tensor 64-B aligned)

Tensor conventions (inference form: BN and the first LearnableAffineBlock
are folded into the preceding conv):
    conv        <name>.w [cout, cin/groups, kh, kw]   <name>.b [cout]
    convT 2x2   <name>.w [cin, cout, 2, 2]            <name>.b [cout]
    linear      <name>.w [in, out]                    <name>.b [out]
    post-activation scalar affine (LCNetV3 "LAB"): <name>.a [1], <name>.c [1]
    layer norm  <name>.g [C], <name>.beta [C]
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np

MAGIC = b"RTWB"

# ---------------------------------------------------------------------------
# Architecture tables (PaddleOCR PP-OCRv4 mobile)
# ---------------------------------------------------------------------------
# (kernel, cin, cout, stride_h, stride_w, use_se) after channel scaling.
DET_BLOCKS: List[Tuple[str, int, int, int, int, int, bool]] = [
    ("s2.0", 3, 16, 32, 1, 1, False),
    ("s3.0", 3, 32, 48, 2, 2, False), ("s3.1", 3, 48, 48, 1, 1, False),
    ("s4.0", 3, 48, 96, 2, 2, False), ("s4.1", 3, 96, 96, 1, 1, False),
    ("s5.0", 3, 96, 192, 2, 2, False), ("s5.1", 5, 192, 192, 1, 1, False),
    ("s5.2", 5, 192, 192, 1, 1, False), ("s5.3", 5, 192, 192, 1, 1, False),
    ("s5.4", 5, 192, 192, 1, 1, False),
    ("s6.0", 5, 192, 384, 2, 2, True), ("s6.1", 5, 384, 384, 1, 1, True),
    ("s6.2", 5, 384, 384, 1, 1, False), ("s6.3", 5, 384, 384, 1, 1, False),
]
# block after which a pyramid level is tapped, its 1x1 projection width
DET_TAPS = [("s3.1", 48, 12), ("s4.1", 96, 18), ("s5.4", 192, 42), ("s6.3", 384, 360)]
DET_FPN_CH = 96

REC_BLOCKS: List[Tuple[str, int, int, int, int, int, bool]] = [
    ("s2.0", 3, 16, 32, 1, 1, False),
    ("s3.0", 3, 32, 64, 1, 1, False), ("s3.1", 3, 64, 64, 1, 1, False),
    ("s4.0", 3, 64, 128, 2, 1, False), ("s4.1", 3, 128, 128, 1, 1, False),
    ("s5.0", 3, 128, 240, 1, 2, False), ("s5.1", 5, 240, 240, 1, 1, False),
    ("s5.2", 5, 240, 240, 1, 1, False), ("s5.3", 5, 240, 240, 1, 1, False),
    ("s5.4", 5, 240, 240, 1, 1, False),
    ("s6.0", 5, 240, 480, 2, 1, True), ("s6.1", 5, 480, 480, 1, 1, True),
    ("s6.2", 5, 480, 480, 2, 1, False), ("s6.3", 5, 480, 480, 1, 1, False),
]
REC_NECK_DIM = 120
REC_HEADS = 8
REC_CLASSES = 6625

# MobileNetV3-small x0.35: (k, mid, cout, se, act, stride_h, stride_w); cin chained from 8
CLS_BLOCKS = [
    (3, 8, 8, True, "relu", 2, 1), (3, 24, 8, False, "relu", 2, 1), (3, 32, 8, False, "relu", 1, 1),
    (5, 32, 16, True, "hswish", 2, 1), (5, 88, 16, True, "hswish", 1, 1), (5, 88, 16, True, "hswish", 1, 1),
    (5, 40, 16, True, "hswish", 1, 1), (5, 48, 16, True, "hswish", 1, 1), (5, 104, 32, True, "hswish", 2, 1),
    (5, 200, 32, True, "hswish", 1, 1), (5, 200, 32, True, "hswish", 1, 1),
]
CLS_STEM = 8
CLS_LAST = 200


def det_dw_has_act(sh: int, sw: int) -> bool:
    """LearnableRepLayer applies its activation unless ``stride == 2`` (det
    strides are ints; rec strides are tuples and never compare equal to 2)."""
    return not (sh == 2 and sw == 2)


class _Gen:
    def __init__(self, seed: int):
        self.rng = np.random.default_rng(seed)
        self.t: Dict[str, np.ndarray] = {}

    def conv(self, name, cout, cin_g, kh, kw, act=True, bias=True, gain=1.0):
        # gain^2 = 2.4 (between He's 2 for relu and 3 for unit-variance hardswish) keeps
        # the activation scale of the LCNetV3 stacks O(1): smaller collapses the signal,
        # larger explodes it and turns fp32 rounding noise into visible output differences.
        fan_in = cin_g * kh * kw
        std = gain * np.sqrt((2.4 if act else 1.0) / fan_in)
        self.t[name + ".w"] = (self.rng.standard_normal((cout, cin_g, kh, kw)) * std).astype(np.float32)
        if bias:
            self.t[name + ".b"] = (self.rng.standard_normal(cout) * 0.02).astype(np.float32)

    def linear(self, name, cin, cout, gain=1.0):
        self.t[name + ".w"] = (self.rng.standard_normal((cin, cout)) * gain / np.sqrt(cin)).astype(np.float32)
        self.t[name + ".b"] = (self.rng.standard_normal(cout) * 0.05).astype(np.float32)

    def lab(self, name):
        self.t[name + ".a"] = self.rng.uniform(0.9, 1.1, 1).astype(np.float32)
        self.t[name + ".c"] = (self.rng.standard_normal(1) * 0.05).astype(np.float32)

    def ln(self, name, c):
        self.t[name + ".g"] = self.rng.uniform(0.9, 1.1, c).astype(np.float32)
        self.t[name + ".beta"] = (self.rng.standard_normal(c) * 0.05).astype(np.float32)

    def se(self, name, c, r=4):
        self.conv(name + ".fc1", c // r, c, 1, 1)
        self.conv(name + ".fc2", c, c // r, 1, 1, act=False)


def _lcnet(g: _Gen, prefix: str, blocks, det: bool):
    g.conv(prefix + ".stem", 16, 3, 3, 3, act=False, gain=1.7)
    for name, k, cin, cout, sh, sw, se in blocks:
        p = f"{prefix}.{name}"
        g.conv(p + ".dw", cin, 1, k, k)
        if (not det) or det_dw_has_act(sh, sw):
            g.lab(p + ".dw")
        if se:
            g.se(p + ".se", cin)
        g.conv(p + ".pw", cout, cin, 1, 1, gain=1.6 if se else 1.0)  # SE gate ~0.5
        g.lab(p + ".pw")


def det_tensors(seed: int = 1) -> Dict[str, np.ndarray]:
    g = _Gen(seed)
    _lcnet(g, "det", DET_BLOCKS, det=True)
    for j, (_, cin, cout) in enumerate(DET_TAPS):
        g.conv(f"det.out{j}", cout, cin, 1, 1, act=False)
        g.conv(f"det.fpn.ins{j}", DET_FPN_CH, cout, 1, 1, act=False, bias=False)
        g.se(f"det.fpn.ins{j}.se", DET_FPN_CH)
        g.conv(f"det.fpn.inp{j}", DET_FPN_CH // 4, DET_FPN_CH, 3, 3, act=False, bias=False)
        g.se(f"det.fpn.inp{j}.se", DET_FPN_CH // 4)
    g.conv("det.head.conv1", 24, 96, 3, 3)
    std = np.sqrt(2.0 / 24)
    g.t["det.head.deconv1.w"] = (g.rng.standard_normal((24, 24, 2, 2)) * std).astype(np.float32)
    g.t["det.head.deconv1.b"] = (g.rng.standard_normal(24) * 0.05).astype(np.float32)
    g.t["det.head.deconv2.w"] = (g.rng.standard_normal((24, 1, 2, 2)) * 0.3 * np.sqrt(1.0 / 24)).astype(np.float32)
    g.t["det.head.deconv2.b"] = (g.rng.standard_normal(1) * 0.05).astype(np.float32)
    return g.t


def rec_tensors(seed: int = 2) -> Dict[str, np.ndarray]:
    g = _Gen(seed)
    _lcnet(g, "rec", REC_BLOCKS, det=False)
    C, D = 480, REC_NECK_DIM
    g.conv("rec.neck.conv1", C // 8, C, 1, 3)
    g.conv("rec.neck.conv2", D, C // 8, 1, 1)
    for i in range(2):
        p = f"rec.neck.blk{i}"
        g.linear(p + ".qkv", D, 3 * D)
        g.linear(p + ".proj", D, D)
        g.ln(p + ".norm1", D)
        g.linear(p + ".fc1", D, 2 * D, gain=np.sqrt(2.0))
        g.linear(p + ".fc2", 2 * D, D)
        g.ln(p + ".norm2", D)
    g.ln("rec.neck.norm", D)
    g.conv("rec.neck.conv3", C, D, 1, 1)
    g.conv("rec.neck.conv4", C // 8, 2 * C, 1, 3)
    g.conv("rec.neck.conv1x1", D, C // 8, 1, 1)
    g.linear("rec.head.fc", D, REC_CLASSES, gain=3.0)
    return g.t


def cls_tensors(seed: int = 3) -> Dict[str, np.ndarray]:
    g = _Gen(seed)
    g.conv("cls.stem", CLS_STEM, 3, 3, 3)
    cin = CLS_STEM
    for i, (k, mid, cout, se, act, sh, sw) in enumerate(CLS_BLOCKS):
        p = f"cls.b{i}"
        g.conv(p + ".expand", mid, cin, 1, 1)
        g.conv(p + ".dw", mid, 1, k, k)
        if se:
            g.se(p + ".se", mid)
        g.conv(p + ".linear", cout, mid, 1, 1, act=False, gain=1.6 if se else 1.0)
        cin = cout
    g.conv("cls.conv2", CLS_LAST, cin, 1, 1)
    g.linear("cls.head.fc", CLS_LAST, 2, gain=3.0)
    return g.t


# ---------------------------------------------------------------------------
# PP-OCRv4 *server* graphs (BASELINE.json config 5; SURVEY.md Appendix C "Server"): PPHGNet_small backbones
# (ppocr/modeling/backbones/rec_hgnet.py), det neck LKPAN(256, large, intracl) + head PFHeadLocal(k=50, large),
# rec neck / head as the mobile model (EncoderWithSVTR on 1024 channels + CTC).  Inference form: every
# ConvBNAct is one conv with bias (BN folded).
# ---------------------------------------------------------------------------
HG_STEM = [64, 64, 128]
HG_LAYERS = 6
# name, in, mid, out, blocks, downsample
HG_STAGES_DET = [("st1", 128, 128, 256, 1, False), ("st2", 256, 160, 512, 1, True),
                 ("st3", 512, 192, 768, 2, True), ("st4", 768, 224, 1024, 1, True)]
HG_STAGES_REC = [("st1", 128, 128, 256, 1, True), ("st2", 256, 160, 512, 1, True),
                 ("st3", 512, 192, 768, 2, True), ("st4", 768, 224, 1024, 1, True)]
HG_STRIDES_DET = [(2, 2)] * 4
HG_STRIDES_REC = [(2, 1), (1, 2), (2, 1), (2, 1)]
LKPAN_CH = 256
SREC_NECK_IN = 1024


def _hgnet(g: _Gen, prefix: str, stages):
    cin = 3
    for i, c in enumerate(HG_STEM):
        g.conv(f"{prefix}.stem{i}", c, cin, 3, 3, gain=1.7 if i == 0 else 1.0)
        cin = c
    for name, cin, mid, cout, blocks, down in stages:
        p = f"{prefix}.{name}"
        if down:
            g.conv(p + ".ds", cin, 1, 3, 3, act=False)      # depthwise ConvBNAct(use_act=False)
        for b in range(blocks):
            bin_ = cin if b == 0 else cout
            c = bin_
            for l in range(HG_LAYERS):
                g.conv(f"{p}.b{b}.l{l}", mid, c, 3, 3, gain=0.92)
                c = mid
            g.conv(f"{p}.b{b}.agg", cout, bin_ + HG_LAYERS * mid, 1, 1, gain=1.25)   # ESE gate ~0.5 follows
            g.conv(f"{p}.b{b}.ese", cout, cout, 1, 1, act=False, gain=0.5)


def sdet_tensors(seed: int = 4) -> Dict[str, np.ndarray]:
    g = _Gen(seed)
    _hgnet(g, "sdet", HG_STAGES_DET)
    C, Q = LKPAN_CH, LKPAN_CH // 4
    for i, (_, _, _, cout, _, _) in enumerate(HG_STAGES_DET):
        g.conv(f"sdet.neck.ins{i}", C, cout, 1, 1, act=False, bias=False)
        g.conv(f"sdet.neck.inp{i}", Q, C, 9, 9, act=False, bias=False)
        g.conv(f"sdet.neck.panlat{i}", Q, Q, 9, 9, act=False, bias=False)
        if i > 0:
            g.conv(f"sdet.neck.panhead{i - 1}", Q, Q, 3, 3, act=False, bias=False)
    for i in range(1, 5):                       # IntraCLBlock(64, reduce_factor=2)
        p = f"sdet.neck.incl{i}"
        R = Q // 2
        g.conv(p + ".reduce", R, Q, 1, 1, act=False)
        for k in (7, 5, 3):
            g.conv(f"{p}.c{k}", R, R, k, k, act=False, gain=0.6)
            g.conv(f"{p}.v{k}", R, R, k, 1, act=False, gain=0.6)
            g.conv(f"{p}.q{k}", R, R, 1, k, act=False, gain=0.6)
        g.conv(p + ".ret", Q, R, 1, 1)
    g.conv("sdet.head.conv1", Q, C, 3, 3)
    g.t["sdet.head.deconv1.w"] = (g.rng.standard_normal((Q, Q, 2, 2)) * np.sqrt(2.0 / Q)).astype(np.float32)
    g.t["sdet.head.deconv1.b"] = (g.rng.standard_normal(Q) * 0.05).astype(np.float32)
    g.t["sdet.head.deconv2.w"] = (g.rng.standard_normal((Q, 1, 2, 2)) * 0.3 * np.sqrt(1.0 / Q)).astype(np.float32)
    g.t["sdet.head.deconv2.b"] = (g.rng.standard_normal(1) * 0.05).astype(np.float32)
    g.conv("sdet.head.local3", Q, Q + 1, 3, 3)
    g.conv("sdet.head.local1", 1, Q, 1, 1, act=False, gain=0.3)
    return g.t


def srec_tensors(seed: int = 5) -> Dict[str, np.ndarray]:
    g = _Gen(seed)
    _hgnet(g, "srec", HG_STAGES_REC)
    C, D = SREC_NECK_IN, REC_NECK_DIM
    g.conv("srec.neck.conv1", C // 8, C, 1, 3)
    g.conv("srec.neck.conv2", D, C // 8, 1, 1)
    for i in range(2):
        p = f"srec.neck.blk{i}"
        g.linear(p + ".qkv", D, 3 * D)
        g.linear(p + ".proj", D, D)
        g.ln(p + ".norm1", D)
        g.linear(p + ".fc1", D, 2 * D, gain=np.sqrt(2.0))
        g.linear(p + ".fc2", 2 * D, D)
        g.ln(p + ".norm2", D)
    g.ln("srec.neck.norm", D)
    g.conv("srec.neck.conv3", C, D, 1, 1)
    g.conv("srec.neck.conv4", C // 8, 2 * C, 1, 3)
    g.conv("srec.neck.conv1x1", D, C // 8, 1, 1)
    g.linear("srec.head.fc", D, REC_CLASSES, gain=8.0)   # peaked like the mobile head (max prob ~0.5): token ids are decisive
    return g.t


def pack_blob(tensors: Dict[str, np.ndarray]) -> bytes:
    names = list(tensors.keys())
    table = bytearray()
    off = 0
    offs = []
    for n in names:
        a = np.ascontiguousarray(tensors[n], dtype=np.float32)
        offs.append((off, a.nbytes))
        off = (off + a.nbytes + 63) // 64 * 64
    for n, (o, nb) in zip(names, offs):
        a = tensors[n]
        nb_name = n.encode("utf-8")
        table += struct.pack("<H", len(nb_name)) + nb_name
        table += struct.pack("<BBH", a.ndim, 0, 0)
        table += struct.pack("<%dI" % a.ndim, *a.shape)
        table += struct.pack("<QQ", o, nb)
    head = MAGIC + struct.pack("<III", 1, len(names), 0)
    pre = head + bytes(table)
    pad = (-len(pre)) % 64
    data = bytearray(off)
    for n, (o, nb) in zip(names, offs):
        data[o:o + nb] = np.ascontiguousarray(tensors[n], dtype=np.float32).tobytes()
    return pre + b"\0" * pad + bytes(data)


def unpack_blob(blob: bytes) -> Dict[str, np.ndarray]:
    if blob[:4] != MAGIC:
        raise ValueError("not an RTWB blob")
    ver, n, _ = struct.unpack_from("<III", blob, 4)
    if ver != 1:
        raise ValueError("unsupported RTWB version %d" % ver)
    p = 16
    ents = []
    for _ in range(n):
        (ln,) = struct.unpack_from("<H", blob, p); p += 2
        name = blob[p:p + ln].decode("utf-8"); p += ln
        ndim, dt, _z = struct.unpack_from("<BBH", blob, p); p += 4
        dims = struct.unpack_from("<%dI" % ndim, blob, p); p += 4 * ndim
        o, nb = struct.unpack_from("<QQ", blob, p); p += 16
        ents.append((name, dims, o, nb))
    base = (p + 63) // 64 * 64
    out = {}
    for name, dims, o, nb in ents:
        out[name] = np.frombuffer(blob, dtype=np.float32, count=nb // 4, offset=base + o).reshape(dims).copy()
    return out


def synth_dict(n_lines: int = REC_CLASSES - 2) -> bytes:
    """A stand-in for ``ppocr_keys_v1.txt`` (6623 lines; the reference inserts
    ``blank`` at 0 and appends ``" "`` -> 6625 classes,
    /root/reference/retto-core/src/processor/rec_processor.rs:29-46)."""
    chars = [chr(0x4E00 + i) for i in range(n_lines)]
    return ("\n".join(chars) + "\n").encode("utf-8")


def synth_models(seed: int = 0):
    """Returns (det_blob, cls_blob, rec_blob, dict_bytes)."""
    return (pack_blob(det_tensors(seed * 10 + 1)), pack_blob(cls_tensors(seed * 10 + 3)),
            pack_blob(rec_tensors(seed * 10 + 2)), synth_dict())


def synth_server_models(seed: int = 0):
    """PP-OCRv4 server det / rec (the angle classifier stays the mobile one, as in PaddleOCR's server pipeline).
    Returns (det_blob, cls_blob, rec_blob, dict_bytes)."""
    return (pack_blob(sdet_tensors(seed * 10 + 4)), pack_blob(cls_tensors(seed * 10 + 3)),
            pack_blob(srec_tensors(seed * 10 + 5)), synth_dict())

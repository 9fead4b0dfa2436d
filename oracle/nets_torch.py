"""oracle/nets_torch.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Plain torch-CPU fp32 forward passes of the three networks the reference runs
through ONNX Runtime (``RettoInnerWorker::{det,cls,rec}``,
/root/reference/retto-core/src/worker.rs:69-73, impl
/root/reference/retto-core/src/worker/ort_worker.rs:189-220).  The graphs are
not in /root/reference (model files are downloaded at run time,
/root/reference/retto-core/build.rs:7-12); the architecture follows the public
PaddleOCR PP-OCRv4 mobile definitions as recorded in SURVEY.md Appendix C and
is driven by an RTWB weight blob (format: retto_amd/synth.py docstring).

PARITY UNPINNED: no reference tensor fixtures exist; this file is the fp32
oracle of record for the HIP kernels (tolerance stated in the tests).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.
"""
from __future__ import annotations

import struct
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

# --- architecture tables (restated) ---------------
DET_BLOCKS = [
    ("s2.0", 3, 16, 32, (1, 1), False),
    ("s3.0", 3, 32, 48, (2, 2), False), ("s3.1", 3, 48, 48, (1, 1), False),
    ("s4.0", 3, 48, 96, (2, 2), False), ("s4.1", 3, 96, 96, (1, 1), False),
    ("s5.0", 3, 96, 192, (2, 2), False), ("s5.1", 5, 192, 192, (1, 1), False),
    ("s5.2", 5, 192, 192, (1, 1), False), ("s5.3", 5, 192, 192, (1, 1), False),
    ("s5.4", 5, 192, 192, (1, 1), False),
    ("s6.0", 5, 192, 384, (2, 2), True), ("s6.1", 5, 384, 384, (1, 1), True),
    ("s6.2", 5, 384, 384, (1, 1), False), ("s6.3", 5, 384, 384, (1, 1), False),
]
DET_TAP_AFTER = {"s3.1": 0, "s4.1": 1, "s5.4": 2, "s6.3": 3}
REC_BLOCKS = [
    ("s2.0", 3, 16, 32, (1, 1), False),
    ("s3.0", 3, 32, 64, (1, 1), False), ("s3.1", 3, 64, 64, (1, 1), False),
    ("s4.0", 3, 64, 128, (2, 1), False), ("s4.1", 3, 128, 128, (1, 1), False),
    ("s5.0", 3, 128, 240, (1, 2), False), ("s5.1", 5, 240, 240, (1, 1), False),
    ("s5.2", 5, 240, 240, (1, 1), False), ("s5.3", 5, 240, 240, (1, 1), False),
    ("s5.4", 5, 240, 240, (1, 1), False),
    ("s6.0", 5, 240, 480, (2, 1), True), ("s6.1", 5, 480, 480, (1, 1), True),
    ("s6.2", 5, 480, 480, (2, 1), False), ("s6.3", 5, 480, 480, (1, 1), False),
]
CLS_BLOCKS = [
    (3, 8, 8, True, "relu", (2, 1)), (3, 24, 8, False, "relu", (2, 1)), (3, 32, 8, False, "relu", (1, 1)),
    (5, 32, 16, True, "hswish", (2, 1)), (5, 88, 16, True, "hswish", (1, 1)), (5, 88, 16, True, "hswish", (1, 1)),
    (5, 40, 16, True, "hswish", (1, 1)), (5, 48, 16, True, "hswish", (1, 1)), (5, 104, 32, True, "hswish", (2, 1)),
    (5, 200, 32, True, "hswish", (1, 1)), (5, 200, 32, True, "hswish", (1, 1)),
]


def read_blob(blob: bytes) -> Dict[str, torch.Tensor]:
    assert blob[:4] == b"RTWB"
    ver, n, _ = struct.unpack_from("<III", blob, 4)
    assert ver == 1
    p = 16
    ents = []
    for _ in range(n):
        (ln,) = struct.unpack_from("<H", blob, p); p += 2
        name = blob[p:p + ln].decode(); p += ln
        ndim, _dt, _z = struct.unpack_from("<BBH", blob, p); p += 4
        dims = struct.unpack_from("<%dI" % ndim, blob, p); p += 4 * ndim
        o, nb = struct.unpack_from("<QQ", blob, p); p += 16
        ents.append((name, dims, o, nb))
    base = (p + 63) // 64 * 64
    return {name: torch.from_numpy(np.frombuffer(blob, np.float32, nb // 4, base + o).reshape(dims).copy())
            for name, dims, o, nb in ents}


# --- activations (Paddle definitions) ---------------------------------------
def hswish(x):  # x * relu6(x + 3) / 6
    return x * torch.clamp(x + 3.0, 0.0, 6.0) / 6.0


def hsigmoid(x, slope, offset=0.5):  # clip(slope * x + offset, 0, 1)
    return torch.clamp(x * slope + offset, 0.0, 1.0)


def swish(x):
    return x * torch.sigmoid(x)


HSIG_LCNET = 0.1666667  # nn.Hardsigmoid default slope in Paddle
HSIG_MBV3 = 0.2         # F.hardsigmoid(slope=0.2, offset=0.5) in det/rec_mobilenet_v3 SEModule


def _conv(w, name, x, stride=(1, 1), pad=(0, 0), groups=1):
    return F.conv2d(x, w[name + ".w"], w.get(name + ".b"), stride=stride, padding=pad, groups=groups)


def _se_scale(w, name, x, slope):
    s = x.mean(dim=(2, 3), keepdim=True)
    s = F.relu(_conv(w, name + ".fc1", s))
    return hsigmoid(_conv(w, name + ".fc2", s), slope)


def _lcnet_block(w, p, x, k, stride, se, dw_act):
    cin = x.shape[1]
    x = _conv(w, p + ".dw", x, stride=stride, pad=(k // 2, k // 2), groups=cin)
    if dw_act:
        x = hswish(x) * w[p + ".dw.a"] + w[p + ".dw.c"]
    if se:
        x = x * _se_scale(w, p + ".se", x, HSIG_LCNET)
    x = _conv(w, p + ".pw", x)
    return hswish(x) * w[p + ".pw.a"] + w[p + ".pw.c"]


@torch.no_grad()
def det_forward(w: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """x [N,3,H,W] (BGR, normalised) -> probability map [N,1,H,W]."""
    x = _conv(w, "det.stem", x, stride=(2, 2), pad=(1, 1))
    taps = [None] * 4
    for name, k, _cin, _cout, stride, se in DET_BLOCKS:
        x = _lcnet_block(w, "det." + name, x, k, stride, se, dw_act=(stride != (2, 2)))
        if name in DET_TAP_AFTER:
            j = DET_TAP_AFTER[name]
            taps[j] = _conv(w, f"det.out{j}", x)

    def rse(name, t, pad):
        t = _conv(w, name, t, pad=pad)
        return t + t * _se_scale(w, name + ".se", t, HSIG_MBV3)

    in5 = rse("det.fpn.ins3", taps[3], (0, 0)); in4 = rse("det.fpn.ins2", taps[2], (0, 0))
    in3 = rse("det.fpn.ins1", taps[1], (0, 0)); in2 = rse("det.fpn.ins0", taps[0], (0, 0))
    up = lambda t, s: F.interpolate(t, scale_factor=s, mode="nearest")
    out4 = in4 + up(in5, 2); out3 = in3 + up(out4, 2); out2 = in2 + up(out3, 2)
    p5 = rse("det.fpn.inp3", in5, (1, 1)); p4 = rse("det.fpn.inp2", out4, (1, 1))
    p3 = rse("det.fpn.inp1", out3, (1, 1)); p2 = rse("det.fpn.inp0", out2, (1, 1))
    fuse = torch.cat([up(p5, 8), up(p4, 4), up(p3, 2), p2], dim=1)
    y = F.relu(_conv(w, "det.head.conv1", fuse, pad=(1, 1)))
    y = F.relu(F.conv_transpose2d(y, w["det.head.deconv1.w"], w["det.head.deconv1.b"], stride=2))
    y = F.conv_transpose2d(y, w["det.head.deconv2.w"], w["det.head.deconv2.b"], stride=2)
    return torch.sigmoid(y)


def _ln(w, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), w[name + ".g"], w[name + ".beta"], eps)


@torch.no_grad()
def rec_forward(w: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """x [n,3,48,W] (RGB, normalised, zero padded) -> softmax probs [n,T,6625]."""
    x = _conv(w, "rec.stem", x, stride=(2, 2), pad=(1, 1))
    for name, k, _cin, _cout, stride, se in REC_BLOCKS:
        x = _lcnet_block(w, "rec." + name, x, k, stride, se, dw_act=True)
    x = F.avg_pool2d(x, (3, 2))
    return _svtr_ctc(w, "rec", x)


def _svtr_ctc(w, prefix, x):
    """EncoderWithSVTR (dims 120, depth 2, 8 heads, kernel [1,3], use_guide) + CTC head on the pooled backbone
    feature x [B, C, 1, T]; tensors named <prefix>.neck.* / <prefix>.head.fc."""
    h = x
    z = swish(_conv(w, prefix + ".neck.conv1", x, pad=(0, 1)))
    z = swish(_conv(w, prefix + ".neck.conv2", z))
    B, C, H, W = z.shape
    z = z.flatten(2).transpose(1, 2)  # [B, T, C]
    nh, hd = 8, C // 8
    for i in range(2):
        p = f"{prefix}.neck.blk{i}"
        qkv = (z @ w[p + ".qkv.w"] + w[p + ".qkv.b"]).reshape(B, -1, 3, nh, hd).permute(2, 0, 3, 1, 4)
        q, k_, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
        attn = torch.softmax(q @ k_.transpose(-2, -1), dim=-1)
        a = (attn @ v).transpose(1, 2).reshape(B, -1, C)
        a = a @ w[p + ".proj.w"] + w[p + ".proj.b"]
        z = _ln(w, p + ".norm1", z + a, 1e-5)
        m = swish(z @ w[p + ".fc1.w"] + w[p + ".fc1.b"]) @ w[p + ".fc2.w"] + w[p + ".fc2.b"]
        z = _ln(w, p + ".norm2", z + m, 1e-5)
    z = _ln(w, prefix + ".neck.norm", z, 1e-6)
    z = z.reshape(B, H, W, C).permute(0, 3, 1, 2)
    z = swish(_conv(w, prefix + ".neck.conv3", z))
    z = torch.cat([h, z], dim=1)
    z = swish(_conv(w, prefix + ".neck.conv4", z, pad=(0, 1)))
    z = swish(_conv(w, prefix + ".neck.conv1x1", z))
    z = z.squeeze(2).transpose(1, 2)  # Im2Seq: [B, T, 120]
    logits = z @ w[prefix + ".head.fc.w"] + w[prefix + ".head.fc.b"]
    return torch.softmax(logits, dim=2)


@torch.no_grad()
def cls_forward(w: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """x [n,3,48,192] -> softmax probs [n,2]."""
    x = hswish(_conv(w, "cls.stem", x, stride=(2, 2), pad=(1, 1)))
    for i, (k, _mid, cout, se, act, stride) in enumerate(CLS_BLOCKS):
        p = f"cls.b{i}"
        a = F.relu if act == "relu" else hswish
        cin = x.shape[1]
        y = a(_conv(w, p + ".expand", x))
        y = a(_conv(w, p + ".dw", y, stride=stride, pad=(k // 2, k // 2), groups=y.shape[1]))
        if se:
            y = y * _se_scale(w, p + ".se", y, HSIG_MBV3)
        y = _conv(w, p + ".linear", y)
        x = x + y if (stride == (1, 1) and cin == cout) else y
    x = hswish(_conv(w, "cls.conv2", x))
    x = F.max_pool2d(x, 2, 2)
    x = x.mean(dim=(2, 3))
    return torch.softmax(x @ w["cls.head.fc.w"] + w["cls.head.fc.b"], dim=1)


# ---------------------------------------------------------------------------
# PP-OCRv4 server graphs (BASELINE.json config 5; not named by the reference, SURVEY.md Appendix C "Server"):
# PPHGNet_small backbone (ppocr/modeling/backbones/rec_hgnet.py), LKPAN(256, large, intracl) neck and PFHeadLocal head
# for detection; the mobile model's SVTR neck / CTC head on 1024 channels for recognition.  Parameter counts
# reproduce the published inference models (det ~113 MB, rec ~90 MB fp32).  PARITY UNPINNED (as above).
# ---------------------------------------------------------------------------
HG_STEM = 3
HG_LAYERS = 6
HG_STAGES_DET = [("st1", 1, False, (2, 2)), ("st2", 1, True, (2, 2)), ("st3", 2, True, (2, 2)), ("st4", 1, True, (2, 2))]
HG_STAGES_REC = [("st1", 1, True, (2, 1)), ("st2", 1, True, (1, 2)), ("st3", 2, True, (2, 1)), ("st4", 1, True, (2, 1))]


def _hg_block(w, p, x, identity):
    outs = [x]
    t = x
    for l in range(HG_LAYERS):
        t = F.relu(_conv(w, f"{p}.l{l}", t, pad=(1, 1)))
        outs.append(t)
    t = F.relu(_conv(w, p + ".agg", torch.cat(outs, dim=1)))
    gate = torch.sigmoid(_conv(w, p + ".ese", t.mean(dim=(2, 3), keepdim=True)))   # ESEModule
    t = t * gate
    return t + x if identity else t


def _hgnet(w, prefix, x, stages, det):
    for i in range(HG_STEM):
        x = F.relu(_conv(w, f"{prefix}.stem{i}", x, stride=(2, 2) if i == 0 else (1, 1), pad=(1, 1)))
    if det:
        x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for name, blocks, down, stride in stages:
        p = f"{prefix}.{name}"
        if down:
            x = _conv(w, p + ".ds", x, stride=stride, pad=(1, 1), groups=x.shape[1])
        for b in range(blocks):
            x = _hg_block(w, f"{p}.b{b}", x, identity=b > 0)
        feats.append(x)
    return feats


def _intracl(w, p, x):
    t = _conv(w, p + ".reduce", x)
    for k in (7, 5, 3):
        t = (_conv(w, f"{p}.c{k}", t, pad=(k // 2, k // 2)) + _conv(w, f"{p}.v{k}", t, pad=(k // 2, 0))
             + _conv(w, f"{p}.q{k}", t, pad=(0, k // 2)))
    return x + F.relu(_conv(w, p + ".ret", t))


@torch.no_grad()
def sdet_forward(w: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """x [N,3,H,W] (BGR, normalised; H, W multiples of 32) -> probability map [N,1,H,W] = 0.5 * (shrink + cbn)."""
    c2, c3, c4, c5 = _hgnet(w, "sdet", x, HG_STAGES_DET, det=True)
    up = lambda t, s: F.interpolate(t, scale_factor=s, mode="nearest")
    n = "sdet.neck"
    in5 = _conv(w, n + ".ins3", c5); in4 = _conv(w, n + ".ins2", c4)
    in3 = _conv(w, n + ".ins1", c3); in2 = _conv(w, n + ".ins0", c2)
    out4 = in4 + up(in5, 2); out3 = in3 + up(out4, 2); out2 = in2 + up(out3, 2)
    f5 = _conv(w, n + ".inp3", in5, pad=(4, 4)); f4 = _conv(w, n + ".inp2", out4, pad=(4, 4))
    f3 = _conv(w, n + ".inp1", out3, pad=(4, 4)); f2 = _conv(w, n + ".inp0", out2, pad=(4, 4))
    pan3 = f3 + _conv(w, n + ".panhead0", f2, stride=(2, 2), pad=(1, 1))
    pan4 = f4 + _conv(w, n + ".panhead1", pan3, stride=(2, 2), pad=(1, 1))
    pan5 = f5 + _conv(w, n + ".panhead2", pan4, stride=(2, 2), pad=(1, 1))
    p2 = _conv(w, n + ".panlat0", f2, pad=(4, 4)); p3 = _conv(w, n + ".panlat1", pan3, pad=(4, 4))
    p4 = _conv(w, n + ".panlat2", pan4, pad=(4, 4)); p5 = _conv(w, n + ".panlat3", pan5, pad=(4, 4))
    p5 = _intracl(w, n + ".incl4", p5); p4 = _intracl(w, n + ".incl3", p4)
    p3 = _intracl(w, n + ".incl2", p3); p2 = _intracl(w, n + ".incl1", p2)
    fuse = torch.cat([up(p5, 8), up(p4, 4), up(p3, 2), p2], dim=1)
    y = F.relu(_conv(w, "sdet.head.conv1", fuse, pad=(1, 1)))
    f = F.relu(F.conv_transpose2d(y, w["sdet.head.deconv1.w"], w["sdet.head.deconv1.b"], stride=2))
    base = torch.sigmoid(F.conv_transpose2d(f, w["sdet.head.deconv2.w"], w["sdet.head.deconv2.b"], stride=2))
    loc = F.relu(_conv(w, "sdet.head.local3", torch.cat([base, up(f, 2)], dim=1), pad=(1, 1)))   # LocalModule
    cbn = torch.sigmoid(_conv(w, "sdet.head.local1", loc))
    return 0.5 * (base + cbn)


@torch.no_grad()
def srec_forward(w: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """x [n,3,48,W] -> softmax probs [n,T,classes], T = W/8."""
    x = _hgnet(w, "srec", x, HG_STAGES_REC, det=False)[-1]
    x = F.avg_pool2d(x, (3, 2))
    return _svtr_ctc(w, "srec", x)

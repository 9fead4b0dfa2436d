"""Test-side ONNX writer: serialises a parameter-carrying graph in the op patterns Paddle2ONNX emits for the
PP-OCRv4 inference models (Conv [+ BatchNormalization], LearnableAffineBlock as Mul/Add with 1-element
parameters before and after the activation, Linear as MatMul + Add, LayerNorm as the LayerNormalization op or
decomposed, ConvTranspose [+ BN]; parameters as initializers or Constant nodes), plus distractor nodes that
carry constants but no parameters.  Only the protobuf fields the importer reads are written.  No onnx package
is needed (none is installed); this is the wire format of onnx.proto3.
"""
import struct

import numpy as np


def _varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _vi(field, v):
    return _varint((field << 3) | 0) + _varint(v)


def _f32(field, v):
    return _varint((field << 3) | 5) + struct.pack("<f", v)


def tensor(name, arr, raw=True):
    arr = np.ascontiguousarray(arr)
    out = b""
    if arr.dtype == np.float32:
        # packed and unpacked dims both occur in the wild
        out += _ld(1, b"".join(_varint(d) for d in arr.shape)) if raw else b"".join(_vi(1, d) for d in arr.shape)
        out += _vi(2, 1)
        out += _ld(9, arr.tobytes()) if raw else _ld(4, arr.tobytes())
    else:
        out += _ld(1, b"".join(_varint(d) for d in arr.shape)) + _vi(2, 7) + _ld(9, arr.astype(np.int64).tobytes())
    return out + _ld(8, name.encode())


def attr_ints(name, vals):
    return _ld(1, name.encode()) + _ld(8, b"".join(_varint(v) for v in vals)) + _vi(20, 7)


def attr_int(name, v):
    return _ld(1, name.encode()) + _vi(3, v) + _vi(20, 2)


def attr_float(name, v):
    return _ld(1, name.encode()) + _f32(2, v) + _vi(20, 1)


def attr_tensor(name, t):
    return _ld(1, name.encode()) + _ld(5, t) + _vi(20, 4)


class GraphWriter:
    def __init__(self, seed=0, constants_as_nodes=False):
        self.nodes, self.inits = [], []
        self.n = 0
        self.cur = "x"
        self.rng = np.random.default_rng(seed)
        self.const_nodes = constants_as_nodes

    def _name(self, p):
        self.n += 1
        return "%s_%d" % (p, self.n)

    def param(self, arr, raw=True):
        name = self._name("p")
        if self.const_nodes and arr.dtype == np.float32 and arr.size > 1:
            self.node("Constant", [], [name], [attr_tensor("value", tensor("", arr, raw))])
        else:
            self.inits.append(tensor(name, arr, raw))
        return name

    def node(self, op, ins, outs, attrs=()):
        b = b"".join(_ld(1, i.encode()) for i in ins) + b"".join(_ld(2, o.encode()) for o in outs)
        b += _ld(3, self._name(op).encode()) + _ld(4, op.encode()) + b"".join(_ld(5, a) for a in attrs)
        self.nodes.append(b)

    def op(self, op, extra=(), attrs=()):
        extra = list(extra)  # parameters first: a Constant node must precede its consumer
        out = self._name("t")
        self.node(op, [self.cur] + extra, [out], attrs)
        self.cur = out
        return out

    # ---- parameter-carrying patterns -------------------------------------------------------
    def conv(self, w, b=None, bn=False, pre_lab=False, transpose=False, bias_as_add=False, record=None):
        """Emits a conv whose folded form is (w, b).  bn / pre_lab un-fold it first.  record (a dict) receives the UN-FOLDED
        parameters exactly as written into the file: w, b (or None), bn = (scale, B, mean, var, eps) or None, lab = (a, c) or None."""
        w = w.astype(np.float64)
        cout = w.shape[1] if transpose else w.shape[0]
        b = np.zeros(cout) if b is None else b.astype(np.float64)
        shape = (1, cout, 1, 1) if transpose else (cout, 1, 1, 1)
        lab = None
        if pre_lab:  # y = a1 * conv(x) + c1
            a1, c1 = self.rng.uniform(0.8, 1.2), self.rng.normal() * 0.05
            w, b, lab = w / a1, (b - c1) / a1, (a1, c1)
        bnp = None
        if bn:
            s = self.rng.uniform(0.5, 1.5, cout); var = self.rng.uniform(0.5, 2.0, cout); mean = self.rng.normal(size=cout) * 0.1
            eps = 1e-5
            k = s / np.sqrt(var + eps)
            w, B = w / k.reshape(shape), b + mean * k
            b = np.zeros(cout)
            bnp = (s, B, mean, var, eps)
        ins = [self.param(w.astype(np.float32))]
        has_bias = bool(np.any(b != 0))
        if has_bias and not bias_as_add:
            ins.append(self.param(b.astype(np.float32)))
        self.op("ConvTranspose" if transpose else "Conv", ins, [attr_ints("strides", [1, 1]), attr_int("group", 1)])
        if has_bias and bias_as_add:
            self.op("Add", [self.param(b.astype(np.float32).reshape(1, cout, 1, 1))])
        if bnp:
            s, B, mean, var, eps = bnp
            self.op("BatchNormalization", [self.param(v.astype(np.float32)) for v in (s, B, mean, var)], [attr_float("epsilon", eps)])
        if lab:
            self.op("Mul", [self.param(np.array([lab[0]], np.float32))])
            self.op("Add", [self.param(np.array([lab[1]], np.float32))])
        if record is not None:
            record.update(w=w.astype(np.float32), b=b.astype(np.float32) if has_bias else None,
                          bn=tuple(np.asarray(v, np.float32) for v in bnp[:4]) + (bnp[4],) if bnp else None,
                          lab=(np.float32(lab[0]), np.float32(lab[1])) if lab else None, transpose=transpose)

    def lab(self, a, c):
        self.op("Mul", [self.param(np.asarray(a, np.float32).reshape(1))])
        self.op("Add", [self.param(np.asarray(c, np.float32).reshape(1))])

    def hardswish(self, decomposed=False):
        if not decomposed:
            self.op("HardSwish")
            return
        if decomposed == "hardsigmoid":   # x * HardSigmoid(alpha = 1/6, beta = 0.5)(x): the opset < 14 export of hard_swish
            x = self.cur
            self.op("HardSigmoid", attrs=[attr_float("alpha", 1.0 / 6.0), attr_float("beta", 0.5)])
            out = self._name("t"); self.node("Mul", [x, self.cur], [out]); self.cur = out
            return
        x = self.cur  # x * clip(x + 3, 0, 6) / 6 with scalar constants: must not be taken for a LAB
        self.op("Add", [self.param(np.array([3.0], np.float32))])
        self.op("Clip", [self.param(np.array([0.0], np.float32)), self.param(np.array([6.0], np.float32))])
        out = self._name("t"); self.node("Mul", [x, self.cur], [out]); self.cur = out
        self.op("Div", [self.param(np.array([6.0], np.float32))])

    def linear(self, w, b, gemm=False):
        if gemm:
            self.op("Gemm", [self.param(np.ascontiguousarray(w.T)), self.param(b)], [attr_int("transB", 1)])
        else:
            self.op("MatMul", [self.param(w)])
            self.op("Add", [self.param(b)])

    def layernorm(self, g, beta, decomposed=True):
        if not decomposed:
            self.op("LayerNormalization", [self.param(g), self.param(beta)], [attr_float("epsilon", 1e-5)])
            return
        self.op("ReduceMean"); self.op("Sub"); self.op("Pow", [self.param(np.array([2.0], np.float32))])
        self.op("ReduceMean"); self.op("Add", [self.param(np.array([1e-5], np.float32))]); self.op("Sqrt"); self.op("Div")
        self.op("Mul", [self.param(g)]); self.op("Add", [self.param(beta)])

    def distract(self):
        self.op("Reshape", [self.param(np.array([0, -1, 120], np.int64))])
        self.op("Transpose", attrs=[attr_ints("perm", [0, 2, 1, 3])])               # head split / merge around the attention
        self.op("Clip", attrs=[attr_float("min", 0.0), attr_float("max", 6.0)])     # opset < 11 form: limits as attributes
        self.op("Mul", [self.param(np.array([0.2581989], np.float32))])  # attention scale: scalar Mul without Add
        self.op("HardSigmoid", attrs=[attr_float("alpha", 0.2), attr_float("beta", 0.5)])

    def finish(self):
        graph = b"".join(_ld(1, n) for n in self.nodes) + _ld(2, b"g") + b"".join(_ld(5, t) for t in self.inits)
        return _vi(1, 8) + _ld(7, graph)


def build_model_onnx(manifest_text, tensors, seed=0, style=0, unfolded=None):
    """manifest_text: rt_model_manifest output; tensors: RTWB name -> array (retto_amd.synth).  style varies the
    patterns (0: BN un-folded + initializers + decomposed LayerNorm; 1: Constant nodes, LayerNormalization op,
    Gemm head, bias as a separate Add, hardswish as Add / Clip / Mul / Div; 2: like 0 with hardswish as x * HardSigmoid(x)).
    unfolded (a dict) receives, per conv base name, the parameters as they stand in the file (GraphWriter.conv)."""
    g = GraphWriter(seed, constants_as_nodes=(style == 1))
    names = [l.split()[0] for l in manifest_text.strip().split("\n")]
    i = 0
    while i < len(names):
        n = names[i]
        base, leaf = n.rsplit(".", 1)
        if leaf == "w":
            w = tensors[n]
            has_b = i + 1 < len(names) and names[i + 1] == base + ".b"
            b = tensors[base + ".b"] if has_b else None
            if w.ndim == 4:
                lc = base.endswith(".dw") or base.endswith(".pw")
                rec = {} if unfolded is not None else None
                g.conv(w, b, bn=(has_b and style != 1 and not lc), pre_lab=lc and base.split(".")[0] != "cls",
                       transpose="deconv" in base, bias_as_add=(style == 1 and has_b and not lc), record=rec)
                if unfolded is not None:
                    unfolded[base] = rec
                if lc and base + ".a" in tensors:
                    g.hardswish(decomposed={0: False, 1: True, 2: "hardsigmoid"}[style])
                elif not lc:
                    g.op("Relu")
            else:
                g.linear(w, b, gemm=(style == 1 and "head" in base))
                g.distract()
            i += 2 if has_b else 1
        elif leaf == "a":
            g.lab(tensors[n], tensors[base + ".c"]); i += 2
        elif leaf == "g":
            g.layernorm(tensors[n], tensors[base + ".beta"], decomposed=(style == 0)); i += 2
        else:
            raise AssertionError(n)
    return g.finish()

"""ONNX weight importer: host-only, runs without a GPU.  No PP-OCRv4 .onnx file exists
offline, so the importer is exercised on files written by tests/onnx_writer.py in the op patterns Paddle2ONNX
emits; the parameters are the seeded synthetic ones, un-folded (BatchNorm, pre-activation LAB) on the way out."""
import numpy as np
import pytest

import retto_amd
from retto_amd import synth

from onnx_writer import GraphWriter, build_model_onnx

KINDS = [(retto_amd.MODEL_DET, synth.det_tensors), (retto_amd.MODEL_CLS, synth.cls_tensors), (retto_amd.MODEL_REC, synth.rec_tensors)]


@pytest.mark.parametrize("which,make", KINDS)
def test_manifest_matches_synth(which, make):
    t = make()
    got = {}
    for line in retto_amd.model_manifest(which).strip().split("\n"):
        name, *dims = line.split()
        got[name] = tuple(int(d) for d in dims)
    assert set(got) == set(t)
    for name, dims in got.items():
        assert len(dims) == t[name].ndim
        assert all(d == -1 or d == s for d, s in zip(dims, t[name].shape)), name


@pytest.mark.parametrize("style", [0, 1])
@pytest.mark.parametrize("which,make", KINDS)
def test_import_round_trip(which, make, style):
    t = make()
    onnx = build_model_onnx(retto_amd.model_manifest(which), t, seed=5 + which, style=style)
    back = synth.unpack_blob(retto_amd.onnx_to_rtwb(which, onnx))
    assert set(back) == set(t)
    for name, ref in t.items():
        assert back[name].shape == ref.shape, name
        # BatchNorm / LAB folding happens in f32 on the way in: a few ulp
        np.testing.assert_allclose(back[name], ref, rtol=2e-6, atol=2e-7, err_msg=name)
    if style == 1:  # nothing to fold for the linears and LayerNorms: exact
        for name, ref in t.items():
            if ".neck.blk" in name or name.endswith((".g", ".beta", ".a", ".c")):
                assert np.array_equal(back[name], ref), name


def test_import_errors():
    with pytest.raises(retto_amd.RettoError):
        retto_amd.onnx_to_rtwb(retto_amd.MODEL_DET, b"\x08\x08")  # a model without a graph
    t = synth.cls_tensors()
    onnx = build_model_onnx(retto_amd.model_manifest(retto_amd.MODEL_CLS), t)
    with pytest.raises(retto_amd.RettoError):
        retto_amd.onnx_to_rtwb(retto_amd.MODEL_CLS, onnx[: len(onnx) // 2])  # truncated
    with pytest.raises(retto_amd.RettoError) as ei:
        retto_amd.onnx_to_rtwb(retto_amd.MODEL_DET, onnx)  # a cls file offered as det
    assert "det.stem.w" in str(ei.value)
    g = GraphWriter()
    g.conv(np.ones((8, 3, 3, 3), np.float32), np.ones(8, np.float32))
    with pytest.raises(retto_amd.RettoError) as ei:
        retto_amd.onnx_to_rtwb(retto_amd.MODEL_CLS, g.finish())  # stops at the first missing layer
    assert "cls.b0.expand.w" in str(ei.value)

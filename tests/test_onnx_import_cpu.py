"""ONNX weight importer: host-only, runs without a GPU.  No PP-OCRv4 .onnx file exists
offline, so the importer is exercised on files written by tests/onnx_writer.py in the op patterns Paddle2ONNX
emits; the parameters are the seeded synthetic ones, un-folded (BatchNorm, pre-activation LAB) on the way out."""
import numpy as np
import pytest

import retto_amd
from retto_amd import synth

from onnx_writer import GraphWriter, build_model_onnx

KINDS = [(retto_amd.MODEL_DET, synth.det_tensors), (retto_amd.MODEL_CLS, synth.cls_tensors), (retto_amd.MODEL_REC, synth.rec_tensors),
         (retto_amd.MODEL_SDET, synth.sdet_tensors), (retto_amd.MODEL_SREC, synth.srec_tensors)]   # + the PP-OCRv4 server graphs (config 5)


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


def test_server_files_come_through_the_det_and_rec_sources():
    """A host hands ch_PP-OCRv4_server_{det,rec}_infer.onnx to the same det / rec model sources as the mobile files
    (worker.rs:30-56): MODEL_DET / MODEL_REC recognise PPHGNet's 3 -> 64 stem and import against the server manifest."""
    for generic, server, make in ((retto_amd.MODEL_DET, retto_amd.MODEL_SDET, synth.sdet_tensors), (retto_amd.MODEL_REC, retto_amd.MODEL_SREC, synth.srec_tensors)):
        t = make()
        onnx = build_model_onnx(retto_amd.model_manifest(server), t, seed=3, style=0)
        back = synth.unpack_blob(retto_amd.onnx_to_rtwb(generic, onnx))
        assert set(back) == set(t) and all(k.startswith(("sdet.", "srec.")) for k in back)
        for name, ref in t.items():
            np.testing.assert_allclose(back[name], ref, rtol=2e-6, atol=2e-7, err_msg=name)


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


def test_real_paddle2onnx_files_when_present():
    """Opt-in: RETTO_REAL_MODELS=<dir with ch_PP-OCRv4_det_infer.onnx, ch_PP-OCRv4_rec_infer.onnx,
    ch_ppocr_mobile_v2.0_cls_infer.onnx, ppocr_keys_v1.txt> (the files retto-core/build.rs:7-12 downloads).  None exists offline,
    so this is skipped here; with the files it checks that the importer's event matcher accepts the real graphs (manifest match)
    and -- when onnxruntime is importable and a GPU is present -- one forward pass per network against onnxruntime-CPU."""
    import os
    import pytest
    d = os.environ.get("RETTO_REAL_MODELS")
    if not d:
        pytest.skip("RETTO_REAL_MODELS not set (the real PP-OCRv4 .onnx files are not available offline)")
    import retto_amd
    files = {retto_amd.MODEL_DET: "ch_PP-OCRv4_det_infer.onnx", retto_amd.MODEL_CLS: "ch_ppocr_mobile_v2.0_cls_infer.onnx",
             retto_amd.MODEL_REC: "ch_PP-OCRv4_rec_infer.onnx"}
    blobs = {}
    for kind, name in files.items():
        data = open(os.path.join(d, name), "rb").read()
        blobs[kind] = retto_amd.onnx_to_rtwb(kind, data)       # raises with the offending node / tensor on any mismatch
        assert blobs[kind][:4] == b"RTWB"
    keys = open(os.path.join(d, "ppocr_keys_v1.txt"), "rb").read()
    from oracle.pipeline import load_dictionary
    assert len(load_dictionary(keys)) == 6625
    try:
        import onnxruntime as ort
    except ImportError:
        pytest.skip("onnxruntime not installed: manifest match checked, forward comparison skipped")
    S = retto_amd.RettoWorkerModelSource
    cfg = retto_amd.RettoSessionConfig()
    cfg.worker_config = retto_amd.RettoHipWorkerConfig(device=0, models=retto_amd.RettoWorkerModelProvider(
        det=S.Path(os.path.join(d, files[retto_amd.MODEL_DET])), rec=S.Path(os.path.join(d, files[retto_amd.MODEL_REC])),
        cls=S.Path(os.path.join(d, files[retto_amd.MODEL_CLS]))))
    cfg.rec_processor_config.character_source = S.Path(os.path.join(d, "ppocr_keys_v1.txt"))
    try:
        sess = retto_amd.RettoSession(cfg)
    except retto_amd.BackendError:
        pytest.skip("no GPU: manifest match checked, forward comparison skipped")
    try:
        rng = np.random.default_rng(0)
        for kind, shape, fn, tol in ((retto_amd.MODEL_DET, (1, 3, 320, 480), sess.worker.det, 1e-4),
                                     (retto_amd.MODEL_CLS, (2, 3, 48, 192), sess.worker.cls, 1e-4),
                                     (retto_amd.MODEL_REC, (2, 3, 48, 320), sess.worker.rec, 2e-4)):
            x = rng.uniform(-1, 1, shape).astype(np.float32)
            o = ort.InferenceSession(os.path.join(d, files[kind]), providers=["CPUExecutionProvider"])
            ref = o.run(None, {o.get_inputs()[0].name: x})[0]
            assert np.abs(fn(x) - ref).max() <= tol
    finally:
        sess.close()

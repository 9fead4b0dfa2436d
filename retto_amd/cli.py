"""Directory batch driver: the retto-cli loop (/root/reference/retto-cli/src/main.rs:41-95) over the
HIP session -- walk a directory, decode every file on the host (like the reference's
ImageHelper::new_from_raw_img_flow), run the pages through RettoSession in batches, log the
average time per image.  Model files are RTWB blobs (retto_amd/synth.py; converting the PP-OCRv4
.onnx files is SURVEY 8(f) rank 1); ``--synthetic`` uses seeded synthetic weights.

    python -m retto_amd.cli --images DIR [--batch 32] [--device-id 0] [--json OUT.jsonl]
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import time

import numpy as np

log = logging.getLogger("retto_cli")


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="retto-cli (hip)")
    ap.add_argument("--det-model-path", default="ch_PP-OCRv4_det_infer.rtwb")
    ap.add_argument("--cls-model-path", default="ch_ppocr_mobile_v2.0_cls_infer.rtwb")
    ap.add_argument("--rec-model-path", default="ch_PP-OCRv4_rec_infer.rtwb")
    ap.add_argument("--rec-keys-path", default="ppocr_keys_v1.txt")
    ap.add_argument("-i", "--images", required=True)
    ap.add_argument("--device", choices=["hip"], default="hip", help="only the MI355X HIP backend exists here")
    ap.add_argument("--device-id", type=int, default=0)
    ap.add_argument("--batch", type=int, default=32, help="pages per rt_run_batch call")
    ap.add_argument("--synthetic", action="store_true", help="seeded synthetic weights + dictionary instead of model files")
    ap.add_argument("--json", default=None, help="write one JSON object per image (det/cls/rec stage results)")
    return ap


def walk_files(root: str):
    """WalkDir semantics of main.rs:72-77: every regular file under root, directory order."""
    out = []
    for d, dirs, files in os.walk(root):
        dirs.sort()
        for f in sorted(files):
            out.append(os.path.join(d, f))
    return out


def main(argv=None) -> int:
    a = build_parser().parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(levelname)s %(name)s: %(message)s")
    import retto_amd
    if a.synthetic:
        cfg = retto_amd.synthetic_session_config(0, device=a.device_id)
    else:
        cfg = retto_amd.RettoSessionConfig()
        S = retto_amd.RettoWorkerModelSource
        cfg.worker_config = retto_amd.RettoHipWorkerConfig(device=a.device_id, models=retto_amd.RettoWorkerModelProvider(
            det=S.Path(a.det_model_path), rec=S.Path(a.rec_model_path), cls=S.Path(a.cls_model_path)))
        cfg.rec_processor_config.character_source = S.Path(a.rec_keys_path)
    session = retto_amd.RettoSession(cfg)
    files = walk_files(a.images)
    log.info("Found %d files, processing...", len(files))
    out = open(a.json, "w") if a.json else None
    start = time.perf_counter()
    n = 0
    for s0 in range(0, len(files), a.batch):
        chunk = files[s0:s0 + a.batch]
        pages = []
        for path in chunk:
            with open(path, "rb") as fh:
                pages.append(retto_amd.decode_image(fh.read()))  # a decode failure aborts, like expect() in main.rs:83
        results = session.run_batch(pages)
        n += len(results)
        if out:
            for path, r in zip(chunk, results):
                out.write(json.dumps({
                    "file": path,
                    "det": [{"boxes": {"inner": [{"x": p.x, "y": p.y} for p in d.boxes.inner]}, "score": d.score} for d in r.det_result],
                    "cls": [{"label": {"label": c.label.label, "score": c.label.score}} for c in r.cls_result],
                    "rec": [{"text": t.text, "score": None if np.isnan(t.score) else t.score} for t in r.rec_result],
                }, ensure_ascii=False) + "\n")
    dur = time.perf_counter() - start
    if out:
        out.close()
    if n:
        log.info("Successfully processed %d images, avg time: %.2fms", n, 1000.0 * dur / n)
    session.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())

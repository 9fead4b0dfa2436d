"""HBM bytes of ONE pass of the DBNet det network, from rocprofv3 PMC counters (north_star: "rocprof-reported HBM GB/s").

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/det_fetch -o f -- python3 tools/layer_profile.py 32 1 0
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/det_write -o w -- python3 tools/layer_profile.py 32 1 0
    python tools/pmc_det_bytes.py gpurun_out/det_fetch/f_counter_collection.csv gpurun_out/det_write/w_counter_collection.csv profiles/pmc_det_c3.json

`tools/layer_profile.py 32 1 0` runs the pipeline on 32 pages of 960 x 960 without text lines: det pre-process, DBNet, DB post.
Every launch of the process is summed EXCEPT the kernels that are not the det network (DB post-processing, the page / map
preparation, the checksum): what is left are the launches between `net/det`'s two events.  The process runs the network
`passes` times (warm-up calls included; counted from the stem launches); bytes per pass = sum / passes.
Units / corrections as tools/pmc_summary.py (MI355X_MICROARCH.md, HBM section): KiB -> bytes, FETCH_SIZE x 2 on gfx950
(128-byte requests tallied at 64 bytes), WRITE_SIZE as is."""
import collections
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# not part of the det network: DB post (dbpost_kernels.hip), pre / post stages (prepost_kernels.hip), the map checksum
NOT_DET = re.compile(r"k_ccl_|k_contour_|k_row_extents|k_sort_boxes|k_pack_boxes|k_thumbnail|k_det_normalize|k_warp_crops|k_resize_norm|"
                     r"k_cls_post|k_ctc_|k_sum_partial|k_where_am_i|__amd_rocclr")   # (rocclr: the runtime's own fill / copy kernels)


def short(name):
    m = re.match(r"(?:void )?(?:rt::\w+::)?([\w]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]][0] += 1
        d[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return d


def main():
    from retto_amd import _lib
    f, w, out = agg(sys.argv[1]), agg(sys.argv[2]), sys.argv[3]
    pages = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    stem = [k for k in f if "k_stem" in k]
    passes = sum(f[k][0] for k in stem)
    if not passes:
        raise SystemExit("no stem launch found: not a det run")
    rows, skipped = [], []
    for k, (n, fv) in f.items():
        if NOT_DET.search(k):
            skipped.append(short(k))
            continue
        wn, wv = w.get(k, [0, 0.0])
        rows.append((short(k), n, fv * 1024 * 2 / passes, wv * 1024 / passes))
    rows.sort(key=lambda r: -(r[2] + r[3]))
    fetch = sum(r[2] for r in rows); write = sum(r[3] for r in rows)
    res = {"command": "tools/layer_profile.py %d 1 0 under separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes" % pages,
           "workload": {"pages": pages, "size": 960, "dtype": "f32", "models": "mobile"},
           "csrc_digest": _lib.source_digest(), "passes": passes,
           "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 128-byte requests tallied at 64 bytes); WRITE_SIZE as is",
           "det_fetch_bytes_per_pass": int(fetch), "det_write_bytes_per_pass": int(write), "det_hbm_bytes_per_pass": int(fetch + write),
           "left_out": sorted(set(skipped)),
           "kernels": [{"kernel": k, "launches_per_pass": round(n / passes, 2), "fetch_bytes_per_pass": int(a), "write_bytes_per_pass": int(b)} for k, n, a, b in rows]}
    json.dump(res, open(out, "w"), indent=1)
    print("det network, %d pages, per pass (%d passes): fetch %.3f GB + write %.3f GB = %.3f GB" % (pages, passes, fetch / 1e9, write / 1e9, (fetch + write) / 1e9))
    for k, n, a, b in rows[:25]:
        print("  %-46s x%5.1f  fetch %8.3f MB  write %8.3f MB" % (k[:46], n / passes, a / 1e6, b / 1e6))
    print("left out:", ", ".join(sorted(set(skipped))))


if __name__ == "__main__":
    main()

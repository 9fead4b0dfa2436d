"""Folds two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass on gfx950) into
per-kernel HBM bytes per launch.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --lanes 1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --lanes 1
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq -o s -- python3 bench.py ... (same flags)
    python tools/pmc_summary.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv profiles/pmc_traffic.json [gpurun_out/pmc_sq/s_counter_collection.csv]

Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of 16-byte-per-lane coalesced reads (128-byte requests tallied at 64 bytes), so
it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  Every load/store in these kernels is a
16-byte vector access.

Optional SQ pass: mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) -- the busy counter
is summed over the SIMDs (it equals 32 cycles x the number of v_mfma_f32_16x16x4_f32 issued), GRBM_GUI_ACTIVE is
summed over the 8 XCDs; clock_ghz = GRBM_GUI_ACTIVE / 8 / kernel time.  SQ_WAVE_CYCLES / WAIT_* are quad-cycles.
"""
import collections
import csv
import json
import re
import sys


def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]][0] += 1
        d[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return d


def short(name):
    m = re.match(r"(?:void )?(?:rt::\w+::)?([\w]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def agg_multi(path):
    d = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
    for r in csv.DictReader(open(path)):
        e = d[r["Kernel_Name"]][r["Counter_Name"]]
        e[0] += 1
        e[1] += float(r["Counter_Value"])
        e[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return d


# profiler-label -> kernel symbol of the families bench.py's roofline can name
LABELS = {"gemm_pw/k_gemm32p": "family:gemm32p", "gemm_pw/k_gemm32p+se": "family:gemm32p_se",
          "gemm_pw/k_gemm_wide<4,5,4,3>": "k_gemm_wide<4, 5, 4, 3, 0, 0, 0, 0, 0>",
          "gemm_pw/k_gemm_wide<2,5,4,3>+se": "k_gemm_wide<2, 5, 4, 3, 0, 0, 1, 0, 0>",
          "gemm_pw/k_gemm_wide<2,4,4,2>": "k_gemm_wide<2, 4, 4, 2, 0, 0, 0, 0, 0>",
          "gemm_pw/k_gemm_split": "family:gemm_split", "gemm_pw/k_gemm_split+se": "family:gemm_split_se",
          "conv16_3x3": "family:conv16_3x3", "gemm16": "family:gemm16", "conv16_9x9": "family:conv16_9x9"}
# fp16 families that several template instances serve: their "family:<name>" entry is the launch-weighted mean over the
# member kernels (traffic per launch) and the cycle-weighted MFMA utilisation.  (The 3x3 family's few register-staged
# launches -- the 3-channel stems -- run k_conv16 instances shared with other families and are left out.)
FAMILIES = {"gemm32p": r"k_gemm32p<-?\d+, -?\d+, (true|false), 0, false>", "gemm32p_se": r"k_gemm32p<-?\d+, -?\d+, (true|false), 0, true>",   # (fp32: the persistent LDS-DMA wide GEMM, K = 32 j and K = 32 j + 16 instances)
            "gemm_split": r"k_gemm_split<-?\d+, -?\d+, 0, false>", "gemm_split_se": r"k_gemm_split<-?\d+, -?\d+, 0, true>",   # (round 6, opt-in leg: split-bf16 form of the same layers)
            "conv16_3x3": r"k_conv16v2<\d, (3, 3|9, 3)(, 0)?(, \d)?(, (true|false))?>", "gemm16": r"k_gemm16p?<", "conv16_9x9": r"k_conv16v2<2, 9, 9(, 0)?(, \d)?(, (true|false))?>"}


def main():
    """pmc_summary.py fetch.csv write.csv out.json [sq.csv] [workload-json]   (workload-json: the dict bench.py compares with its
    own flags, e.g. '{"workload": "c3", "pages": 32, "size": 960, "lines": 32, "dtype": "f32", "models": "mobile"}')"""
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from retto_amd import _lib
    f, w, out = agg(sys.argv[1]), agg(sys.argv[2]), sys.argv[3]
    sq = agg_multi(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] else {}
    wl = json.loads(sys.argv[5]) if len(sys.argv) > 5 else {"workload": "c3", "pages": 32, "size": 960, "lines": 32, "dtype": "f32", "models": "mobile"}
    res = {}
    for k, (n, fv) in f.items():
        wn, wv = w.get(k, [0, 0.0])
        res[short(k)] = {"launches": n, "fetch_bytes_per_launch": int(fv / n * 1024 * 2),
                         "write_bytes_per_launch": int(wv / max(wn, 1) * 1024)}
    for k, ctrs in sq.items():
        if short(k) not in res or "GRBM_GUI_ACTIVE" not in ctrs:
            continue
        avg = {c: v[1] / v[0] for c, v in ctrs.items()}
        ns = ctrs["GRBM_GUI_ACTIVE"][2] / ctrs["GRBM_GUI_ACTIVE"][0]
        xcd_cycles = avg["GRBM_GUI_ACTIVE"] / 8.0
        res[short(k)]["sq"] = {
            "mfma_util": round(avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * xcd_cycles), 4),
            "clock_ghz": round(xcd_cycles / ns, 3),
            "wave_wait_any_frac": round(avg.get("SQ_WAIT_ANY", 0.0) / max(avg.get("SQ_WAVE_CYCLES", 1.0), 1.0), 3),
            "wave_wait_inst_frac": round(avg.get("SQ_WAIT_INST_ANY", 0.0) / max(avg.get("SQ_WAVE_CYCLES", 1.0), 1.0), 3),
            "wave_active_inst_frac": round(avg.get("SQ_ACTIVE_INST_ANY", 0.0) / max(avg.get("SQ_WAVE_CYCLES", 1.0), 1.0), 3)}
    for fam, pat in FAMILIES.items():
        mem = {k: v for k, v in res.items() if re.match(pat, k)}
        n = sum(v["launches"] for v in mem.values())
        if not n:
            continue
        e = {"launches": n, "members": sorted(mem),
             "fetch_bytes_per_launch": int(sum(v["fetch_bytes_per_launch"] * v["launches"] for v in mem.values()) / n),
             "write_bytes_per_launch": int(sum(v["write_bytes_per_launch"] * v["launches"] for v in mem.values()) / n)}
        sqm = [(k, v["sq"]) for k, v in mem.items() if "sq" in v]
        if sqm:
            # weights: each member's share of the family's GPU cycles (launches x cycles per launch of the SQ pass)
            cyc = {}
            for k0, ctrs in sq.items():
                if short(k0) in mem and "GRBM_GUI_ACTIVE" in ctrs:
                    cyc[short(k0)] = ctrs["GRBM_GUI_ACTIVE"][1]
            tot = sum(cyc.get(k, 0.0) for k, _ in sqm) or 1.0
            e["sq"] = {"mfma_util": round(sum(q["mfma_util"] * cyc.get(k, 0.0) for k, q in sqm) / tot, 4),
                       "clock_ghz": round(sum(q["clock_ghz"] * cyc.get(k, 0.0) for k, q in sqm) / tot, 3)}
        res["family:" + fam] = e
    top = dict(sorted(res.items(), key=lambda kv: -(kv[1]["fetch_bytes_per_launch"] + kv[1]["write_bytes_per_launch"]) * kv[1]["launches"])[:40])
    json.dump({"command": "bench.py --steps 2 --warmup 2 --no-cpu-baseline --lanes 1 (+ the workload flags below), separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes",
               "workload": wl, "csrc_digest": _lib.source_digest(), "labels": LABELS,
               "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 128-byte requests tallied at 64 bytes); WRITE_SIZE as is",
               "kernels": top}, open(out, "w"), indent=1)
    for k, v in list(top.items())[:12]:
        print("%-44s n=%4d fetch %.3f GB write %.3f GB %s" % (k[:44], v["launches"], v["fetch_bytes_per_launch"] / 1e9,
                                                             v["write_bytes_per_launch"] / 1e9, v.get("sq", "")))


if __name__ == "__main__":
    main()

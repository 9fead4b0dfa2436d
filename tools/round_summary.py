"""Prints the figures DESIGN.md section 5 quotes from one collection (tools/collect_profiles.sh <tag> -> gpurun_out/<tag>/ or
profiles/<tag>_*): python tools/round_summary.py gpurun_out/round6_e/ [prefix]"""
import csv, json, sys

d0 = sys.argv[1]; pre = sys.argv[2] if len(sys.argv) > 2 else ""
def J(f): return json.loads(open(d0 + pre + f).read().strip().splitlines()[-1])
d = J("bench_default.json")
print("value", d["value"], d["ms_per_step"], d["repeat"]["ms_per_step"])
r = d["roofline"]; print("roofline frac", r["frac"], "ach", r["achieved"], "ms/launch", r["avg_launch_ms"], "traffic", r["traffic"], "stale", r["traffic_source"].get("stale"))
print("sync", d["synchronous_calls"]["value"], d["synchronous_calls"]["ms_per_step"])
e = d["roofline_e2e"]; print("e2e tflops", e["tflops"], e["frac_of_mfma_peak"], "gbs", e["gbs"], e["frac_of_hbm_peak"])
print("host pages", d["pages_on_host"]["value"], round(d["pages_on_host"]["value"] / d["value"], 4), "cores", d["host"]["cpu_cores_busy"])
n = d["networks"]; print("networks det", n["det_ms"], "cls", n["cls_ms"], "rec", n["rec_ms"], "rec_tflops", n["rec_tflops"], "det b_layer gbs", n.get("det_b_layer_gbs"), n.get("det_b_layer_frac_of_hbm_peak"), "pmc", {k: v for k, v in n.get("det_pmc", {}).items() if k in ("stale", "gbs", "frac_of_hbm_peak", "hbm_bytes_per_page")})
s = d["split_bf16"]; print("split", s["value"], s["ms_per_step"], "rec", s.get("rec_net_ms"), s["kernels"]["layers_moved_ms_per_step"], s["kernels"]["split_fp32_equivalent_tflops"])
c = d["c5"]; print("c5", c["value"], c["ms_per_step"], "frac", c["roofline"]["frac"], c["roofline"]["achieved"])
print("cpu", d["cpu_baseline"]["value"])
for f in ("bench_c2.json", "bench_c4.json"):
    x = J(f); print(f, x["value"], x["ms_per_step"], "cores", x["host"]["cpu_cores_busy"], "sync ms", x["synchronous_calls"]["ms_per_step"])
for f in ("c3_layers.txt", "c3_layers_split.txt", "c5_layers.txt"):
    print(f, open(d0 + pre + f).readline().strip())
rows = list(csv.DictReader(open(d0 + pre + "c3_kernel_stats.csv")))
for r_ in rows:
    if any(k in r_["Name"] for k in ("k_gemm32p", "k_gemm_split", "attention_mfma", "k_dwconv_sweep<5")): print("  ", r_["Name"][:64], r_["Calls"], round(float(r_["AverageNs"]) / 1e3, 1))
for f in ("pmc_c3.txt",):
    for l in open(d0 + pre + f):
        if l.startswith("family:gemm32p") or l.startswith("family:gemm_split"): print("  ", l.strip()[:150])
print(open(d0 + pre + "pmc_det.txt").readline().strip())
for l in open(d0 + pre + "gemm_split.txt"):
    if l.startswith("{"):
        x = json.loads(l); print("  gemm", x["M"], x["K"], x["N"], "v", x["variant"], "ms", x["ms"], "max", "%.3g" % x["max_abs_err_vs_fp64"], "rms", "%.3g" % x["rms_err_vs_fp64"])
print("digest", json.load(open(d0 + ("pmc_traffic_c3.json")))["csrc_digest"])

"""Merge the per-counter-set outputs of tools/pmc_bench.sh (gpurun_out/pmc_<tag><COUNTER>.json) into one summary in the layout of
profiles/round1_pmc_summary.json.  usage: python tools/pmc_summary.py <tag> <out.json> "<source note>" """
import json, os, sys
tag, out, note = sys.argv[1], sys.argv[2], sys.argv[3]
R = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
kern = {}
for first in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "SQ_VALU_MFMA_BUSY_CYCLES"):
    d = json.load(open(os.path.join(R, f"pmc_{tag}{first}.json")))
    for k, cs in d.items():
        for c, v in cs.items():
            mean = v["mean"] * (2.0 if c == "FETCH_SIZE" else 1.0)          # gfx950: 128-B requests counted as 64 B (MI355X_MICROARCH.md, HBM section)
            kern.setdefault(k, {})[c] = {"launches": v["launches"], "mean_per_launch": round(mean, 1)}
for k, m in kern.items():
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        m["hbm_side_bytes_per_launch"] = int((m["FETCH_SIZE"]["mean_per_launch"] + m["WRITE_SIZE"]["mean_per_launch"]) * 1024)
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
        h, ms = m["TCC_HIT_sum"]["mean_per_launch"], m["TCC_MISS_sum"]["mean_per_launch"]
        m["l2_hit_rate"] = round(h / max(h + ms, 1.0), 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        m["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / (128.0 * m["GRBM_GUI_ACTIVE"]["mean_per_launch"]), 4)
json.dump({"source": note,
           "corrections": "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B for 16 B/lane coalesced reads, MI355X_MICROARCH.md section HBM); "
                          "WRITE_SIZE taken as reported (uncalibrated); values are KiB per launch; Infinity-Cache hits are included in FETCH_SIZE",
           "kernels": kern}, open(out, "w"), indent=1)
print(json.dumps({k: {c: v for c, v in m.items() if not isinstance(v, dict)} for k, m in kern.items()}))

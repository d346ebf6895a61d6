"""Summarise rocprofv3 --pmc passes per kernel: python tools/summarize_pmc.py OUT.json DIR [DIR ...]

Each DIR is the -d directory of one `rocprofv3 --pmc <COUNTERS> --output-format csv` run of the same command.  Per kernel and
counter: calls, total, mean, min and max over the dispatches.  FETCH_SIZE and WRITE_SIZE are reported by the tool in KB
(MI355X_MICROARCH.md, HBM section; `mean_KB` / `max_KB` / `total_KB` are kept as aliases); the 2x correction for
16-byte-per-lane streaming reads is applied by the reader (bench.py), not here.  SQ_* cycle counters count quad-cycles."""
import csv, glob, json, os, sys
out, dirs = sys.argv[1], sys.argv[2:]
acc = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            c = row["Counter_Name"]
            v = float(row["Counter_Value"])
            e = acc.setdefault(k, {}).setdefault(c, {"calls": 0, "total": 0.0, "max": 0.0, "min": float("inf")})
            e["calls"] += 1
            e["total"] += v
            e["max"] = max(e["max"], v)
            e["min"] = min(e["min"], v)
for k in acc:
    for c, e in acc[k].items():
        e["mean"] = e["total"] / max(e["calls"], 1)
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            e["total_KB"], e["max_KB"], e["mean_KB"] = e["total"], e["max"], e["mean"]
    m = acc[k]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m and m["SQ_BUSY_CYCLES"]["total"] > 0:
        # MFMA busy cycles are summed over the SIMDs that ran the kernel, SQ_BUSY_CYCLES over the shader engines: the ratio is
        # only comparable between kernels; the absolute utilisation is MOPS-derived flops / (time x peak) in DESIGN.md
        m["derived"] = {"mfma_busy_over_sq_busy": m["SQ_VALU_MFMA_BUSY_CYCLES"]["total"] / m["SQ_BUSY_CYCLES"]["total"]}
json.dump(acc, open(out, "w"), indent=1, sort_keys=True)
print("kernels:", len(acc))

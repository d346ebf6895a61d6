"""Summarise rocprofv3 --pmc passes per kernel: python tools/summarize_pmc.py OUT.json DIR [DIR ...]

Each DIR is the -d directory of one `rocprofv3 --pmc <COUNTER> --output-format csv` run of the same command.  FETCH_SIZE and
WRITE_SIZE are reported by the tool in KB (MI355X_MICROARCH.md, HBM section); the 2x correction for 16-byte-per-lane
streaming reads is applied by the reader (bench.py), not here."""
import csv, glob, json, os, sys
out, dirs = sys.argv[1], sys.argv[2:]
acc = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            c = row["Counter_Name"]
            e = acc.setdefault(k, {}).setdefault(c, {"calls": 0, "total_KB": 0.0, "max_KB": 0.0})
            e["calls"] += 1
            e["total_KB"] += float(row["Counter_Value"])
            e["max_KB"] = max(e["max_KB"], float(row["Counter_Value"]))
for k in acc:
    for c in acc[k]:
        acc[k][c]["mean_KB"] = acc[k][c]["total_KB"] / max(acc[k][c]["calls"], 1)
json.dump(acc, open(out, "w"), indent=1, sort_keys=True)
print("kernels:", len(acc))

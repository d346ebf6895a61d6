"""Timeline of one N = 10000 factorisation from a rocprofv3 kernel trace: per panel step, when the panel kernels
(potrf / trsm) and the trailing updates ran, on which stream, and how much of the step the update covers.
python tools/chol_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if any(k in n for k in ("potrf", "trsm", "syrk", "assemble", "kernel_matrix")):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("alabi::", "").replace("void ", ""), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
# the last factorisation: from the last kernel-matrix assembly on
starts = [i for i, r in enumerate(rows) if "assemble" in r[2] or "kernel_matrix" in r[2]]
i0 = starts[-1] if starts else 0
rows = rows[i0:]
t0 = rows[0][0]
tot = {}
for s, e, n, q in rows:
    tot.setdefault(n, [0, 0.0]); tot[n][0] += 1; tot[n][1] += (e - s) / 1e3
print(f"factorisation span {(rows[-1][1] - t0) / 1e6:.3f} ms, {len(rows)} kernels")
for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:40s} {c:5d} launches {us / 1e3:8.3f} ms total {us / c:8.1f} us mean")
# union of busy time of the update kernels vs the panel kernels
def union(iv):
    iv = sorted(iv); t = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: t += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return t + ce - cs
upd = [(s, e) for s, e, n, q in rows if "syrk" in n]
pan = [(s, e) for s, e, n, q in rows if "syrk" not in n]
print(f"update kernels busy (union) {union(upd) / 1e6:.3f} ms; panel kernels busy (union) {union(pan) / 1e6:.3f} ms; any kernel busy {union(upd + pan) / 1e6:.3f} ms")
print("first 40 kernels: start_us dur_us stream name")
for s, e, n, q in rows[:40]:
    print(f"  {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {q:>4s} {n}")

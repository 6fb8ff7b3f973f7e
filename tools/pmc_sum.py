"""Diagnostic: per-kernel mean of each counter in a rocprofv3 --pmc counter_collection.csv (values shown in MB when the counter
is in KB, as FETCH_SIZE / WRITE_SIZE are).
    python tools/pmc_sum.py <counter_collection.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
per = collections.defaultdict(list)
for r in rows:
    per[(r["Kernel_Name"].split("(")[0][:90], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:22]:
    print(f"{k:90s} {c:11s} n={len(v):5d} mean={sum(v)/len(v)/1024:10.1f} MB(KB-units)")

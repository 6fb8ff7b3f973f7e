"""Diagnostic helper: builds profiles/tcn_pointwise_pmc.json (what bench.py's roofline.traffic reads) from the FETCH_SIZE /
WRITE_SIZE passes that tools/refresh_profiles.sh collected under gpurun_out/<prefix>/pmc_*.
    python tools/pmc_json.py r2 <lib_version>"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix, lib_version = sys.argv[1], int(sys.argv[2])
base = os.path.join(ROOT, "gpurun_out", prefix)


def mean_counter(d, counter, kernel_substr):
    rows = list(csv.DictReader(open(os.path.join(base, d, "a_counter_collection.csv"))))
    per = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]:
            per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    name = max(per, key=lambda k: len(per[k]))          # the instantiation with most launches
    v = per[name]
    return name, sum(v) / len(v), len(v)


out = {}
for key, sub, fdir, wdir, cmd, note in (
        ("fp32_b16_t32000", "gemm_kernel<msf2::ProjNT<0, 5>", "pmc_fp32_fetch", "pmc_fp32_write",
         "python3 tools/probe_op.py tcn_pointwise --reps 5",
         "gfx950 FETCH_SIZE tallies 128-B requests at 64 B: bytes read = 2 x FETCH_SIZE; WRITE_SIZE exact"),
        ("bf16_b64_t32000", "proj_astat_bf16_kernel<0, 64", "pmc_bf16_fetch", "pmc_bf16_write",
         "python3 tools/time_tcn16.py",
         "gfx950 FETCH_SIZE doubled as for wide streaming reads (activation rows as 32-byte segments that add up to whole lines, weights as whole lines)")):
    kn, f, nf = mean_counter(fdir, "FETCH_SIZE", sub)
    _, w, nw = mean_counter(wdir, "WRITE_SIZE", sub)
    out[key] = {"kernel": kn.split("(")[0].replace("void ", "").replace("msf2::", ""),
                "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- " + cmd,
                "FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w, "correction": note,
                "traffic_bytes_per_launch": (2 * f + w) * 1024.0, "lib_version": lib_version,
                "launches_fetch": nf, "launches_write": nw, "round": prefix}
json.dump(out, open(os.path.join(ROOT, "profiles", "tcn_pointwise_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

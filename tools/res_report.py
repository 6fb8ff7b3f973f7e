"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output: one line per kernel (diagnostic helper)."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
filt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
rows = {}
for line in txt.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    body = m.group(1).strip()
    if body.startswith("Function Name:") or body.startswith("Name:"):
        cur = body.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in body:
        k, v = body.split(":", 1)
        rows[cur][k.strip()] = v.strip()
names = list(rows)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for n, d in zip(names, dem):
    if filt and filt not in d:
        continue
    r = rows[n]
    print(f"{d[:90]:90s} V {r.get('VGPRs','?'):>4} A {r.get('AGPRs','?'):>4} spill {r.get('VGPRs Spill','?'):>3} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?')} lds {r.get('LDS Size [bytes/block]','?')}")

#!/bin/bash
# Diagnostic helper: bf16 bench (C3) with several builds of the library on ONE GPU box, so the numbers compare.
# usage: tools/ab_libs.sh <lib path relative to repo | base> ...
cd /root/repo
cmd=""
for l in "$@"; do
  if [ "$l" = base ]; then e=""; else e="MSF2_LIB=/root/repo/$l"; fi
  cmd="$cmd echo LIB $l && $e timeout -k 10 150 python bench.py --precision bf16 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d[\"value\"],1), round(d[\"ms_per_step\"],2), round(d[\"roofline\"][\"launch_ms\"]*1000,1))' &&"
done
cmd="$cmd true"
timeout 1700 /usr/local/graft/bin/gpurun --timeout 900 -- "$cmd" 2>&1 | grep -v "every call\|^\[gpurun\] sending"

#!/bin/bash
# Diagnostic: per-kernel times (rocprofv3 --kernel-trace --stats) of the bf16 batch-64 forward under several knob settings, on ONE box.
# usage (on the GPU box, from the repo root): tools/prof_knobs.sh <tag>:<knobs> ...     e.g.  base:- sa0:store_aware=0
export TMPDIR=/tmp
mkdir -p gpurun_out
for spec in "$@"; do
  tag="${spec%%:*}"; knobs="${spec#*:}"
  rm -rf /tmp/prof_$tag
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python3 tools/knob_forward.py "$knobs" ${PROF_BATCH:-64} ${PROF_PREC:-bf16} > gpurun_out/knob_$tag.log 2>&1 || { echo "run $tag failed"; tail -5 gpurun_out/knob_$tag.log; exit 1; }
  f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
  cp "$f" gpurun_out/knob_${tag}_kernel_stats.csv
  echo "== $tag ($knobs)"; grep knobs gpurun_out/knob_$tag.log; python3 tools/kstats.py "$f" ${PROF_ROWS:-14}
done

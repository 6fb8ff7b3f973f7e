#!/bin/bash
# Diagnostic: average time of the fused attention kernel in the bf16 batch-64 forward for several builds of the library
# (speech_separation_amd/lib/ab/libmsf2_<name>.so; "base" = the product library), all on ONE GPU box.
cd /root/repo
cmd="cd /tmp && export TMPDIR=/tmp"
for l in "$@"; do
  if [ "$l" = base ]; then e=""; else e="MSF2_LIB=/root/repo/speech_separation_amd/lib/ab/libmsf2_$l.so"; fi
  cmd="$cmd && $e timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ab_$l -o a -- python3 /root/repo/tools/ab_knob.py attn_fused 64 > /root/repo/gpurun_out/ab_$l.out 2>&1"
done
timeout 1700 /usr/local/graft/bin/gpurun --timeout 900 -- "$cmd" 2>&1 | grep -v "every call\|^\[gpurun\] sending"
for l in "$@"; do echo "== $l"; python tools/kstats.py gpurun_out/ab_$l/a_kernel_stats.csv 40 | grep -i "attn\|AttnOut\|AttnSim"; done

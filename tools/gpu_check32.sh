#!/bin/bash
# Diagnostic helper: GPU parity tests + a profiled fp32 bench (C2) on the GPU box; prints the summary.
cd /root/repo
timeout 1700 /usr/local/graft/bin/gpurun --timeout 900 -- "timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t1.log 2>&1 && cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_c -o a -- python3 /root/repo/bench.py --no-cpu-baseline --steps 4 --warmup 2 $* > /root/repo/gpurun_out/b3.log 2>&1" 2>&1 | grep -v "every call\|^\[gpurun\] sending"
tail -3 gpurun_out/t1.log
grep -h '"metric"' gpurun_out/b3.log | cut -c1-200
python tools/kstats.py gpurun_out/prof_c/a_kernel_stats.csv ${TOPN:-12}

#!/bin/bash
# Diagnostic helper: regenerates the bench lines and rocprofv3 kernel summaries kept under profiles/ on one GPU box
# (bf16 64 x 4 s and fp32 16 x 4 s: the plain bench line, then the same command under rocprofv3 --kernel-trace --stats;
# the profiled fp32 run leaves out the secondary bf16 measurement so that its kernel table is fp32 only).
cd /root/repo
timeout 1900 /usr/local/graft/bin/gpurun --timeout 1100 -- "timeout -k 10 200 python bench.py --precision bf16 --batch 64 > gpurun_out/bf16.json 2> gpurun_out/bf16.err && timeout -k 10 200 python bench.py > gpurun_out/fp32.json 2> gpurun_out/fp32.err && cd /tmp && export TMPDIR=/tmp && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_bf16 -o a -- python3 /root/repo/bench.py --precision bf16 --batch 64 > /root/repo/gpurun_out/bf16_rocprof.json 2> /root/repo/gpurun_out/bf16_rocprof.err && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_fp32 -o a -- python3 /root/repo/bench.py --no-secondary > /root/repo/gpurun_out/fp32_rocprof.json 2> /root/repo/gpurun_out/fp32_rocprof.err" 2>&1 | grep -v "every call\|^\[gpurun\] sending"
for f in bf16 fp32 bf16_rocprof fp32_rocprof; do tail -1 gpurun_out/$f.json | cut -c1-160; done

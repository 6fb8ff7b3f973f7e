#!/bin/bash
# Diagnostic helper: regenerates, on ONE GPU box, everything kept under profiles/ for this round (prefix given as $1,
# default r3): the bench lines, the rocprofv3 kernel summaries of the same commands, the FETCH_SIZE / WRITE_SIZE passes
# of the dominant kernel (separate --pmc runs), the long-form / sweep / segmented-window / batch-1 latency lines and the
# FSMN-form bench lines with their kernel summaries.
P=${1:-r3}
cd /root/repo
O=/root/repo/gpurun_out/$P
RP="rocprofv3 --kernel-trace --stats --output-format csv"
B="python3 /root/repo/bench.py"
cmd="mkdir -p $O && cd /tmp && export TMPDIR=/tmp"
cmd="$cmd && timeout -k 10 300 $B > $O/fp32_b16_bench.json 2> $O/fp32.err"
cmd="$cmd && timeout -k 10 300 $B --precision bf16 --batch 64 > $O/bf16_b64_bench.json 2> $O/bf16.err"
cmd="$cmd && timeout -k 10 300 $RP -d $O/prof_fp32 -o a -- $B --no-secondary > $O/fp32_b16_bench_under_rocprof.json 2> $O/fp32_rocprof.err"
cmd="$cmd && timeout -k 10 300 $RP -d $O/prof_bf16 -o a -- $B --precision bf16 --batch 64 --no-secondary > $O/bf16_b64_bench_under_rocprof.json 2> $O/bf16_rocprof.err"
cmd="$cmd && timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fp32_fetch -o a -- python3 /root/repo/tools/probe_op.py tcn_pointwise --reps 5 > $O/pmc1.out 2>&1"
cmd="$cmd && timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_fp32_write -o a -- python3 /root/repo/tools/probe_op.py tcn_pointwise --reps 5 > $O/pmc2.out 2>&1"
cmd="$cmd && timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_bf16_fetch -o a -- python3 /root/repo/tools/time_tcn16.py > $O/pmc3.out 2>&1"
cmd="$cmd && timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_bf16_write -o a -- python3 /root/repo/tools/time_tcn16.py > $O/pmc4.out 2>&1"
# long form (one 60 s mixture), the segmented path's window batch (81 windows of 1 s per minute of audio), batch-1 latency
cmd="$cmd && timeout -k 10 200 $B --workload c4 --steps 3 --warmup 1 > $O/c4_fp32.json 2>> $O/cfg.err && timeout -k 10 200 $B --workload c4 --precision bf16 --steps 3 --warmup 1 > $O/c4_bf16.json 2>> $O/cfg.err"
cmd="$cmd && timeout -k 10 200 $B --batch 81 --seconds 1 --no-cpu-baseline --no-secondary > $O/seg_fp32.json 2>> $O/cfg.err && timeout -k 10 200 $B --batch 81 --seconds 1 --precision bf16 --no-cpu-baseline --no-secondary > $O/seg_bf16.json 2>> $O/cfg.err"
cmd="$cmd && timeout -k 10 200 python3 /root/repo/tools/latency_b1.py > $O/latency_b1.jsonl 2>> $O/cfg.err"
timeout 2700 /usr/local/graft/bin/gpurun --timeout 1200 -- "$cmd" 2>&1 | grep -v "every call\|^\[gpurun\] sending"
# second call (a call is limited to 20 minutes): the sweep and the FSMN form
cmd="mkdir -p $O && cd /tmp && export TMPDIR=/tmp"
# the 3000-utterance sweep on one GPU: resident inputs (value) and from host memory (value_incl_host_copies)
cmd="$cmd && timeout -k 10 500 $B --workload c5 --steps 4 --warmup 1 --host-copy-pass > $O/c5_fp32.json 2>> $O/cfg.err && timeout -k 10 300 $B --workload c5 --precision bf16 --steps 4 --warmup 1 --host-copy-pass > $O/c5_bf16.json 2>> $O/cfg.err"
# the FSMN form: bench lines and kernel summaries of the same commands
cmd="$cmd && timeout -k 10 300 $RP -d $O/prof_fsmn_fp32 -o a -- $B --recurrent fsmn --no-cpu-baseline --no-secondary > $O/fsmn_fp32_b16_bench.json 2>> $O/cfg.err"
cmd="$cmd && timeout -k 10 300 $RP -d $O/prof_fsmn_bf16 -o a -- $B --recurrent fsmn --precision bf16 --batch 64 --no-cpu-baseline --no-secondary > $O/fsmn_bf16_b64_bench.json 2>> $O/cfg.err"
timeout 2700 /usr/local/graft/bin/gpurun --timeout 1200 -- "$cmd" 2>&1 | grep -v "every call\|^\[gpurun\] sending"
ls $O
# copy what is kept into profiles/ (tracked) and rebuild the traffic record bench.py reads
V=$(python3 -c "from speech_separation_amd import _lib; print(_lib.load().msf2_version())")
cp $O/fp32_b16_bench.json profiles/${P}_fp32_b16_bench.json
cp $O/bf16_b64_bench.json profiles/${P}_bf16_b64_bench.json
cp $O/fp32_b16_bench_under_rocprof.json profiles/${P}_fp32_b16_bench_under_rocprof.json
cp $O/bf16_b64_bench_under_rocprof.json profiles/${P}_bf16_b64_bench_under_rocprof.json
cp $O/prof_fp32/a_kernel_stats.csv profiles/${P}_fp32_b16_bench_kernel_stats.csv
cp $O/prof_bf16/a_kernel_stats.csv profiles/${P}_bf16_b64_bench_kernel_stats.csv
cp $O/pmc_fp32_fetch/a_counter_collection.csv profiles/${P}_fp32_tcn_pmc_fetch.csv
cp $O/pmc_fp32_write/a_counter_collection.csv profiles/${P}_fp32_tcn_pmc_write.csv
cp $O/pmc_bf16_fetch/a_counter_collection.csv profiles/${P}_bf16_tcn_pmc_fetch.csv
cp $O/pmc_bf16_write/a_counter_collection.csv profiles/${P}_bf16_tcn_pmc_write.csv
cat $O/c4_fp32.json $O/c4_bf16.json $O/c5_fp32.json $O/c5_bf16.json > profiles/${P}_c4_c5_one_gpu.jsonl
cat $O/seg_fp32.json $O/seg_bf16.json > profiles/${P}_segmented_windows_b81.jsonl
cp $O/latency_b1.jsonl profiles/${P}_latency_b1.jsonl
cat $O/fsmn_fp32_b16_bench.json $O/fsmn_bf16_b64_bench.json > profiles/${P}_fsmn_form_bench.jsonl
cp $O/prof_fsmn_fp32/a_kernel_stats.csv profiles/${P}_fsmn_fp32_b16_kernel_stats.csv
cp $O/prof_fsmn_bf16/a_kernel_stats.csv profiles/${P}_fsmn_bf16_b64_kernel_stats.csv
python3 tools/pmc_json.py $P $V > /dev/null
ls -la profiles/ | grep ${P}_

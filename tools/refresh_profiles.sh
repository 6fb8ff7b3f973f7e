#!/bin/bash
# Diagnostic helper: regenerates, on ONE GPU box, everything kept under profiles/ for this round (prefix given as $1,
# default r2): the bench lines, the rocprofv3 kernel summaries of the same commands, the FETCH_SIZE / WRITE_SIZE passes
# of the dominant kernel (separate --pmc runs), the long-form / sweep / batch-1 latency lines and the FSMN-form bench.
P=${1:-r2}
cd /root/repo
O=/root/repo/gpurun_out/$P
RP="rocprofv3 --kernel-trace --stats --output-format csv"
cmd="mkdir -p $O && cd /tmp && export TMPDIR=/tmp"
cmd="$cmd && timeout -k 10 300 python3 /root/repo/bench.py > $O/fp32_b16_bench.json 2> $O/fp32.err"
cmd="$cmd && timeout -k 10 300 python3 /root/repo/bench.py --precision bf16 --batch 64 > $O/bf16_b64_bench.json 2> $O/bf16.err"
cmd="$cmd && timeout -k 10 300 $RP -d $O/prof_fp32 -o a -- python3 /root/repo/bench.py --no-secondary > $O/fp32_b16_bench_under_rocprof.json 2> $O/fp32_rocprof.err"
cmd="$cmd && timeout -k 10 300 $RP -d $O/prof_bf16 -o a -- python3 /root/repo/bench.py --precision bf16 --batch 64 --no-secondary > $O/bf16_b64_bench_under_rocprof.json 2> $O/bf16_rocprof.err"
cmd="$cmd && timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fp32_fetch -o a -- python3 /root/repo/tools/probe_op.py tcn_pointwise --reps 5 > $O/pmc1.out 2>&1"
cmd="$cmd && timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_fp32_write -o a -- python3 /root/repo/tools/probe_op.py tcn_pointwise --reps 5 > $O/pmc2.out 2>&1"
cmd="$cmd && timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_bf16_fetch -o a -- python3 /root/repo/tools/time_tcn16.py > $O/pmc3.out 2>&1"
cmd="$cmd && timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_bf16_write -o a -- python3 /root/repo/tools/time_tcn16.py > $O/pmc4.out 2>&1"
cmd="$cmd && cd /root/repo"
cmd="$cmd && timeout -k 10 200 python3 tools/bench_configs.py --config c4 > $O/c4_fp32.json 2>> $O/cfg.err && timeout -k 10 200 python3 tools/bench_configs.py --config c4 --precision bf16 > $O/c4_bf16.json 2>> $O/cfg.err"
cmd="$cmd && timeout -k 10 400 python3 tools/bench_configs.py --config c5 > $O/c5_fp32.json 2>> $O/cfg.err && timeout -k 10 300 python3 tools/bench_configs.py --config c5 --precision bf16 > $O/c5_bf16.json 2>> $O/cfg.err"
cmd="$cmd && timeout -k 10 200 python3 tools/latency_b1.py > $O/latency_b1.jsonl 2>> $O/cfg.err"
cmd="$cmd && timeout -k 10 300 python3 bench.py --recurrent fsmn --no-cpu-baseline > $O/fsmn_fp32_b16_bench.json 2>> $O/cfg.err && timeout -k 10 300 python3 bench.py --recurrent fsmn --precision bf16 --batch 64 --no-cpu-baseline > $O/fsmn_bf16_b64_bench.json 2>> $O/cfg.err"
timeout 2400 /usr/local/graft/bin/gpurun --timeout 1200 -- "$cmd" 2>&1 | grep -v "every call\|^\[gpurun\] sending"
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
cp $O/latency_b1.jsonl profiles/${P}_latency_b1.jsonl
cat $O/fsmn_fp32_b16_bench.json $O/fsmn_bf16_b64_bench.json > profiles/${P}_fsmn_form_bench.jsonl
python3 tools/pmc_json.py $P $V > /dev/null
ls -la profiles/ | grep ${P}_

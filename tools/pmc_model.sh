#!/bin/bash
# Diagnostic: hardware counters of chosen kernels inside the bf16 batch-64 forward (separate rocprofv3 --pmc passes, one
# counter block each), on ONE GPU box.  usage: tools/pmc_model.sh <out-name> <kernel regex> ["<ab_knob.py arguments>"]
N=$1; RE=$2; ARGS=${3:-"attn_fused 64"}
cd /root/repo
O=/root/repo/gpurun_out/$N
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_EA0_RDREQ_sum"
 "TCC_WRITE_sum TCC_EA0_WRREQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum"
)
cmd="mkdir -p $O && cd /tmp && export TMPDIR=/tmp && export AB_VALUES=1 && (rocprofv3 -L > $O/avail.txt 2>&1 || true)"
i=0
for p in "${PASSES[@]}"; do
  cmd="$cmd && timeout -s KILL 240 rocprofv3 --pmc $p --kernel-include-regex '$RE' --output-format csv -d $O/p$i -o a -- python3 /root/repo/tools/ab_knob.py $ARGS > $O/p$i.out 2>&1"
  i=$((i+1))
done
timeout 2400 /usr/local/graft/bin/gpurun --timeout 1200 -- "$cmd" 2>&1 | grep -v "every call\|^\[gpurun\] sending"
python3 - <<PY
import csv, collections, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$O/p*/a_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} | {"launches": max(len(v) for v in d.values())} for k, d in agg.items()}
json.dump(out, open("$O/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY

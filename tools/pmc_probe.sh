#!/bin/bash
# SQ counter passes over a probe binary: bash tools/pmc_probe.sh <binary> [args]  -> gpurun_out/pmc_probe/summary.txt
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_probe; rm -rf $OUT; mkdir -p $OUT
BIN=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P3="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"
P4="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
P5="SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do i=$((i+1)); rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -- $BIN "$@" > $OUT/p$i.log 2>&1; done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as out:
    for k, d in acc.items():
        out.write(f"== {k}\n")
        for c, v in sorted(d.items()):
            v = v[1:] if len(v) > 1 else v
            out.write(f"   {c:32s} {sum(v)/len(v):14.0f}\n")
print(open("$OUT/summary.txt").read())
PY

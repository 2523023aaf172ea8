#!/bin/bash
# FETCH_SIZE of every kernel of one workload's timed loop, for library/env variants: bash tools/pmc_quick.sh workload tag [ENV=..]
W=$1; TAG=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out/pmcq_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/bench.py --workload $W --steps 300 --warmup 60 --timed-only > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/bench.py --workload $W --steps 300 --warmup 60 --timed-only > $OUT/write.log 2>&1
python3 $REPO/tools/pmc_summary.py --workload $W --fetch $OUT/fetch --write $OUT/write --out $OUT/pmc_${W}_$TAG.csv
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/pmc_${W}_$TAG.csv"))); seen=set(); tot=0; base=max(int(r['calls']) for r in rows if 'tn' in r['kernel'])
for r in rows:
    k=(r['kernel'],r['threads'])
    if k in seen: continue
    seen.add(k); f=float(r['fetch_bytes'])/1e6; tot+=f*int(r['calls'])/base
    print(f"  {r['kernel']:26s} {r['threads']:>7s} x{int(r['calls'])/base:4.2f} fetch {f:6.2f} MB write {float(r['write_bytes'])/1e6:5.2f}")
print("$W $TAG fetch per iteration %.1f MB" % tot)
PY

#!/bin/bash
# Round profiles on the GPU box: kernel stats + separate FETCH_SIZE / WRITE_SIZE passes of the timed loop of each workload,
# summarised into profiles/ (run through gpurun from the repo root: `bash tools/gpu_profiles.sh r03`).
set -e
R=${1:-r03}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in hopper_sac halfcheetah_td3 humanoid_sac; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -- python3 $REPO/bench.py --workload $W --steps 600 --warmup 100 --timed-only > $OUT/stats_$W.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$W -- python3 $REPO/bench.py --workload $W --steps 300 --warmup 60 --timed-only > $OUT/fetch_$W.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$W -- python3 $REPO/bench.py --workload $W --steps 300 --warmup 60 --timed-only > $OUT/write_$W.log 2>&1
  python3 $REPO/tools/pmc_summary.py --workload $W --fetch $OUT/fetch_$W --write $OUT/write_$W --out $OUT/${R}_pmc_$W.csv
  cp $(find $OUT/stats_$W -name "*kernel_stats.csv" | head -1) $OUT/${R}_bench_${W}_kernel_stats.csv
  echo "done $W"
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_gather -- python3 $REPO/bench.py --workload humanoid_sac --gather-profile 65536 > $OUT/fetch_gather.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_gather -- python3 $REPO/bench.py --workload humanoid_sac --gather-profile 65536 > $OUT/write_gather.log 2>&1
python3 $REPO/tools/pmc_summary.py --workload gather_humanoid_b65536 --fetch $OUT/fetch_gather --write $OUT/write_gather --out $OUT/${R}_pmc_gather_humanoid_b65536.csv
# drop the raw per-dispatch traces (tens of MB): the summaries are what profiles/ keeps
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
ls -la $OUT

#!/bin/bash
# full-iteration A/B of library variants on one box: bash tools/ab_iter.sh workload "ENV=.. lib" ...
W=$1; shift
for spec in "$@"; do
  lib=${spec##* }; envs=${spec% *}; [ "$envs" == "$spec" ] && envs=""
  for rep in 1 2; do
    env $envs SACTD3_LIBRARY=$PWD/$lib python bench.py --workload $W --steps 3000 --warmup 300 --timed-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$spec', round(d['value']), '%.2f us' % (d['ms_per_step']*1e3))"
  done
done

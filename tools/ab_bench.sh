#!/bin/bash
# Same-box A/B of the train step: the working tree against a baseline export under .ab_base/ (git archive of a commit with
# its own built library), alternating, N rounds.  usage (on the GPU box): bash tools/ab_bench.sh [rounds] [extra bench flags]
rounds=${1:-3}; shift
flags="--no-cpu-baseline --no-decode --no-extras --no-roofline $*"
for i in $(seq $rounds); do
  a=$(cd .ab_base && python bench.py $flags 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  b=$(python bench.py $flags 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "round $i: base $a ms  new $b ms"
done

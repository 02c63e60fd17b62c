#!/bin/bash
# Same-box A/B of the train step between two settings of ONE environment switch, alternating, N rounds.
# usage (on the GPU box): bash tools/ab_env.sh VAR A_VALUE B_VALUE [rounds] [extra bench flags]
if [ $# -lt 3 ]; then echo "usage: bash tools/ab_env.sh VAR A_VALUE B_VALUE [rounds] [extra bench flags]" >&2; exit 2; fi
var=$1; a=$2; b=$3; rounds=${4:-3}; shift $(( $# < 4 ? $# : 4 ))  # (`shift 4` with three arguments shifts nothing in bash)
flags="--no-cpu-baseline --no-decode --no-extras --no-roofline $*"
# bench.py's stderr stays visible: a rejected flag or a failed run shows as its message, not as a JSON traceback per round
ms() { python bench.py $flags | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for i in $(seq $rounds); do
  x=$(env $var=$a bash -c "$(declare -f ms); flags='$flags'; ms")
  y=$(env $var=$b bash -c "$(declare -f ms); flags='$flags'; ms")
  echo "round $i: $var=$a $x ms   $var=$b $y ms"
done

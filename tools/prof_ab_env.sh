#!/bin/bash
# Kernel stats of the train-only bench under two settings of one environment switch.  usage: bash tools/prof_ab_env.sh VAR A B outdir [rows]
set -e -o pipefail
export TMPDIR=/tmp
var=$1; a=$2; b=$3; out=$4; rows=${5:-45}
mkdir -p $out
for v in $a $b; do
  export $var=$v
  rm -rf /tmp/prof_kt_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt_$v -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-roofline --no-extras > $out/bench_$v.json 2> $out/err_$v.log
  mkdir -p $out/$v
  cp /tmp/prof_kt_$v/*kernel_stats.csv $out/$v/kernel_stats.csv
  echo "== $var=$v"; python tools/kstats.py $out/$v 27 $rows
done

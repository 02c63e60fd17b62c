#!/bin/bash
# Where the varying-batch step's time goes against the fixed-batch step: kernel-trace stats of both, per step, side by side.
# usage (GPU box): bash tools/varying_trace.sh r05
set -e -o pipefail
tag=${1:-r05}
out=gpurun_out/vary_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vt_fixed -o kt -- python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-extras --no-roofline > $out/fixed.json 2> $out/fixed.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vt_vary -o kt -- python bench.py --varying --steps 600 --warmup 60 > $out/varying.json 2> $out/varying.err
python tools/kstats_ab.py /tmp/vt_fixed /tmp/vt_vary 45 > $out/fixed_vs_varying.txt
cat $out/fixed_vs_varying.txt

#!/bin/bash
# Kernel trace of the graph-replayed train step: the torch / runtime kernels (copies, fills, element-wise) between ours, with neighbours.
set -e -o pipefail
export TMPDIR=/tmp
out=${1:-gpurun_out/trace_foreign}
mkdir -p $out
rm -rf /tmp/prof_tf
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tf -o kt -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-decode --no-roofline --no-extras > $out/bench.json 2> $out/err.log
python tools/trace_seq_graph.py /tmp/prof_tf > $out/foreign.txt
python tools/trace_step.py /tmp/prof_tf $out/step.txt
tail -60 $out/foreign.txt

#!/bin/bash
# Round profile recipe (run on the GPU box through gpurun): default bench line, kernel-trace stats of the same
# command, and two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of an eager 2-step run.
# usage: bash tools/profile_round.sh r01
set -e -o pipefail
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/bench.json 2> $out/bench.err
tail -c 3000 $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- python bench.py > $out/bench_profiled.json 2> $out/kt.err
cp /tmp/prof_kt/*kernel_stats.csv $out/kernel_stats.csv
cp /tmp/prof_kt/*domain_stats.csv $out/domain_stats.csv 2>/dev/null || true
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt2 -o kt -- python bench.py --no-cpu-baseline --no-decode --no-extras > $out/bench_train_only_profiled.json 2> $out/kt2.err
cp /tmp/prof_kt2/*kernel_stats.csv $out/kernel_stats_train_only.csv
echo "kernel-trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o f -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-decode --no-roofline --no-extras > $out/pmc_fetch.json 2> $out/pmc_fetch.err
python tools/pmc_traffic.py /tmp/prof_f FETCH_SIZE > $out/pmc_fetch_by_kernel.csv
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o w -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-decode --no-roofline --no-extras > $out/pmc_write.json 2> $out/pmc_write.err
python tools/pmc_traffic.py /tmp/prof_w WRITE_SIZE > $out/pmc_write_by_kernel.csv
python tools/gemm_traffic.py $out $out/gemm_traffic.json
echo "write done"

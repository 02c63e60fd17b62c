#!/bin/bash
# One-rank rehearsal of the data-parallel bench path (a real one-rank RCCL communicator, JS2T_BENCH_FORCE_DDP=1) on a 1-GPU box:
# the cut step with and without the collectives, and with the backward pass in one piece (JS2T_EARLY_EXCHANGE=0).
flags="--no-cpu-baseline --no-decode --no-extras --no-roofline"
ms() { python bench.py $flags 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['config']['loss'], j['config']['capture_error'])"; }
echo "single graph (N = 1 path)            : $(ms)"
export JS2T_BENCH_FORCE_DDP=1
echo "cut at encoder output, exchange      : $(ms)"
echo "cut at encoder output, no collectives: $(JS2T_BENCH_NO_EXCHANGE=1 ms)"
export JS2T_EARLY_EXCHANGE=0
echo "one backward piece, exchange         : $(ms)"
echo "one backward piece, no collectives   : $(JS2T_BENCH_NO_EXCHANGE=1 ms)"

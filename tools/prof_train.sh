set -e -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/prof12
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-roofline > gpurun_out/prof12/bench.json 2> gpurun_out/prof12/err.log
cp /tmp/prof_kt/*kernel_stats.csv gpurun_out/prof12/kernel_stats.csv
python tools/kstats.py gpurun_out/prof12 27 40

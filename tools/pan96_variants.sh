#!/bin/bash
# Measurement builds of the panel-resident kernel (gemm_panel.hip, JS2T_PAN_DBG bits: 1 = every strip reads L2-resident rows,
# 2 = no LDS-DMA requests, 4 = no B fragment reads, 8 = no epilogue) next to the shipped library, for tools/pan96_bench.py:
#   bash tools/pan96_variants.sh build            (here: hipcc cross-compiles)
#   bash tools/pan96_variants.sh run > out.txt    (GPU box)
set -e
cd "$(dirname "$0")/.."
B=joeys2t_amd/build
VARIANTS="${PAN_VARIANTS:-1 2 3 8 9 6 15}"
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-inline-asm -DJS2T_PAN_DBG=$v -x hip -c joeys2t_amd/csrc/gemm_panel.hip -o $B/pan_dbg$v.o
    objs=$(ls $B/*.o | grep -v pan_dbg | grep -v gemm_panel.hip.o)
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $B/libpan_dbg$v.so $objs $B/pan_dbg$v.o
  done
  ls -la $B/*.so
else
  echo "== shipped"; python tools/pan96_bench.py --time-only
  for v in $VARIANTS; do echo "== JS2T_PAN_DBG=$v"; JS2T_LIB=$PWD/$B/libpan_dbg$v.so python tools/pan96_bench.py --time-only; done
fi

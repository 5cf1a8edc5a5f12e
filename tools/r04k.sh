#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
O=gpurun_out/r04k_split2.txt
{
CELLS=22 LLD=12 python tools/ab_option.py s5_split=3 s5_waves=12
CELLS=22 LLD=8 HOH=1 python tools/ab_option.py s5_split=3 s5_waves=8
tools/ab_bench.sh "--cells 22" "" "s5_split=3" "s5_split=3 s5_waves=12"
tools/ab_bench.sh "--cells 46" "" "s5_split=3" "s5_split=3 s5_waves=12"
} > $O 2>&1
cat $O

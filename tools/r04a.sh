#!/bin/bash
# round 4, first GPU call: the GPU suite, the new default bench line (46^3) as the driver runs it, and the K-split timing probe
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/r04a; mkdir -p $O
cd $ROOT
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log; tail -3 $O/tests.log
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; tail -4 $O/bench_default.err
echo "== A/B default lib vs K-split probe (22^3)"
tools/ab_lib.sh "--cells 22" rslmtoasa_amd/librsrec.so build/librsrec_ksplit.so 2>&1 | tee $O/ab_ksplit_c22.txt
echo "== PMC: default lib, probe lib, s5_waves=4 (22^3)"
BENCH_EXTRA="--cells 22" tools/profile_bench.sh r04a_c22 > $O/pmc_default.txt 2>&1; tail -8 $O/pmc_default.txt
RSREC_LIB=$ROOT/build/librsrec_ksplit.so BENCH_EXTRA="--cells 22" tools/profile_bench.sh r04a_c22_ksplit > $O/pmc_ksplit.txt 2>&1; tail -8 $O/pmc_ksplit.txt
BENCH_EXTRA="--cells 22 --opt s5_waves=4" tools/profile_bench.sh r04a_c22_w4 > $O/pmc_w4.txt 2>&1; tail -8 $O/pmc_w4.txt

#!/bin/bash
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/kubo_run; rm -rf $W; mkdir -p $W; cp $ROOT/tests/golden/scf/inputs/conductivity_fccPt/* $W/
cd $W
python3 - "$ROOT" <<'PY'
import sys,json
root=sys.argv[1]; sys.path.insert(0,root)
from oracle.make_fixtures import patch_namelist
m=json.load(open(root+'/tests/golden/scf/manifest.json'))
txt=open('input.nml').read()
open('input.nml','w').write(patch_namelist(txt,m['Generated_conductivity_fccPt_spin_hoh']['patch']))
PY
ulimit -s unlimited; export OMP_NUM_THREADS=8 OMP_STACKSIZE=1G
$ROOT/oracle/_ref/kubo_gpu.x > run.log 2>&1 || true
grep -E "kubo_gpu_driver|error|fatal" run.log | head; sed -n '500p;1000p;1500p' Pt_cond.out

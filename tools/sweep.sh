#!/bin/bash
# run <workload> [path/to/lib.so]: bench.py against the in-tree library or, through tools/with_lib.py, another build
run() { wl=$1; shift; timeout -k 10 280 python ${1:+tools/with_lib.py $1} bench.py --workload $wl --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl $*', d['ms_per_step'], d['stages_ms'])" || exit 1; }
for i in 1 2 3; do STEPS=20 run config2 gpurun_tmp/lib_logspace.so; STEPS=20 run config2;  STEPS=20 run config2 gpurun_tmp/lib_k1old_k2new.so; done

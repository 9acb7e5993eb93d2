#!/bin/bash
run() { wl=$1; shift; env "$@" timeout -k 10 280 python bench.py --workload $wl --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl $*', d['ms_per_step'], d['stages_ms'])" || exit 1; }
for i in 1 2 3; do STEPS=20 run config2 RNNT_AMD_LIB=$PWD/gpurun_tmp/lib_logspace.so; STEPS=20 run config2 A=new;  STEPS=20 run config2 RNNT_AMD_LIB=$PWD/gpurun_tmp/lib_k1old_k2new.so; done

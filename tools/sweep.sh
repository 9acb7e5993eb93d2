#!/bin/bash
run() { env "$@" timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['stages_ms'])" || exit 1; }
run A=0
for w in 2 3 4; do run RNNT_AMD_LIB=$PWD/gpurun_tmp/lib_k1wpe$w.so; done
run A=0

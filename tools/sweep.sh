#!/bin/bash
run() { env "$@" timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['stages_ms'])" || exit 1; }
for i in 1 2 3; do run A=new; run RNNT_AMD_LIB=$PWD/gpurun_tmp/lib_head.so; done

#!/bin/bash
run() { timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'], d['stages_ms'])" || exit 1; }
run; run --no-check-lengths; run --force-sharded; run; run --no-check-lengths; run --force-sharded

#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's default run (config 2): kernel-trace stats and,
# in separate passes, the HBM traffic counters.  Run on the GPU box: ./tools/profile.sh
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/trace.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1 || exit 1
find $O -name "*.csv" | head -20

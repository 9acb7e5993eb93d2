#!/bin/bash
# rocprofv3 evidence for the fused projection + loss kernel (run on the GPU box): kernel-trace stats, then in
# separate passes the matrix-pipe busy cycles and the HBM read / write bytes of every kernel of tools/bench_fused.py.
#   tools/profile_fused.sh [NAME [bench_fused.py arguments]]   e.g.  tools/profile_fused.sh ragged --ragged 3
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_fused${1:+_$1}
shift || true
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/bench_fused.py --reps 10 --out $O/bench_fused.json "$@" > $O/trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python3 $R/tools/bench_fused.py --reps 2 "$@" > $O/sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/bench_fused.py --reps 2 "$@" > $O/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/bench_fused.py --reps 2 "$@" > $O/write.log 2>&1 || exit 1
python3 $R/tools/summarize_pmc.py $O $O/pmc_summary.json
find $O -name "*kernel_stats.csv" | head -2

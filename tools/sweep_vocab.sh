#!/bin/bash
# RNN-T loss fwd+bwd at other vocabulary sizes (~2.2e9 logits each, fp32), one line per V:
#   V  ms/step  K1 GB/s  K3 GB/s  whole-step fraction of 8 TB/s
# Run on the GPU box: ./tools/sweep_vocab.sh > gpurun_out/sweep_vocab.txt
R=${GRAFT_REPO_ROOT:-$PWD}
for spec in "512,400,40,29" "512,400,40,64" "256,400,40,128" "128,400,40,256" "128,400,40,512" "64,400,40,1024" "64,400,40,2048" "32,400,40,4232" "16,400,40,8192" "8,400,40,16384" "4,400,40,32768" "2,400,40,65536"; do
  timeout -k 10 120 python3 $R/bench.py --workload custom --shape $spec,f32 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$spec', d['ms_per_step'], d['k1_achieved_GBs'], d['roofline']['achieved'], d['whole_step']['frac_of_hbm_peak'], d['stages_ms'])
" || exit 1
done

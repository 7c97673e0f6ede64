#!/bin/bash
# experiment: Omega launch as one four-wave workgroup per CU (SFFT_G1_QUAD=1), leaving half of every CU to the other pairs' kernels
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "omega_launch_variants" 2>&1 | tail -3
bash scripts/ab.sh "" SFFT_G1_QUAD=0 -- SFFT_G1_QUAD=1 -- SFFT_G1_QUAD=0 -- SFFT_G1_QUAD=1
for st in 5 6; do for q in 0 1; do SFFT_G1_QUAD=$q python bench.py --streams $st --steps 10 --warmup 2 --no-cpu --no-host-arrays --no-other-configs 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams $st quad $q', round(d['value'],1), d['post_check']['bitwise_equal'])"; done; done
} 2>&1 | tee gpurun_out/exp_q.log

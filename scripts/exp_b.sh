#!/bin/bash
# experiment: uneven row chunks of the grouped Omega launch (g1_balance_chunks) + one greek_g2 launch, against the previous build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or full_size or lhmat or system" 2>&1 | tail -4
bash scripts/ab_libs.sh lib_base.so libsfft_amd.so
bash scripts/ab.sh "" SFFT_G1_RPC=1856 -- SFFT_G1_RPC=1792 -- SFFT_G1_RPC=1376 -- SFFT_G1_BALANCE=0
} 2>&1 | tee gpurun_out/exp_b.log

#!/bin/bash
# experiment: prefetch depth of the grouped Omega launch (DF_BURST builds) and its row-chunk count
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
bash scripts/ab_libs.sh libsfft_amd.so lib_burst2.so lib_burst3.so
bash scripts/ab.sh "" SFFT_G1_S=4 -- SFFT_G1_S=8 -- SFFT_G1_S=9 -- SFFT_G1_S=6 -- SFFT_G1_S=2
} 2>&1 | tee gpurun_out/exp_a.log

#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or full_size or lhmat or system or config3 or config5" 2>&1 | tail -3
bash scripts/ab_libs.sh lib_prev.so lib_new.so
} 2>&1 | tee gpurun_out/exp_f.log

# MFMA / VALU utilisation of the hot kernels (separate PMC pass; --kernel-trace only).  Run on the GPU box:
#   bash scripts/pmc_mfma.sh   -> gpurun_out/profiles/pmc_mfma.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "MFMA|SQ_BUSY_CYCLES|SQ_ACTIVE_INST_VALU|SQ_INSTS_VALU\b|GRBM_GUI_ACTIVE" | head -40 > $OUT/pmc_avail_mfma.txt
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  tag=$(echo $c | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --batch 4 --steps 2 --warmup 1 --no-cpu --no-host-arrays --no-other-configs > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/scripts/pmc_summary.py /tmp/pmc_$tag/p_counter_collection.csv 2>/dev/null | grep -E "kernel|greek|chol_step|chol_dataflow|vconv|cols_fwd|rows_r2c" >> $OUT/pmc_mfma.txt
done
cat $OUT/pmc_mfma.txt

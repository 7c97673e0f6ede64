# How busy are the matrix pipe and the vector ALUs during the Omega launches of a config?  (close the over-fetch question
# with counters.)  One rocprofv3 --pmc pass per counter pair (--kernel-trace only), one pair in flight; mean per launch of the kernels matching PAT.
#   CFG=3 bash scripts/pmc_issue.sh > gpurun_out/r06/pmc_issue_cfg3.txt          (on the GPU box, via gpurun)
cd /tmp && export TMPDIR=/tmp
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $c | tr ' ' '_')
  rm -rf /tmp/pmc_$tag
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --config ${CFG:-3} --streams 1 --batch 2 --steps 2 --warmup 1 --no-cpu --no-host-arrays --no-other-configs > /dev/null 2>&1
  echo "== $c"; python $GRAFT_REPO_ROOT/scripts/pmc_summary.py /tmp/pmc_$tag/p_counter_collection.csv 2>/dev/null | grep -E "${PAT:-greek_g1_mfma}"
done

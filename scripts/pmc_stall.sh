# What the hot kernels wait for (run on the GPU box via gpurun): wave-cycle, s_waitcnt, LDS and L1 stall counters per kernel, one
# rocprofv3 --pmc pass per counter pair (with --kernel-trace only), mean per launch.  CFG=2|3|5 picks the bench config, PAT the kernels.
#   CFG=2 bash scripts/pmc_stall.sh > gpurun_out/pmc_stall_cfg2.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_WAIT[A-Z_]*|SQ_ACTIVE_INST[A-Z_]*|SQ_INST_CYCLES[A-Z_]*|SQ_INSTS_[A-Z_]*|SQ_LDS[A-Z_]*|TCP_PENDING[A-Z_]*|TCP_TCC[A-Z_]*|TA_BUSY[a-z\[\]0-9_]*|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|TCP_TA_TCP_STATE_READ[A-Z_]*|TCP_GATE_EN[0-9]*[A-Z_]*)\b" | sort -u | tr '\n' ' '
echo
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "TCP_PENDING_STALL_CYCLES TCP_GATE_EN1"; do
  tag=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --config ${CFG:-3} --streams 1 --batch 2 --steps 2 --warmup 1 --no-cpu --no-host-arrays --no-other-configs > /dev/null 2>&1
  echo "== $c"; python $GRAFT_REPO_ROOT/scripts/pmc_summary.py /tmp/pmc_$tag/p_counter_collection.csv 2>/dev/null | grep -E "${PAT:-rows_r2c|cols_fwd|vconv|rows_c2r|omega_sparse|greek_g1_mfma}"
done

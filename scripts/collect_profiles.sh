# Run on the GPU box (via gpurun): rocprofv3 kernel stats + PMC traffic passes for bench.py, summaries into gpurun_out/profiles/
# usage: bash scripts/collect_profiles.sh [extra bench flags, e.g. --config 3]
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles; mkdir -p $OUT
EXTRA="$@"
cd /tmp && export TMPDIR=/tmp
ONE="--streams 1 --batch 4 --no-cpu --no-host-arrays --no-other-configs --no-lu-leg $EXTRA"
# (1) kernel trace + stats, one pair in flight (kernel durations not interleaved with other streams)
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python $GRAFT_REPO_ROOT/bench.py $ONE --steps 5 --warmup 2 > $OUT/bench_streams1.json 2> /dev/null
cp $GRAFT_REPO_ROOT/profiles/bench_last_full.json $OUT/bench_streams1_full.json      # (carries stage_kernels: which kernels each stage launched)
python $GRAFT_REPO_ROOT/scripts/prof_stats.py /tmp/ks > $OUT/kernel_stats_streams1.txt
rm -rf /tmp/ks
# (2) PMC passes (separate runs; --pmc with --kernel-trace only)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py $ONE --steps 2 --warmup 1 > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/scripts/pmc_summary.py /tmp/pmc_$c/p_counter_collection.csv > $OUT/pmc_$c.txt
  rm -rf /tmp/pmc_$c
done

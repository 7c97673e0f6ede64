# Run on the GPU box (via gpurun): rocprofv3 kernel stats + PMC traffic passes for bench.py, summaries into gpurun_out/profiles/
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (1) kernel trace + stats, one pair in flight (kernel durations not interleaved with other streams)
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 10 --warmup 3 --cpu-sample 0 > $OUT/bench_streams1.json 2> /dev/null
python $GRAFT_REPO_ROOT/scripts/prof_stats.py /tmp/ks > $OUT/kernel_stats_streams1.txt
# (2) PMC passes (separate runs; --pmc with --kernel-trace only)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/scripts/pmc_summary.py /tmp/pmc_$c/p_counter_collection.csv > $OUT/pmc_$c.txt
done
python $GRAFT_REPO_ROOT/scripts/make_pmc_traffic.py $OUT/pmc_FETCH_SIZE.txt $OUT/pmc_WRITE_SIZE.txt > $OUT/pmc_traffic.json
cp $OUT/pmc_traffic.json $GRAFT_REPO_ROOT/profiles/pmc_traffic.json   # so that the default bench line below carries the fresh traffic figure
# (3) the default bench line (pipelined), unprofiled
python $GRAFT_REPO_ROOT/bench.py > $OUT/bench_default.json 2> /dev/null
tail -c 600 $OUT/bench_default.json

#!/bin/bash
# (1) pairs in flight / hardware queues sweep at config 2; (2) dataflow Cholesky on the big systems of configs 3 and 5
for s in 3 4 5 6 8; do
  for q in 8 16; do
    GPU_MAX_HW_QUEUES=$q python bench.py --streams $s --steps 8 --warmup 2 --no-cpu --no-host-arrays 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('streams $s queues $q : %.1f pairs/s' % d['value'])"
  done
done
for c in 3 5; do
  for e in "SFFT_X=0" "SFFT_CHOL_OUTER_MIN=100000 SFFT_CHOL_DF_WG=128" "SFFT_CHOL_OUTER_MIN=100000 SFFT_CHOL_DF_WG=256" "SFFT_CHOL_OUTER_MIN=100000 SFFT_CHOL_DF_WG=512"; do
    env $e python bench.py --config $c --streams 1 --batch 2 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('config $c $e : single %.2f ms solve %.2f ms solver %s post %s' % (d['single_pair']['ms'], d['single_pair']['stage_ms']['solve'], d['config']['solver'], d['post_check']['bitwise_equal']))"
  done
done

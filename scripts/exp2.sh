#!/bin/bash
for s in 3 4 5 6; do
  python bench.py --streams $s --steps 10 --warmup 2 --no-cpu --no-host-arrays 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('streams $s : %.1f pairs/s' % d['value'])"
done
for w in 32 48 64 96; do
  SFFT_CHOL_DF_WG=$w python bench.py --streams 4 --steps 10 --warmup 2 --no-cpu --no-host-arrays 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('df_wg $w : %.1f pairs/s single %.3f solve %.3f' % (d['value'], d['single_pair']['ms'], d['single_pair']['stage_ms']['solve']))"
done

#!/bin/bash
# A/B of library builds / environment switches on the headline config: one pair in flight (stage times) and the default 4 in flight.
# usage: scripts/ab_bench.sh "<label>|<env assignments>" ...     (run on the GPU box; writes gpurun_out/ab_<label>.txt)
mkdir -p gpurun_out
for spec in "$@"; do
  label="${spec%%|*}"; envs="${spec#*|}"
  for mode in "--streams 1 --steps 4 --warmup 2" "--steps 12 --warmup 3"; do
    env $envs python bench.py --no-cpu --no-host-arrays --no-other-configs $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
st=d['single_pair']['stage_ms']
print('$label', '[$mode]', 'value %.1f' % d['value'], 'single %.3f ms' % d['single_pair']['ms'], ' '.join('%s=%.3f' % (k, st[k]) for k in ('fwd_rows','fwd_cols','greek_g1','greek_g2','solve','prelim_apply','construct','inverse')), 'bitwise', d['post_check']['bitwise_equal'])
"
  done
done | tee -a gpurun_out/ab_results.txt

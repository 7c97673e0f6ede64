#!/bin/bash
# FETCH_SIZE / duration of the Omega launch for a config under environment variants: scripts/pmc_g1.sh <config> "<env>" ...
cfg=$1; shift
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  rm -rf /tmp/pmc_x /tmp/ks_x
  env $e timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_x -o p -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --streams 1 --batch 1 --no-other-configs --no-cpu --no-host-arrays --steps 1 --warmup 1 > /dev/null 2>&1
  env $e timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_x -o ks -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --streams 1 --batch 1 --no-other-configs --no-cpu --no-host-arrays --steps 1 --warmup 1 > /dev/null 2>&1
  echo "== $e"
  python $GRAFT_REPO_ROOT/scripts/pmc_summary.py /tmp/pmc_x/p_counter_collection.csv | grep -E "greek_g1_mfma4|greek_g2|greek_g1_lastcol"
  python $GRAFT_REPO_ROOT/scripts/prof_stats.py /tmp/ks_x | grep -E "greek_g1_mfma4|greek_g2|greek_g1_lastcol"
done

#!/bin/bash
# Round-3 profile set: config 2 (kernel table, FETCH / WRITE PMC passes, default bench line with the other-config legs and the
# cpu_baseline), MFMA counters, then the kernel tables + PMC passes of configs 3 and 5.  Everything lands in gpurun_out/$1
tag=${1:-r03}; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash scripts/collect_profiles.sh > $out/collect.log 2>&1
mkdir -p $out/cfg2; cp gpurun_out/profiles/* $out/cfg2/
bash scripts/pmc_mfma.sh > $out/pmc_mfma.log 2>&1
cp gpurun_out/profiles/pmc_mfma.txt $out/cfg2/ 2>/dev/null
for c in 3 5; do
  rm -f gpurun_out/profiles/*
  bash scripts/collect_profiles.sh --config $c > $out/collect_cfg$c.log 2>&1
  mkdir -p $out/cfg$c; cp gpurun_out/profiles/* $out/cfg$c/
done
cut -c1-300 $out/cfg2/bench_default.json; cat $out/cfg2/kernel_stats_streams1.txt | head -20

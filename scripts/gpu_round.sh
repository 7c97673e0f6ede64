#!/bin/bash
# One GPU session: parity tests, the bench modes, then the profile set; everything lands in gpurun_out/$1
tag=${1:-run}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 ) > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -18 $out/pytest.log
for mode in "--config 3 --steps 5 --warmup 1" "--config 5 --steps 4 --warmup 1" "--pairs 62 --steps 5 --warmup 1"; do
  name=$(echo $mode | tr -d ' -' | cut -c1-16)
  ( time timeout 900 python bench.py $mode ) > $out/bench_$name.log 2>&1; echo "rc=$?" >> $out/bench_$name.log
  grep '^{' $out/bench_$name.log | cut -c1-220
done
bash scripts/collect_profiles.sh > $out/collect.log 2>&1
bash scripts/pmc_mfma.sh > $out/pmc_mfma.log 2>&1
mkdir -p $out/profiles; cp gpurun_out/profiles/* $out/profiles/
cut -c1-400 $out/profiles/bench_default.json; cat $out/profiles/kernel_stats_streams1.txt

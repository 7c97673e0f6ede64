#!/bin/bash
# One GPU session: parity tests, then the bench modes; everything lands in gpurun_out/$1
tag=${1:-run}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -25 $out/pytest.log
for mode in "--steps 20 --warmup 5 --no-cpu" "--config 3 --steps 5 --warmup 1" "--config 5 --steps 4 --warmup 1" "--pairs 62 --steps 3 --warmup 1"; do
  name=$(echo $mode | tr -d ' -' | cut -c1-16)
  ( time timeout 900 python bench.py $mode ) > $out/bench_$name.log 2>&1; echo "rc=$?" >> $out/bench_$name.log
  tail -c 1500 $out/bench_$name.log | tail -8
done

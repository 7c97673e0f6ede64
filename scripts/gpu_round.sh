#!/bin/bash
# The round's profile set, on the GPU box (via gpurun): for configs 2, 3 and 5 a rocprofv3 kernel table + FETCH / WRITE PMC passes with one
# pair in flight (scripts/collect_profiles.sh), the MFMA counters of config 2, pmc_traffic.json for all three configs, then the driver's
# default command (compact last line + profiles/bench_last_full.json).  Everything lands in gpurun_out/$1; copy what is to be judged into profiles/.
tag=${1:-r04}; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in 2 3 5; do
  rm -rf gpurun_out/profiles; mkdir -p gpurun_out/profiles
  bash scripts/collect_profiles.sh --config $c > $out/collect_cfg$c.log 2>&1
  mkdir -p $out/cfg$c; cp gpurun_out/profiles/* $out/cfg$c/
done
bash scripts/pmc_mfma.sh > $out/pmc_mfma.log 2>&1
cp gpurun_out/profiles/pmc_mfma.txt $out/cfg2/ 2>/dev/null
python scripts/make_pmc_traffic.py $out > $out/pmc_traffic.json && cp $out/pmc_traffic.json profiles/pmc_traffic.json
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err
cp profiles/bench_last_full.json $out/bench_default_full.json
tail -c 3500 $out/bench_default.json; for c in 2 3 5; do head -14 $out/cfg$c/kernel_stats_streams1.txt; done

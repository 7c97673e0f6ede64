#!/bin/bash
# How much of each 4096-point pass is on-chip time?  (VERDICT r05 "commit the cache-resident experiment as a build flag + driver".)
# Builds a SECOND library with -DSFFT_CACHE_RESIDENT (device_common.hpp: every workgroup of rows_r2c_4096 / cols_fwd_weighted_4096_q / _z /
# rows_c2r_diff_4096 touches the memory of the same 32 rows / 8 - 16 column tiles: the same instructions, LDS exchanges and L1 requests, next to
# no HBM traffic, wrong results) into build/cr/, then times one pair's stages with both libraries on the GPU (scripts/cache_resident.py).
#   here:        bash scripts/cache_resident.sh build
#   on the GPU:  gpurun -- 'bash scripts/cache_resident.sh run'
set -e
cd "$(dirname "$0")/.."
mode=${1:-build}
if [ "$mode" == "build" ]; then
  mkdir -p build/cr
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form -DSFFT_CACHE_RESIDENT -c -o build/cr/sfft_amd.o sfft_amd/csrc/sfft_amd.hip
  [ -f build/obj/lu.o ] || python -m sfft_amd.build
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/cr/libsfft_amd_cr.so build/cr/sfft_amd.o build/obj/lu.o
  ls -la build/cr/libsfft_amd_cr.so
else
  python scripts/cache_resident.py as_built > /tmp/cr_a.json
  SFFT_AMD_LIB=$PWD/build/cr/libsfft_amd_cr.so python scripts/cache_resident.py cache_resident > /tmp/cr_b.json
  python - <<'PY'
import json
a, b = json.load(open("/tmp/cr_a.json")), json.load(open("/tmp/cr_b.json"))
print("config 2 (4096 x 4096, KerHW 8), one pair in flight, stage times by HIP events (ms), median of %d calls" % a["calls"])
print("%-14s %10s %16s %14s   %s" % ("stage", "as built", "cache-resident", "on-chip share", "kernels"))
for k in ("fwd_rows", "fwd_cols", "prelim_apply", "inverse"):
    print("%-14s %10.4f %16.4f %13.0f %%   %s" % (k, a["ms"][k], b["ms"][k], 100.0 * b["ms"][k] / a["ms"][k], " + ".join(a["kernels"].get(k, []))))
PY
fi

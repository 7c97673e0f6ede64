# Run on the GPU box (via gpurun): the GPU tests under the guarded debug allocator (SFFT_GUARD=1: every plan buffer ends flush with
# its own mapping; 2: starts flush with it; unmapped address space on both sides), in separate processes because guarded buffers
# are never freed.  An out-of-bounds access of a kernel -- even a masked, "harmless" read -- is a GPU memory fault here.
for g in 1 2; do
  for k in "forward_spectrum" "golden or reference" "oracle and not large" "solver or outer_blocked or lu_redo or graph or lu_gpu" \
           "generic_fft_variants or rfft2" "regularisation or (varying_scaling and not large)" "varying_scaling_large" \
           "decorrelation or pcdc or grid_convolve or matching_kernel" "error_behaviour or same_tensor or contamination" \
           "mixed_domain or baseline_size or full_size or variants_agree or pair_column_pass" "bspline or BSpline or sv_ or separate or config3" "large_shapes" "random_packet or random_bspline" "config5 or strip_matches" "nircam or sharding"; do
    echo "SFFT_GUARD=$g -k $k"
    SFFT_GUARD=$g timeout 700 python -m pytest tests -m gpu -x -q --timeout 600 -k "$k" 2>&1 | grep -v "^Extension" | tail -1
  done
  for c in 3 5; do SFFT_GUARD=$g timeout 300 python bench.py --config $c --steps 1 --warmup 1 --no-other-configs 2>&1 | tail -1 | cut -c1-100; done
  SFFT_GUARD=$g timeout 300 python bench.py --streams 3 --batch 6 --steps 2 --warmup 1 --no-cpu --no-host-arrays --no-other-configs 2>&1 | tail -1 | cut -c1-100
done

#!/bin/bash
# Compute-side ablation of the bf16 edge attention at the cfg 5 scene (experiments build; results are GARBAGE, timing only):
#   tools/flash_ablate_matrix.sh            (on the GPU box, after `python cvpr2023-vlsat_amd/build.py --experiments`)
# bits: 1 no K/V loads after the first tile, 4 no exponentials, 8 no maximum, 16 no cross-half exchanges (ds_bpermute), 32 no P.V MFMAs,
#       64 no Q.K MFMAs, 256 no barrier per tile
cd "$(dirname "$0")/.."
for r in 1 2; do for a in 0 1 4 8 16 28 32 64 96 124 256 380; do
  timeout 300 python bench.py --no-cpu --no-extra --no-measure-traffic --steps 6 --warmup 2 --scenes 1 --objects 200 --points 1024 --gemm-precision bf16_mixed \
     --lib tools/bin/libvlsat_hip_exp.so --debug-option flash_ablate=$a --debug-option prof_dual=0 "$@" 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('flash_ablate=%-4s %7.2f scenes/s  %7.3f ms/step   attention %6.1f TFLOP/s (nominal flops)' % ('$a', d['value'], d['ms_per_step'], r['class_tflops'].get('flash_attn_f32', 0)))"
done; done

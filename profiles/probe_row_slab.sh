#!/bin/bash
# -DWN_TUNE_ENV build: the texture / evaluate3D point lists through row_slab_points_kernel and (WN_NO_ROW_SLAB=1) through
# plane_sorted_points_kernel, launch us over 20 launches; then the config-3 renderer's stream
export WN_HIP_LIBRARY=$GRAFT_REPO_ROOT/wavelet-noise-in-ray-tracing_amd/build/tune/libwnoise_hip.so
run() { python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-per-config --no-measured-peak --no-sustained 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(r['kernel'], round(r['avg_launch_us'],1), 'us; value', round(d['value']), d['unit'])"; }
for rep in 1 2; do
echo -n "row slab:     "; run --workload texture_points
echo -n "plane sorted: "; WN_NO_ROW_SLAB=1 run --workload texture_points
done
echo -n "row slab, always sorted (WN_ROW_SLAB_SHARE=5): "; WN_ROW_SLAB_SHARE=5 run --workload texture_points
echo -n "row slab, stream order from half a chunk (2):   "; WN_ROW_SLAB_SHARE=2 run --workload texture_points

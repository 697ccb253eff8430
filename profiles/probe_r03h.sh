#!/bin/bash
# same box: product build vs -DWN_TUNE_ENV build on the large lattices; the point-list kernel with 9 dwordx3 vs 27 dword gathers
T=$GRAFT_REPO_ROOT/wavelet-noise-in-ray-tracing_amd/build/tune/libwnoise_hip.so
run() { python bench.py "$@" --steps 20 --warmup 10 --no-cpu-baseline --no-per-config --no-sustained 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3))"; }
for args in "--lattice 1024" "--lattice 2048 --planes 256"; do
  echo -n "$args product: "; run $args
  echo -n "$args tune:    "; WN_HIP_LIBRARY=$T run $args
  echo -n "$args tune, brick kernel: "; WN_HIP_LIBRARY=$T WN_NO_MBP1=1 run $args
done
for w in texture_points; do
  echo -n "$w padded (9 x dwordx3):  "; WN_HIP_LIBRARY=$T run --workload $w
  echo -n "$w unpadded (27 x dword): "; WN_HIP_LIBRARY=$T WN_POINTS_UNPADDED=1 run --workload $w
done

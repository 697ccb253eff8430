#!/bin/bash
# -DWN_TUNE_ENV build, static ranges: workgroup w takes range rotl8(w, r) (WN_MBP_PERMUTE=10+r; workgroup w runs on XCD w mod 8),
# launch / sustained us
export WN_HIP_LIBRARY=$GRAFT_REPO_ROOT/wavelet-noise-in-ray-tracing_amd/build/tune/libwnoise_hip.so
export WN_MBP_STATIC=1
run() { python bench.py "$@" --steps 20 --warmup 10 --no-cpu-baseline --no-per-config --no-measured-peak --sustained-seconds 0.3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(r['avg_launch_us'],1), '/', round(r['sustained']['mean_us'],1), end='   ')"; }
for r in 0 1 2 3 4 5 6 7; do
  echo -n "rotl $r: "
  for args in "--lattice 1024" "--lattice 2048 --planes 256" "--lattice 512" "--workload multiband5" "--lattice 1536 --planes 128"; do WN_MBP_PERMUTE=$((10+r)) run $args; done
  echo
done

#!/bin/bash
# repeat of profiles/probe_rotate.sh for the candidates, alternating
export WN_HIP_LIBRARY=$GRAFT_REPO_ROOT/wavelet-noise-in-ray-tracing_amd/build/tune/libwnoise_hip.so
export WN_MBP_STATIC=1
run() { python bench.py "$@" --steps 20 --warmup 10 --no-cpu-baseline --no-per-config --no-measured-peak --sustained-seconds 0.5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(r['avg_launch_us'],1), '/', round(r['sustained']['mean_us'],1), end='   ')"; }
for rep in 1 2 3; do
for r in 0 1 4; do
  echo -n "rotl $r: "
  for args in "--lattice 1024" "--lattice 2048 --planes 256" "--lattice 2048 --planes 64" "--lattice 1024 --planes 128"; do WN_MBP_PERMUTE=$((10+r)) run $args; done
  echo
done
done

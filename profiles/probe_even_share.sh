#!/bin/bash
# -DWN_TUNE_ENV build: uneven brick shares for even / odd workgroups of the plane pipeline (WN_MBP_EVEN_SHARE, permille), sustained
export WN_HIP_LIBRARY=$GRAFT_REPO_ROOT/wavelet-noise-in-ray-tracing_amd/build/tune/libwnoise_hip.so
run() { python bench.py "$@" --steps 20 --warmup 10 --no-cpu-baseline --no-per-config --no-measured-peak --sustained-seconds 0.5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(r['avg_launch_us'],1), '/', round(r['sustained']['mean_us'],1), end='   ')"; }
for share in 500 530 545 560 500; do
  echo -n "even share $share: "
  for args in "--lattice 2048 --planes 256" "--lattice 1024" "--lattice 512" "--workload multiband5"; do WN_MBP_EVEN_SHARE=$share run $args; done
  echo
done

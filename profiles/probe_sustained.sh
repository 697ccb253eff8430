#!/bin/bash
# same box, alternating: builds under wavelet-noise-in-ray-tracing_amd/build/<name>/ with the sustained (>= 1 s) measurement
B=$GRAFT_REPO_ROOT/wavelet-noise-in-ray-tracing_amd/build
args="$1"; shift
run() { python bench.py $args --steps 20 --warmup 10 --no-cpu-baseline --no-per-config 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(r['avg_launch_us'],1), 'sustained', round(r['sustained']['mean_us'],1), round(r['sustained']['min_us'],1), round(r['sustained']['max_us'],1), 'fill', round(r['measured_peak']['fill_GBps']))"; }
for rep in 1 2; do
for v in product "$@"; do
  echo -n "$v: "
  if [ $v = product ]; then run; else WN_HIP_LIBRARY=$B/$v/libwnoise_hip.so run; fi
done
done

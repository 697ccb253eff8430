#!/bin/bash
# WN_TUNE_ENV build: single-band lattices through the plane pipeline (default) and through the brick kernel (WN_NO_MBP1=1)
run() { python bench.py "$@" --steps 20 --warmup 10 --no-cpu-baseline --no-per-config --no-sustained 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3))"; }
for args in "--lattice 1024" "--lattice 2048 --planes 256" "--lattice 768"; do
  echo -n "$args  plane pipeline: "; run $args
  echo -n "$args  brick kernel:   "; WN_NO_MBP1=1 run $args
done

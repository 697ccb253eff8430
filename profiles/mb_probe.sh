#!/bin/bash
# WN_TUNE_ENV build: launch time of the multiband kernel under its timing probes (WN_MBP_DEBUG)
for d in "$@"; do
  echo -n "WN_MBP_DEBUG=$d: "
  WN_MBP_DEBUG=$d timeout -k 10 120 python bench.py --workload multiband5 --steps 20 --warmup 5 --no-cpu-baseline --no-sustained --no-measured-peak 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['avg_launch_us'],1))"
done

#!/bin/bash
# WN_TUNE_ENV build: the 512^3 headline lattice through the strip kernel (default) and through the single-band plane pipeline (WN_NO_STRIP=1)
run() { python bench.py "$@" --steps 50 --warmup 10 --no-cpu-baseline --no-per-config 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(r['avg_launch_us'],1), round(r['frac'],3), 'sustained', round(r['sustained']['mean_us'],1), round(r['sustained']['frac'],3))"; }
for i in 1 2; do
echo -n "512^3 strip kernel:   "; run
echo -n "512^3 plane pipeline: "; WN_NO_STRIP=1 run
done
echo -n "512x512x64 strip:    "; run --planes 64
echo -n "512x512x64 pipeline: "; WN_NO_STRIP=1 run --planes 64

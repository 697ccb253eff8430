#!/bin/bash
# profiles/run_bench_lines.sh <tag> -- run ON THE GPU BOX: one bench.py line per BASELINE config (with cpu_baseline,
# measured fill/copy ceiling) into gpurun_out/bench_<tag>/; profiles/summarize_bench_lines.py commits them.
TAG=${1:-r02}
OUT=gpurun_out/bench_$TAG
mkdir -p $OUT
python bench.py --steps 50 --warmup 5 --two-stream-probe > $OUT/wavelet3d.json 2> $OUT/wavelet3d.err || echo "wavelet3d failed"
for wl in multiband5 turb7 perlin texture_points texture_points_perlin wavelet3d_exact; do
  python bench.py --workload $wl --steps 10 --warmup 2 --cpu-seconds 8 > $OUT/$wl.json 2> $OUT/$wl.err || echo "$wl failed"
done
python bench.py --lattice 1024 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/wavelet3d_1024.json 2> /dev/null || echo "1024 failed"
python bench.py --lattice 2048 --planes 256 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/wavelet3d_2048x2048x256.json 2> /dev/null || echo "shard failed"
WN_BENCH_BACKEND=gloo python bench.py --gpus 2 --lattice 1024 --steps 5 --warmup 1 --no-measured-peak > $OUT/two_gloo_ranks_one_gpu_1024.json 2> $OUT/two_gloo.err || echo "2-rank rehearsal failed"
for f in $OUT/*.json; do python3 - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"{sys.argv[1].split('/')[-1][:-5]:28s} {d['value']:12.1f} Ms/s  launch {r['avg_launch_us']:9.1f} us  {r['bound']} frac {r['frac']:.3f}  cpu {d.get('cpu_baseline',{}).get('value')}")
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done

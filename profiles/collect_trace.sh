#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench.py command (headline kernel); run ON THE GPU BOX
TAG=${1:-r03s}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-per-config > "$OUT/trace.json" 2> "$OUT/trace.err"
tail -c 600 "$OUT/trace.json"

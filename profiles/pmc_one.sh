#!/bin/bash
# profiles/pmc_one.sh <outdir> <counters...> -- <bench.py args>: one rocprofv3 --pmc pass of a bench.py command (GPU box).
set -u
OUT=$1; shift
CNT=()
while [ "$1" != "--" ]; do CNT+=("$1"); shift; done
shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/gpurun_out/$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc "${CNT[@]}" --output-format csv -d "$ROOT/gpurun_out/$OUT" -- python3 $ROOT/bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline --no-per-config --no-sustained --no-measured-peak > "$ROOT/gpurun_out/$OUT/log.txt" 2>&1
echo "rc=$?"

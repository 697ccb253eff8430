#!/bin/bash
# profiles/collect_ta.sh <tag> [workload] -- run ON THE GPU BOX: the texture-addresser (TA) counters of one bench.py workload,
# ONE counter per rocprofv3 --pmc pass.  Round 2 asked for four TA_*_sum counters in one pass: rocprofv3 refused the request
# ("error code 38: Request exceeds the capabilities of the hardware to collect", the TA block has fewer slots than that),
# aborted (signal 6) and sat until the 240 s limit -- a profiler abort on the counter request, not a hang of the product.
set -u
TAG=${1:-r03}
WL=${2:-texture_points}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG/${WL}_ta
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for c in TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUSY_avr TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$i" -- python3 $ROOT/bench.py --workload $WL --steps 4 --warmup 1 --no-cpu-baseline --no-per-config --no-sustained --no-measured-peak > "$OUT/pmc_$i.log" 2>&1
  rc=$?
  echo "$c rc=$rc" | tee -a "$OUT/passes.txt"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT on $c: stopping (no further GPU step after a kill)"; exit 1; fi
  i=$((i+1))
done
echo collected-ta

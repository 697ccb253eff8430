#!/bin/bash
# profiles/collect_workloads.sh <tag> [workload ...] -- run ON THE GPU BOX (gpurun).
# Per bench.py workload: one rocprofv3 --kernel-trace --stats run and separate --pmc passes
# (never combined with tracing).  Output: gpurun_out/prof_<tag>/<workload>/ (scratch);
# profiles/summarize_workloads.py turns it into the committed per-kernel summaries.
set -u
TAG=${1:-r02}
shift || true
WORKLOADS=${*:-"perlin turb7 multiband5 texture_points texture_points_perlin"}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

# a pass that hits its time limit ends the whole collection: no further GPU step after a kill
run() { # <log> <cmd...>
  local log=$1; shift
  timeout -k 10 240 "$@" > "$log" 2>&1
  local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $* (see $log)"; exit 1; fi
  [ $rc -eq 0 ] || echo "failed rc=$rc: $log"
}

# (four TA_* counters in one pass exceed the TA block's slots: rocprofv3 refuses them with error 38 and aborts; they are taken one per
# pass by profiles/collect_ta.sh)
PASSES=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
  "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS"
  "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
  "GRBM_GUI_ACTIVE"
  "WRITE_SIZE"
  "FETCH_SIZE"
)
for wl in $WORKLOADS; do
  D=$OUT/$wl
  mkdir -p "$D"
  BENCH="python3 $ROOT/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-per-config --no-sustained"
  run "$D/trace.log" rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -- $BENCH
  i=0
  for pass in "${PASSES[@]}"; do
    run "$D/pmc_$i.log" rocprofv3 --pmc $pass --output-format csv -d "$D/pmc_$i" -- python3 $ROOT/bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-per-config --no-sustained
    i=$((i+1))
  done
  echo "collected $wl"
done
echo collected-all

#!/bin/bash
# profiles/collect_args.sh <tag> <name> <bench.py args...> -- run ON THE GPU BOX: like collect_workloads.sh (one
# rocprofv3 --kernel-trace --stats run + separate --pmc passes) for an arbitrary bench.py command line.
# Output: gpurun_out/prof_<tag>/<name>/ ; summarise with profiles/summarize_workloads.py <tag> <name> (KERNEL_OF must know <name>).
set -u
TAG=$1; NAME=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=$ROOT/gpurun_out/prof_$TAG/$NAME
mkdir -p "$D"
cd /tmp && export TMPDIR=/tmp
run() { local log=$1; shift; timeout -k 10 240 "$@" > "$log" 2>&1; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; [ $rc -eq 0 ] || echo "failed rc=$rc: $log"; }
PASSES=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
  "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT"
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
  "GRBM_GUI_ACTIVE"
  "WRITE_SIZE"
  "FETCH_SIZE"
)
run "$D/trace.log" rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -- python3 $ROOT/bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline --no-per-config --no-sustained --no-measured-peak
i=0
for pass in "${PASSES[@]}"; do
  run "$D/pmc_$i.log" rocprofv3 --pmc $pass --output-format csv -d "$D/pmc_$i" -- python3 $ROOT/bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline --no-per-config --no-sustained --no-measured-peak
  i=$((i+1))
done
echo "collected $NAME"

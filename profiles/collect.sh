#!/bin/bash
# profiles/collect.sh -- run ON THE GPU BOX (gpurun): collects the rocprofv3 evidence bench.py's
# roofline fields are checked against.  Output goes to gpurun_out/prof_<tag>/ (scratch);
# profiles/summarize.py turns it into the committed summaries.
#   --kernel-trace --stats   per-kernel durations of the bench command
#   --pmc (separate passes)  WRITE_SIZE / FETCH_SIZE / SQ counters (never combined with tracing)
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-per-config --no-sustained"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || echo "trace failed"
for pass in "WRITE_SIZE" "FETCH_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d" " -f1)
  rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$tag" -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-per-config --no-sustained > "$OUT/pmc_$tag.log" 2>&1 || echo "pmc pass $tag failed"
done
for wl in multiband5 turb7 perlin texture_points wavelet3d_exact; do
  python3 $ROOT/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-per-config --no-sustained > "$OUT/bench_$wl.json" 2> "$OUT/bench_$wl.err" || echo "bench $wl failed"
done
python3 $ROOT/bench.py --lattice 1024 --steps 20 --warmup 3 --no-cpu-baseline --no-per-config --no-sustained > "$OUT/bench_wavelet3d_1024.json" 2> /dev/null || echo "bench 1024 failed"
python3 $ROOT/bench.py --lattice 2048 --planes 256 --steps 20 --warmup 3 --no-cpu-baseline --no-per-config --no-sustained > "$OUT/bench_wavelet3d_2048x2048x256.json" 2> /dev/null || echo "bench shard failed"
python3 $ROOT/bench.py --steps 50 --warmup 5 > "$OUT/bench_wavelet3d.json" 2> "$OUT/bench_wavelet3d.err" || echo "bench failed"
echo collected

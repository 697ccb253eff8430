# knob sweeps on one MI355X (library built with -DWN_TUNE_ENV); bench.py HIP-event launch times
run() { python bench.py "$@" --no-cpu-baseline --no-per-config --no-sustained --no-measured-peak 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3))"; }
echo "512^3 strip (default): $(run --steps 50 --warmup 5)"
for k in 1 2 3 4; do echo "512^3 brick k=$k: $(WN_NO_STRIP=1 WN_SEP_K=$k run --steps 50 --warmup 5)"; done
for k in 1 2; do echo "512^3 brick xw=1 k=$k: $(WN_NO_STRIP=1 WN_SEP_XW=1 WN_SEP_K=$k run --steps 50 --warmup 5)"; done
echo "768^3 default: $(run --lattice 768 --steps 20 --warmup 3)"
echo "768^3 brick k=1: $(WN_NO_STRIP=1 WN_SEP_K=1 run --lattice 768 --steps 20 --warmup 3)"

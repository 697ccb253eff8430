run() { python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-per-config --no-sustained --no-measured-peak 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3))"; }
for cfg in "128 2 2" "64 2 2" "64 3 3" "64 3 6" "32 4 4" "32 4 8" "32 3 3" "64 2 4" "128 1 1"; do set -- $cfg
  cp profiles/libwnoise_hip_chunk$1.so wavelet-noise-in-ray-tracing_amd/libwnoise_hip.so  # variants: hipcc -DWN_TUNE_ENV -DWN_STRIP_PLANES=.. -DWN_STRIP_CHUNK=.. (not kept in the tree)
  echo "chunk=$1 wgs/cu=$2 ranges=$3: $(WN_STRIP_WGS=$2 WN_STRIP_RANGES=$3 run) | $(WN_STRIP_WGS=$2 WN_STRIP_RANGES=$3 run)"
done

for cfg in "2 8 9" "1 8 9" "1 4 9" "2 4 9" "1 8 1" "1 8 2" "1 2 9"; do set -- $cfg
  r=$(WN_SEP_XW=$1 WN_SEP_BZ=$2 WN_SEP_K=$3 python bench.py --workload multiband5 --steps 20 --warmup 3 --no-cpu-baseline --no-measured-peak 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['avg_launch_us'])")
  echo "multiband xw=$1 bz=$2 k=$3: $r us"
done
for cfg in "2 8 9" "1 8 9" "2 4 9" "1 4 9" "2 8 1" "2 8 2" "2 8 3"; do set -- $cfg
  r=$(WN_SEP_XW=$1 WN_SEP_BZ=$2 WN_SEP_K=$3 python bench.py --lattice 2048 --planes 256 --steps 10 --warmup 2 --no-cpu-baseline --no-measured-peak 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['avg_launch_us'], d['roofline']['frac'])")
  echo "slab2048 xw=$1 bz=$2 k=$3: $r"
  r=$(WN_SEP_XW=$1 WN_SEP_BZ=$2 WN_SEP_K=$3 python bench.py --lattice 1024 --steps 10 --warmup 2 --no-cpu-baseline --no-measured-peak 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['avg_launch_us'], d['roofline']['frac'])")
  echo "1024^3 xw=$1 bz=$2 k=$3: $r"
done

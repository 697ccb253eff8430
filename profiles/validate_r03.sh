#!/bin/bash
# full GPU validation of a build: tests, smoke, default bench line, two gloo ranks on one GPU, kernel trace of the 1024^3 lattice
set -e
tag=${1:-r03h}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
if [ -z "$SKIP_TESTS" ]; then
python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee $out/smoke.log
fi
python bench.py > $out/bench.json 2> $out/bench.err
python3 -c "import json;d=json.load(open('$out/bench.json'));print(d['value'],d['roofline']['avg_launch_us'],d['roofline']['frac'],d['roofline']['sustained']['mean_us'])"
WN_BENCH_BACKEND=gloo python bench.py --gpus 2 --lattice 1024 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_2rank.json 2> $out/bench_2rank.err
python3 -c "import json;d=json.loads(open('$out/bench_2rank.json').read().strip().splitlines()[-1]);print(d['value'],d.get('gather_check'),d.get('strong_scaling'))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace1024 -o t -- python3 $GRAFT_REPO_ROOT/bench.py --lattice 1024 --steps 10 --warmup 5 --no-cpu-baseline --no-per-config --no-sustained --no-measured-peak > $out/trace1024.json 2> $out/trace1024.err
python3 - <<P
import csv,glob
for f in glob.glob('$out/trace1024/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(r['Name'][:70], r['Calls'], r['AverageNs'])
P

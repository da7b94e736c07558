#!/bin/bash
# GPU box: the round's record -- bench.py (full line) and rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 3
# --no-cpu-baseline --no-extra`; tools/mk_kernel_stats_md.py TAG turns the CSV into profiles/TAG_kernel_stats.md.
# usage: bash tools/profile_r05.sh TAG
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
mkdir -p $ROOT/gpurun_out/${TAG}_stats
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $ROOT/gpurun_out/${TAG}_bench.json 2> $ROOT/gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_stats -o s -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $ROOT/gpurun_out/${TAG}_prof_bench.json 2> $ROOT/gpurun_out/${TAG}_prof.err
f=$(find $ROOT/gpurun_out/${TAG}_stats -name "*kernel_stats.csv" | head -1)
cp "$f" $ROOT/gpurun_out/${TAG}_stats/s_kernel_stats.csv 2>/dev/null
find $ROOT/gpurun_out/${TAG}_stats -name "*.csv" ! -name "s_kernel_stats.csv" -delete
find $ROOT/gpurun_out/${TAG}_stats -type d -empty -delete
python3 -c "import json; d=json.loads(open('$ROOT/gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline_k1']['frac'], d.get('roofline_conv'))"

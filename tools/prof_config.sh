#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of `bench.py --config CONFIG --steps 10 --warmup 3 --no-cpu-baseline --no-extra`
# -> gpurun_out/TAG_stats/s_kernel_stats.csv, gpurun_out/TAG_prof_bench.json and the top kernels on stdout
# usage: bash tools/prof_config.sh TAG CONFIG [extra bench.py flags]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; CFG=$2; shift 2
mkdir -p $ROOT/gpurun_out/${TAG}_stats
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_stats -o s -- python3 $ROOT/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-extra "$@" > $ROOT/gpurun_out/${TAG}_prof_bench.json 2> $ROOT/gpurun_out/${TAG}_prof.err
f=$(find $ROOT/gpurun_out/${TAG}_stats -name "*kernel_stats.csv" | head -1)
cp "$f" $ROOT/gpurun_out/${TAG}_stats/s_kernel_stats.csv 2>/dev/null
find $ROOT/gpurun_out/${TAG}_stats -name "*.csv" ! -name "s_kernel_stats.csv" -delete
find $ROOT/gpurun_out/${TAG}_stats -type d -empty -delete
python3 - $ROOT/gpurun_out/${TAG}_stats/s_kernel_stats.csv $ROOT/gpurun_out/${TAG}_prof_bench.json <<'PY'
import csv, sys, json, re
rows = list(csv.DictReader(open(sys.argv[1])))
b = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
steps = max([int(r['Calls']) for r in rows if 'wide_bwd4' in r['Name'] or 'adam_flat' in r['Name']] or [1])
tot = sum(int(r['TotalDurationNs']) for r in rows)
print('bench under the profiler: %.3f ms/step; %d steps traced; device ms per step %.2f; launches per step %.0f'
      % (b['ms_per_step'], steps, tot / 1e6 / steps, sum(int(r['Calls']) for r in rows) / steps))
for r in rows[:45]:
    nm = re.sub(r'\(anonymous namespace\)::', '', r['Name'])[:100]
    print('%-100s calls/step %6.1f avg %9.1f us  ms/step %8.3f' % (nm, int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, int(r['TotalDurationNs']) / 1e6 / steps))
PY

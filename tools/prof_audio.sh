#!/bin/bash
# GPU box: tools/time_audio.py alone, then under rocprofv3 --kernel-trace --stats (kernels alone on the chip)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r06_audio}
mkdir -p $ROOT/gpurun_out/${TAG}_stats
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/time_audio.py 512 128 5 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_stats -o s -- python3 $ROOT/tools/time_audio.py 512 128 3 > /dev/null 2>&1
f=$(find $ROOT/gpurun_out/${TAG}_stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:30]:
    nm = re.sub(r'\(anonymous namespace\)::', '', r['Name'])[:100]
    print('%-100s calls %4s avg %9.1f us' % (nm, r['Calls'], float(r['AverageNs']) / 1e3))
PY
find $ROOT/gpurun_out/${TAG}_stats -name "*.csv" -delete; find $ROOT/gpurun_out/${TAG}_stats -type d -empty -delete

#!/bin/bash
# usage (GPU box): bash tools/profile_round.sh TAG   -> gpurun_out/TAG_bench.json, TAG_stats/ (rocprofv3
# --kernel-trace --stats of the same bench command), to be summarised into profiles/ by tools/profile_summary.py
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $ROOT/gpurun_out/${TAG}_bench.json 2> $ROOT/gpurun_out/${TAG}_bench.err
tail -c 600 $ROOT/gpurun_out/${TAG}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_stats -o s -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/${TAG}_stats.log 2>&1
tail -1 $ROOT/gpurun_out/${TAG}_stats.log | cut -c1-160

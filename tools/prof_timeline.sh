#!/bin/bash
# GPU box: kernel trace of the graph-replayed cfg3 step -> gpurun_out/<tag>_timeline.txt (tools/timeline.py)
tag=${1:-r02}
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
d="$ROOT/gpurun_out/${tag}_trace"; mkdir -p "$d"
rocprofv3 --kernel-trace --output-format csv -d "$d" -o t -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-extra ${BENCH_ARGS:-} > "$d/bench.json" 2> "$d/err.log"
f=$(find "$d" -name '*kernel_trace.csv' | head -1)
python3 "$ROOT/tools/timeline.py" "$f" ${BIN_MS:-1.0} ${2:-5} > "$ROOT/gpurun_out/${tag}_timeline.txt"
cp "$f" "$ROOT/gpurun_out/${tag}_kernel_trace.csv"
cat "$ROOT/gpurun_out/${tag}_timeline.txt"

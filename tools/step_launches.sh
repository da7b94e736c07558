#!/bin/bash
# GPU box: every launch of ONE graph-replayed step (the step tools/timeline.py picks), aggregated by kernel name:
# calls, total and mean duration -- what the replayed step itself runs (no warm-up, no eager probe steps).
# usage: bash tools/step_launches.sh TAG [CONFIG]
tag=${1:-r06}; cfg=${2:-cfg3}
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
d="$ROOT/gpurun_out/${tag}_trace"; mkdir -p "$d"
rocprofv3 --kernel-trace --output-format csv -d "$d" -o t -- python3 "$ROOT/bench.py" --config $cfg --steps 8 --warmup 2 --no-cpu-baseline --no-extra > "$d/bench.json" 2> "$d/err.log"
f=$(find "$d" -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY' | tee "$ROOT/gpurun_out/${tag}_step_launches.txt"
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])) for r in rows)
ends = [i for i, x in enumerate(ev) if 'FusedOptimizer' in x[2] or 'adam_flat_kernel' in x[2]]
gaps = [(a, b) for a, b in zip(ends, ends[1:]) if b - a > 100]
lo, hi = gaps[6][0] + 1, gaps[6][1] + 1
step = ev[lo:hi]
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, _g in step:
    n = re.sub(r'\(anonymous namespace\)::|void |at::native::|_ZN12_GLOBAL__N_1\d+', '', n)[:96]
    agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
print('one replayed step: %d launches, wall %.3f ms, device time summed %.3f ms' % (len(step), (step[-1][1] - step[0][0]) / 1e6, sum(v[1] for v in agg.values()) / 1e3))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print('%4d x %9.1f us = %8.3f ms  %s' % (c, t / c, t / 1e3, n))
import os
pat = os.environ.get('SHOW', '')
if pat:
    print('launches matching %r (start ms, us, grid threads, name; with the launch before it):' % pat)
    for i, (s0, e0, n, g) in enumerate(step):
        if pat in n:
            prev = re.sub(r'\(anonymous namespace\)::|void |at::native::', '', step[i - 1][2])[:50] if i else ''
            print('%8.3f %8.1f %12d  after %s' % ((s0 - step[0][0]) / 1e6, (e0 - s0) / 1e3, g, prev))
PY
rm -rf "$d"

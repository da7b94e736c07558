#!/bin/bash
# GPU box: L2 fetch volume of the wide weight-gradient kernel for the launch variants of tools/ab_wgrad.sh
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/pmc_wgrad"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in "64 0" "42 0" "42 1"; do
  set -- $cfg
  export MDMM_WGRAD_SPLIT=$1 MDMM_WGRAD_XCD=$2
  d="$OUT/s$1_x$2"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$d" -o p -- python3 "$ROOT/tools/bench_sweep.py" P=4 B=256 T=40 D=256 H=256 n=2 bf16=1 K=25 > "$d.log" 2>&1
  f=$(find "$d" -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$cfg" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'wide_' in r['Kernel_Name']:
        agg[r['Kernel_Name'].split('::')[-1][:40]].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(sys.argv[2], k, 'FETCH_SIZE KiB avg %.0f (n=%d)' % (sum(v) / len(v), len(v)))
PY
done

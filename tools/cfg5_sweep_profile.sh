#!/bin/bash
# GPU box: the K = 1 (MAP) sweeps at cfg5's per-GPU size (vidTIMIT-shaped: T = 128, 2 modalities -> 3 passes,
# z = h = 256, 512 sequences per GPU = 4096 / 8) in isolation: per-launch time by HIP events and the HBM
# traffic from FETCH_SIZE / WRITE_SIZE (separate --pmc passes).  usage: tools/cfg5_sweep_profile.sh OUTDIR
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/$1"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="K=1 P=3 B=512 T=128 D=256 H=256 n=3 bf16=1 inv=0 rev=1"
python3 "$ROOT/tools/bench_sweep.py" $ARGS > "$OUT/time.log" 2>&1
cat "$OUT/time.log" | grep ms/launch
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o p -- python3 "$ROOT/tools/bench_sweep.py" $ARGS > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'wide' not in k: continue
        k = k.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in sorted(agg.items()):
    rd = sum(cs['FETCH_SIZE']) / max(len(cs['FETCH_SIZE']), 1) * 1024 * 2
    wr = sum(cs['WRITE_SIZE']) / max(len(cs['WRITE_SIZE']), 1) * 1024
    print('%-40s read %.1f MB (FETCH_SIZE x2)  write %.1f MB  per launch' % (k, rd / 1e6, wr / 1e6))
PY
find "$OUT" -name "*.csv" -delete; find "$OUT" -name "*.db" -delete

#!/bin/bash
# GPU box: instruction mix of every kernel of the cfg3 step (one --pmc pass) next to its duration: which
# kernels are bound by instruction issue rather than by memory (t_valu = VALU x 4 cycles / 1024 SIMDs)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/mix"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY \
  --output-format csv -d "$OUT" -o m -- python3 "$ROOT/bench.py" --eager --steps 2 --warmup 1 --no-cpu-baseline --no-extra ${BENCH_ARGS:-} > "$OUT/bench.json" 2> "$OUT/err.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVES':
            n[r['Kernel_Name']] += 1
dur = collections.defaultdict(float)
for f in glob.glob(out + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name']] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
rows = []
for k, c in cnt.items():
    if not n[k] or dur[k] < 200:
        continue
    t_valu = c['SQ_INSTS_VALU'] * 4 / (1024 * 2.4e3)      # us
    rows.append((dur[k], k, n[k], t_valu, c))
rows.sort(reverse=True)
print('%-58s %5s %9s %9s %6s | per launch: VALU SALU VMEM LDS MFMA (millions)' % ('kernel', 'calls', 'us/call', 't_valu', 'ratio'))
for d, k, calls, tv, c in rows[:32]:
    name = re.sub(r'\(anonymous namespace\)::|void |at::native::', '', k)[:58]
    print('%-58s %5d %9.1f %9.1f %6.2f | %6.2f %6.2f %6.2f %6.2f %6.2f' % (name, calls, d / calls, tv / calls, tv / d,
          c['SQ_INSTS_VALU'] / calls / 1e6, c['SQ_INSTS_SALU'] / calls / 1e6, c['SQ_INSTS_VMEM'] / calls / 1e6,
          c['SQ_INSTS_LDS'] / calls / 1e6, c['SQ_INSTS_MFMA'] / calls / 1e6))
PY
find "$OUT" -name "*.csv" -delete

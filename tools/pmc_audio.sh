#!/bin/bash
# GPU box: instruction mix and stall split of the audio nodes' kernels alone on the chip (tools/time_audio.py), two --pmc passes
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/pmc_audio"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \
  --output-format csv -d "$OUT/a" -o m -- python3 "$ROOT/tools/time_audio.py" 512 128 1 > /dev/null 2> "$OUT/err_a.log"
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM \
  --output-format csv -d "$OUT/b" -o m -- python3 "$ROOT/tools/time_audio.py" 512 128 1 > /dev/null 2> "$OUT/err_b.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] in ('SQ_WAVES',):
            n[r['Kernel_Name']] += 1
dur = collections.defaultdict(float); nd = collections.Counter()
for f in glob.glob(out + '/a/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name']] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; nd[r['Kernel_Name']] += 1
print('per launch; cycles in millions of quad-cycles summed over waves (WAVE = lifetime, WAITANY = parked at waitcnt/barrier, WINST = issue stalls, ACT = issuing)')
print('%-52s %8s | %7s %7s %7s %6s | %7s %7s %7s %7s | %6s %6s' % ('kernel', 'us', 'VALU', 'LDS', 'SALU', 'VMEM', 'WAVE', 'WAITANY', 'WINST', 'ACT', 'LDSact', 'confl'))
for k in sorted(cnt, key=lambda k: -dur[k]):
    if 'audio' not in k or not n[k]:
        continue
    c, m = cnt[k], n[k]
    name = re.sub(r'\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1\d+', '', k)[:52]
    print('%-52s %8.1f | %7.2f %7.2f %7.2f %6.2f | %7.1f %7.1f %7.1f %7.1f | %6.1f %6.1f' % (name, dur[k] / max(nd[k], 1),
          c['SQ_INSTS_VALU'] / m / 1e6, c['SQ_INSTS_LDS'] / m / 1e6, c['SQ_INSTS_SALU'] / m / 1e6, c['SQ_INSTS_VMEM'] / m / 1e6,
          c['SQ_WAVE_CYCLES'] / m / 1e6, c['SQ_WAIT_ANY'] / m / 1e6, c['SQ_WAIT_INST_ANY'] / m / 1e6, c['SQ_ACTIVE_INST_ANY'] / m / 1e6,
          c['SQ_LDS_IDX_ACTIVE'] / m / 1e6, c['SQ_LDS_BANK_CONFLICT'] / m / 1e6))
PY
find "$OUT" -name "*.csv" -delete

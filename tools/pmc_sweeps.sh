#!/bin/bash
# GPU box: SQ counters of the sweep kernels of one cfg2 step (separate --pmc passes, kernel-trace only).
# usage: tools/pmc_sweeps.sh OUTDIR   (OUTDIR under gpurun_out/)
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/$1"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
  "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
  "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o p -- python3 "$ROOT/tools/run_sweep.py" 2 > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'sweep' not in k: continue
        k = k.split('(')[0].replace('void (anonymous namespace)::', '')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fo:
    for k, cs in sorted(agg.items()):
        fo.write(k + '\n')
        for c, v in sorted(cs.items()):
            fo.write('   %-32s %16.0f  (n=%d)\n' % (c, sum(v) / len(v), len(v)))
print(open(out + '/summary.txt').read())
PY

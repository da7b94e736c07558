#!/bin/bash
# GPU box: SQ counters of one csrc/gemm_heads.hip kernel (separate --pmc passes; a fifth pass with five TCC_* counters
# was refused by the profiler -- "exceeds the capabilities of the hardware" -- and then hung: keep passes small).
# usage: tools/pmc_heads.sh OUTDIR contract|expand|wgrad [M]
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/$1"; shift; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
  "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o p -- python3 "$ROOT/tools/heads_one.py" "$@" > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        m = re.search(r'((contract|expand|wgrad|gemm)_\w+)', k)
        if not m: continue
        agg[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fo:
    for k, cs in sorted(agg.items()):
        fo.write(k + '\n')
        for c, v in sorted(cs.items()):
            fo.write('   %-32s %16.0f  (n=%d)\n' % (c, sum(v) / len(v), len(v)))
print(open(out + '/summary.txt').read())
PY
rm -rf "$OUT"/p[0-9]

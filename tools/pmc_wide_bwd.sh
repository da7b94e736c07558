#!/bin/bash
# GPU box: SQ / SQC counters of the wide K = 25 backward sweep in isolation (separate --pmc passes).
# usage: tools/pmc_wide_bwd.sh OUTDIR [bench_sweep args...]   (OUTDIR under gpurun_out/)
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/$1"; shift; mkdir -p "$OUT"
ARGS="${*:-K=25 P=4 B=256 T=40 D=256 H=256 bf16=1 n=2}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQC?_[A-Z0-9_]+|TCP_[A-Z0-9_]+|TCC_[A-Z0-9_]+)\b" | sort -u > "$OUT/counters_available.txt"
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
  "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
  "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"; do
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
        import re
        m = re.search(r'(wide_\w+(<[^>]*>)?)', k)
        k = m.group(1) if m else k
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fo:
    for k, cs in sorted(agg.items()):
        fo.write(k + '\n')
        for c, v in sorted(cs.items()):
            fo.write('   %-32s %16.0f  (n=%d)\n' % (c, sum(v) / len(v), len(v)))
print(open(out + '/summary.txt').read())
PY
rm -rf "$OUT"/p[0-9]

#!/bin/bash
# GPU box: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, kernel-trace only) and the SQ
# picture of the wide sweep kernels at cfg3 size.  usage: tools/pmc_traffic.sh OUTDIR (under gpurun_out/)
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/$1"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
  "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "GRBM_GUI_ACTIVE SQ_WAVES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o p -- python3 "$ROOT/tools/bench_sweep.py" P=4 B=256 T=40 D=256 H=256 n=2 bf16=1 K=25 > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'wide' not in k: continue
        k = k.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
lines = []
for k, cs in sorted(agg.items()):
    lines.append(k)
    for c, v in sorted(cs.items()):
        lines.append('   %-32s %18.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))
open(out + '/summary.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
# HBM bytes of one sweep_wide_bwd call = backward sweep + weight-gradient contraction + reduce
# (FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE x2 on gfx950 for 16-byte-per-lane coalesced reads)
def kib(k, c):
    v = agg.get(k, {}).get(c, [])
    return sum(v) / len(v) if v else 0.0
tab = {}
call = {'sweep_wide_bwd': [k for k in agg if 'wide_bwd' in k or 'wide_wgrad' in k or 'wide_reduce' in k],
        'sweep_wide_fwd': [k for k in agg if 'wide_fwd' in k]}
for tag, ks in call.items():
    rd = sum(kib(k, 'FETCH_SIZE') for k in ks) * 1024 * 2
    wr = sum(kib(k, 'WRITE_SIZE') for k in ks) * 1024
    tab[tag] = {'bytes_per_launch': rd + wr, 'read_bytes': rd, 'write_bytes': wr, 'kernels': ks,
                'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.sh, K=25 P=4 B=256 T=40, '
                          'FETCH_SIZE x2 per the gfx950 correction), profiles/r02_pmc_summary.txt'}
json.dump(tab, open(out + '/r02_pmc_traffic.json', 'w'), indent=1)
print(json.dumps(tab, indent=1))
PY

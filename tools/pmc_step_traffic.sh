#!/bin/bash
# GPU box: HBM-side traffic of one whole cfg3 step (every kernel; FETCH_SIZE and WRITE_SIZE in separate --pmc passes with
# --kernel-trace only, the guide's recipe; FETCH_SIZE x2 per the gfx950 correction -- exact for 16-byte-per-lane streaming
# reads, an upper bound for narrower ones) -> OUTDIR/<tag>_step_traffic.txt: bytes per step by kernel family and in total.
# usage: [CONFIG=cfg5] tools/pmc_step_traffic.sh OUTDIR TAG        (OUTDIR under gpurun_out/; CONFIG: bench.py --config, default cfg3)
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/$1"; TAG=$2; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for set in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc_$set" -o p -- python3 "$ROOT/bench.py" --config ${CONFIG:-cfg3} --eager --steps 2 --warmup 1 --no-cpu-baseline --no-extra > "$OUT/pmc_$set.log" 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections, re
out, tag = sys.argv[1], sys.argv[2]
fams = ('audio_loss', 'audio_up', 'audio_down', 'audio_fold', 'conv_wgrad', 'conv_down', 'conv_up', 'conv_fold', 'wide_fwd_kernel<false, 4', 'wide_fwd_kernel<false, 1', 'wide_bwd4', 'wide_wgrad7',
        'wide_bwd_kernel', 'wide_wgrad_kernel', 'nllb', 'colsum', 'contract', 'expand', 'wgrad_kernel', 'gemm_kernel', 'bn_', 'cat_head',
        'nan_to_zero', 'CatArray', 'elementwise', 'adam_flat', 'trans_wide', 'fold_slabs', 'wide_reduce')
def fam(n):
    for k in fams:
        if k in n:
            return k
    return re.sub(r'\(anonymous namespace\)::|void |at::native::', '', n)[:40]
tot = {}
steps = 1
for cset in ('FETCH_SIZE', 'WRITE_SIZE'):
    acc = collections.defaultdict(float)
    n_adam = 0
    for f in glob.glob('%s/pmc_%s/**/*counter_collection.csv' % (out, cset), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != cset:
                continue
            acc[fam(r['Kernel_Name'])] += float(r['Counter_Value'])
            if 'adam_flat' in r['Kernel_Name']:
                n_adam += 1
    steps = max(1, n_adam)
    tot[cset] = {k: v * 1024 * (2 if cset == 'FETCH_SIZE' else 1) / steps for k, v in acc.items()}
import os
lines = [('%s: HBM-side bytes per ' + os.environ.get('CONFIG', 'cfg3') + ' step (eager, one process, every kernel; %d steps in the trace; FETCH_SIZE x2)') % (tag, steps),
         '%-34s %10s %10s %10s' % ('kernel family', 'read GB', 'written GB', 'sum GB')]
keys = sorted(set(tot['FETCH_SIZE']) | set(tot['WRITE_SIZE']), key=lambda k: -(tot['FETCH_SIZE'].get(k, 0) + tot['WRITE_SIZE'].get(k, 0)))
for k in keys[:30]:
    r, w = tot['FETCH_SIZE'].get(k, 0) / 1e9, tot['WRITE_SIZE'].get(k, 0) / 1e9
    lines.append('%-34s %10.3f %10.3f %10.3f' % (k, r, w, r + w))
R, W = sum(tot['FETCH_SIZE'].values()) / 1e9, sum(tot['WRITE_SIZE'].values()) / 1e9
lines.append('%-34s %10.3f %10.3f %10.3f' % ('TOTAL', R, W, R + W))
open('%s/%s_step_traffic.txt' % (out, tag), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
rm -rf "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE"

#!/bin/bash
# GPU box: HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, kernel-trace only) of the wide sweep
# calls at cfg3 size, one record per (call, K) -> OUTDIR/r03_pmc_traffic.json (bench.py reads profiles/ copy).
# usage: tools/pmc_traffic3.sh OUTDIR [B] [T]      (OUTDIR under gpurun_out/)
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/$1"; mkdir -p "$OUT"
B="${2:-256}"; T="${3:-40}"
cd /tmp && export TMPDIR=/tmp
for cfg in "k25:K=25 rev=1 inv=0" "k1flt:K=1 rev=1 inv=0" "k1smt:K=1 rev=0 inv=1"; do
  name=${cfg%%:*}; args=${cfg#*:}
  for set in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/$name/$set" -o p -- python3 "$ROOT/tools/bench_sweep.py" P=4 B=$B T=$T D=256 H=256 n=2 bf16=1 $args > "$OUT/$name.$set.log" 2>&1
  done
done
python3 - "$OUT" "$B" "$T" <<'PY'
import csv, glob, sys, collections, json, re
out, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
tab, lines = {}, []
for name, ktag in (('k25', 'K=25,'), ('k1flt', 'K=1,rev'), ('k1smt', 'K=1,fwd,inv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (out, name), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r'(wide_\w+(<[^>]*>)?)', r['Kernel_Name'])
            if m:
                agg[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
    def kib(k, c):
        v = agg.get(k, {}).get(c, [])
        return sum(v) / len(v) if v else 0.0
    lines.append(name)
    for k in sorted(agg):
        lines.append('   %-40s FETCH_SIZE %14.0f KiB  WRITE_SIZE %14.0f KiB' % (k, kib(k, 'FETCH_SIZE'), kib(k, 'WRITE_SIZE')))
    for call, ks in (('sweep_wide_bwd', [k for k in agg if 'wide_bwd' in k or 'wide_wgrad' in k or 'wide_reduce' in k]),
                     ('sweep_wide_fwd', [k for k in agg if 'wide_fwd' in k])):
        rd = sum(kib(k, 'FETCH_SIZE') for k in ks) * 1024 * 2        # gfx950: FETCH_SIZE counts half of 16-byte-per-lane reads
        wr = sum(kib(k, 'WRITE_SIZE') for k in ks) * 1024
        tab['%s[P=4,%s' % (call, ktag)] = {
            'bytes_per_launch': rd + wr, 'read_bytes': rd, 'write_bytes': wr, 'kernels': sorted(ks),
            'shape': {'B': B, 'T': T, 'P': 4},
            'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic3.sh: tools/bench_sweep.py P=4 B=%d T=%d '
                      'bf16, FETCH_SIZE x2 per the gfx950 correction), profiles/r03_pmc_traffic_summary.txt' % (B, T)}
open(out + '/r03_pmc_traffic_summary.txt', 'w').write('\n'.join(lines) + '\n')
json.dump(tab, open(out + '/r03_pmc_traffic.json', 'w'), indent=1)
print('\n'.join(lines)); print(json.dumps({k: v['bytes_per_launch'] for k, v in tab.items()}, indent=1))
PY
rm -rf "$OUT"/k25 "$OUT"/k1flt "$OUT"/k1smt

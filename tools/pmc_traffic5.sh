#!/bin/bash
# GPU box: HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, kernel-trace only: the guide's HBM recipe) of
# the sweep calls at the BASELINE shapes -> OUTDIR/${ROUND}_pmc_traffic.json + ${ROUND}_pmc_traffic_summary.txt (bench.py reads
# the profiles/ copy; one record per (call, K, shape)).   usage: [ROUND=r06] tools/pmc_traffic5.sh OUTDIR      (OUTDIR under gpurun_out/)
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/$1"; mkdir -p "$OUT"
export ROUND="${ROUND:-r06}"
cd /tmp && export TMPDIR=/tmp
# shape name : bench_sweep arguments
SHAPES=("cfg3:P=4 B=256 T=40 D=256 H=256 bf16=1" "cfg5:P=3 B=512 T=128 D=256 H=256 bf16=1" "cfg2:P=3 B=1024 T=100 D=32 H=32 bf16=0")
for sh in "${SHAPES[@]}"; do
  sname=${sh%%:*}; sargs=${sh#*:}
  for cfg in "k25:K=25 rev=1 inv=0" "k1flt:K=1 rev=1 inv=0" "k1smt:K=1 rev=0 inv=1"; do
    name=${cfg%%:*}; args=${cfg#*:}
    for set in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/$sname/$name/$set" -o p -- python3 "$ROOT/tools/bench_sweep.py" $sargs n=2 $args > "$OUT/$sname.$name.$set.log" 2>&1
    done
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json, re, os
out = sys.argv[1]
rnd = os.environ.get('ROUND', 'r06')
shapes = {'cfg3': dict(B=256, T=40, P=4), 'cfg5': dict(B=512, T=128, P=3), 'cfg2': dict(B=1024, T=100, P=3)}
tab, lines = {}, []
for sname, shp in shapes.items():
    for name, ktag in (('k25', 'K=25,'), ('k1flt', 'K=1,rev'), ('k1smt', 'K=1,fwd,inv')):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob('%s/%s/%s/**/*counter_collection.csv' % (out, sname, name), recursive=True):
            for r in csv.DictReader(open(f)):
                m = re.search(r'((wide|sweep_mfma|sweep)_\w+(<[^>]*>)?)', r['Kernel_Name'])
                if m and 'pack' not in m.group(1):
                    agg[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
        def kib(k, c):
            v = agg.get(k, {}).get(c, [])
            return sum(v) / len(v) if v else 0.0
        lines.append('%s %s' % (sname, name))
        for k in sorted(agg):
            lines.append('   %-52s FETCH_SIZE %14.0f KiB  WRITE_SIZE %14.0f KiB' % (k[:52], kib(k, 'FETCH_SIZE'), kib(k, 'WRITE_SIZE')))
        fam = 'sweep_wide' if sname != 'cfg2' else 'sweep'
        for call, ks in ((fam + '_bwd', [k for k in agg if 'bwd' in k or 'wgrad' in k or 'reduce' in k]),
                         (fam + '_fwd', [k for k in agg if 'fwd' in k])):
            if not ks:
                continue
            rd = sum(kib(k, 'FETCH_SIZE') for k in ks) * 1024 * 2        # gfx950: FETCH_SIZE counts half of 16-byte-per-lane reads
            wr = sum(kib(k, 'WRITE_SIZE') for k in ks) * 1024
            tab['%s:%s[P=%d,%s' % (sname, call, shp['P'], ktag)] = {
                'key': '%s[P=%d,%s' % (call, shp['P'], ktag), 'bytes_per_launch': rd + wr, 'read_bytes': rd, 'write_bytes': wr,
                'kernels': sorted(ks), 'shape': shp,
                'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic5.sh: tools/bench_sweep.py at the %s '
                          'shape, FETCH_SIZE x2 per the gfx950 correction), profiles/%s_pmc_traffic_summary.txt' % (sname, rnd)}
open(out + '/%s_pmc_traffic_summary.txt' % rnd, 'w').write('\n'.join(lines) + '\n')
json.dump(tab, open(out + '/%s_pmc_traffic.json' % rnd, 'w'), indent=1)
print('\n'.join(lines)); print(json.dumps({k: round(v['bytes_per_launch'] / 1e9, 3) for k, v in tab.items()}, indent=1))
PY
for sh in cfg3 cfg5 cfg2; do rm -rf "$OUT/$sh"; done

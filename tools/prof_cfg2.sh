# GPU box: rocprofv3 kernel trace of the cfg2 bench (Spirals-synthetic, z = h = 32), summary -> gpurun_out/<tag>_kernel_stats.csv
tag=${1:-r02_cfg2}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${tag}_stats
python3 bench.py --config cfg2 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_err.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o ${tag} -- python3 bench.py --config cfg2 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/${tag}_bench_prof.json 2>> gpurun_out/${tag}_err.log
f=$(find gpurun_out/${tag}_stats -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv
find gpurun_out/${tag}_stats -name "*.csv" ! -name "*kernel_stats.csv" -delete
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total device ms', tot/1e6)
lib=sum(int(r['TotalDurationNs']) for r in rows if 'at::native' in r['Name'] or 'rocclr' in r['Name'])
print('at::native + rocclr share %.1f%%' % (100.0*lib/tot))
for r in rows[:32]:
    print('%-100s calls %5s total %8.3f ms avg %9.1f us %5s%%' % (r['Name'][:100], r['Calls'], int(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
python3 -c "import json; d=json.load(open('gpurun_out/${tag}_bench.json')); print(d['ms_per_step'], d['value'])"

# GPU box: rocprofv3 kernel trace of the default bench (cfg3), summary -> gpurun_out/<tag>_kernel_stats.csv
# (an unprofiled run first: it fills MIOpen's find cache, which would otherwise run inside the trace)
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${tag}_stats
time python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_err.log
grep real gpurun_out/${tag}_err.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o ${tag} -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/${tag}_bench_prof.json 2>> gpurun_out/${tag}_err.log
f=$(find gpurun_out/${tag}_stats -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv
find gpurun_out/${tag}_stats -name "*.csv" ! -name "*kernel_stats.csv" -delete
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total device ms', tot/1e6)
for r in rows[:30]:
    print('%-100s calls %5s total %8.3f ms avg %9.1f us %5s%%' % (r['Name'][:100], r['Calls'], int(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
python3 -c "import json; d=json.load(open('gpurun_out/${tag}_bench.json')); print(d['ms_per_step'], d['value'])"

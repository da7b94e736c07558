cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cfg4_stats
CFG1=0 B3=0 B4=${1:-256} python3 tools/bench_configs.py 2>/dev/null | tail -2
CFG1=0 B3=0 B4=${1:-256} rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cfg4_stats -o c4 -- python3 tools/bench_configs.py > /dev/null 2>&1
f=$(find gpurun_out/cfg4_stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total device ms', tot/1e6, '(4 steps)')
for r in rows[:22]:
    print('%-90s calls %5s total %8.3f ms avg %9.1f us %5s%%' % (r['Name'][:90], r['Calls'], int(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
find gpurun_out/cfg4_stats -name "*.csv" -delete

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cfg5_stats
export MIOPEN_DEBUG_CONV_GEMM=0 MIOPEN_DEBUG_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_NAIVE_CONV_WRW=0
CFG1=0 B3=0 B4=0 B5=${1:-128} python3 tools/bench_configs.py 2>/dev/null | tail -1 | cut -c1-700
CFG1=0 B3=0 B4=0 B5=${1:-128} rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cfg5_stats -o c5 -- python3 tools/bench_configs.py > /dev/null 2>&1
f=$(find gpurun_out/cfg5_stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total device ms', tot/1e6, '(3 steps)')
for r in rows[:24]:
    print('%-90s calls %5s total %8.3f ms avg %9.1f us %5s%%' % (r['Name'][:90], r['Calls'], int(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
find gpurun_out/cfg5_stats -name "*.csv" -delete

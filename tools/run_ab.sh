cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_wide
for a in "K=25" "K=1 inv=1"; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wide -o x -- python3 tools/bench_sweep.py P=4 B=256 T=40 D=256 H=256 n=3 bf16=1 $a > /dev/null 2>&1
f=$(find gpurun_out/prof_wide -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print('%-90s calls %s total %.3f ms avg %.1f us' % (r['Name'][:90], r['Calls'], int(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
done

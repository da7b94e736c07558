cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/cp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cp -o cp -- python3 tools/time_conv_parts.py > /dev/null 2>&1
f=$(find gpurun_out/cp -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-100s calls %5s avg %9.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3))
PY
find gpurun_out/cp -name "*.csv" -delete

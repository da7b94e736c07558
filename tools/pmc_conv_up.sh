#!/bin/bash
# GPU box: HBM traffic and SQ picture of the thin conv_up kernel (S = 32, 3 / 1 output channels)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/cup
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM" \
           "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/cup/p$i -o p -- python3 tools/time_conv_parts.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/cup/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv_up_kernel' not in k and 'conv_down_kernel' not in k: continue
        k = k.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in sorted(agg.items()):
    print(k)
    print('   ' + '  '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
PY
find gpurun_out/cup -name "*.csv" -delete

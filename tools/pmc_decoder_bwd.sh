#!/bin/bash
# GPU box: instruction / wait counters of the conv kernels of one ImageDecoder backward (tools/time_decoder_bwd.py)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p /tmp/cpp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/cpp/p$i -o p -- python3 $R/tools/time_decoder_bwd.py 2 > /dev/null 2>&1
done
python3 - "${1:-conv_wgrad_kernel}" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/cpp/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if sys.argv[1] not in k: continue
        k = k.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in sorted(agg.items()):
    print(k)
    print('   ' + '  '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
PY

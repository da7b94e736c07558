#!/bin/bash
# GPU box: same-box A/B of the tree's library against an A/B build (tools/build_variant.sh NAME ...): ms per cfg3 step, two rounds
# usage: ab_lib.sh NAME
for r in 1 2; do for v in tree $1; do
  lib=""; [ $v != tree ] && lib=$PWD/multimodal-dmm_amd/mdmm/lib/ab_$v/libmdmm_hip.so
  MDMM_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['config']['replay_matches_eager']['ok'], d['calls_ms_per_step'].get('conv_down[S=8]'), d['calls_ms_per_step'].get('conv_wgrad[S=32]'))"
done; done

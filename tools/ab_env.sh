#!/bin/bash
# GPU box: bench.py (cfg3, no extras) with an environment switch off / on, alternating.  usage: tools/ab_env.sh VAR OFF ON [steps]
for i in 1 2; do
for v in "$2" "$3"; do
env "$1=$v" python bench.py --no-cpu-baseline --no-extra --steps ${4:-30} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1=$v', d['ms_per_step'], d['config'].get('loss'))"
done; done

#!/bin/bash
# GPU box: same-box A/B through tools/bench_patched.py, two rounds; usage: ab_patched.sh <patch> [<patch> ...]
for r in 1 2; do for v in "$@"; do
  python tools/bench_patched.py $v --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['config']['replay_matches_eager']['ok'], d['config']['loss'])"
done; done

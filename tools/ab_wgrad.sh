#!/bin/bash
# GPU box: A/B of the wide weight-gradient launch (slices of the rows x XCD-turned workgroup ids) at cfg3 size
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/ab_wgrad"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in ${AB_CFGS:-"64 0" "42 0" "42 1" "40 1" "84 1"}; do
  set -- $cfg
  export MDMM_WGRAD_SPLIT=$1 MDMM_WGRAD_XCD=$2
  d="$OUT/s$1_x$2"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o p -- python3 "$ROOT/tools/bench_sweep.py" P=4 B=256 T=40 D=256 H=256 n=3 bf16=1 K=25 > "$d.log" 2>&1
  f=$(find "$d" -name '*kernel_stats.csv' | head -1)
  echo "split=$1 xcd=$2: $(grep -E 'wide_wgrad|wide_bwd_kernel' "$f" | awk -F, '{printf "%s calls=%s avg_us=%.1f | ", substr($1,1,40), $2, $4/1000}')"
done

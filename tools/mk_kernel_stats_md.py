"""rocprofv3 --stats CSV of a bench.py run -> profiles/<tag>_kernel_stats.md.
usage: python tools/mk_kernel_stats_md.py TAG ["title line"]   (reads gpurun_out/<tag>_stats/s_kernel_stats.csv and
gpurun_out/<tag>_prof_bench.json, written by tools/prof_bench.sh)"""
import csv, re, json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
tag = sys.argv[1]
title = sys.argv[2] if len(sys.argv) > 2 else 'cfg3 kernel statistics'
rows = list(csv.DictReader(open(os.path.join(ROOT, 'gpurun_out', '%s_stats' % tag, 's_kernel_stats.csv'))))
b = json.loads(open(os.path.join(ROOT, 'gpurun_out', '%s_prof_bench.json' % tag)).read().strip().splitlines()[-1])
steps = [int(r['Calls']) for r in rows if 'wide_bwd4' in r['Name']][0]
out = ['# %s: %s\n' % (tag, title),
       '`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra` (MI355X).',
       'bench.py under the profiler, same run: %.3f ms/step = %.0f sequences/s; %d steps in the trace (harness warm-up, graph replays, eager probe steps); %.0f launches per step.'
       % (b['ms_per_step'], b['value'], steps, sum(int(r['Calls']) for r in rows) / steps),
       "Kernels of different graph branches share the GPU (HBM-bound ones queue behind each other, anything queues behind the K = 25 backward sweep whose 256 workgroups take every CU), so a kernel's duration here includes its waiting.\n",
       '| kernel | calls per step | avg us | ms per step |', '|---|---|---|---|']
for r in rows[:60]:
    nm = re.sub(r'\(anonymous namespace\)::', '', r['Name'])[:110]
    out.append('| `%s` | %.1f | %.1f | %.3f |' % (nm, int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6 / steps))
open(os.path.join(ROOT, 'profiles', '%s_kernel_stats.md' % tag), 'w').write('\n'.join(out) + '\n')

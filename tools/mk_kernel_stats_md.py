import csv,re,json,sys
tag=sys.argv[1]
rows=list(csv.DictReader(open('/root/repo/gpurun_out/%s_stats/s_kernel_stats.csv'%tag)))
b=json.loads(open('/root/repo/gpurun_out/%s_prof_bench.json'%tag).read().strip().splitlines()[-1])
steps=[int(r['Calls']) for r in rows if 'wide_bwd4' in r['Name']][0]
out=['# %s: cfg3, last measured state of round 4 (BatchNorm adjoint on the conv kernels, conv_wgrad / conv_up rework)\n' % tag,
 '`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra` (MI355X).',
 'bench.py under the profiler, same run: %.3f ms/step = %.0f sequences/s; %d steps in the trace (harness warm-up, graph replays, eager probe steps); %.0f launches per step.' % (b['ms_per_step'], b['value'], steps, sum(int(r['Calls']) for r in rows)/steps),
 "Kernels of different graph branches share the GPU (HBM-bound ones queue behind each other, anything queues behind the K = 25 backward sweep whose 256 workgroups take every CU), so a kernel's duration here includes its waiting: compare with `r03c_kernel_stats.md` (before the decoder passes were batched: little overlap) for per-kernel times.\n",
 '| kernel | calls per step | avg us | ms per step |','|---|---|---|---|']
for r in rows[:60]:
    nm=re.sub(r'\(anonymous namespace\)::','',r['Name'])[:110]
    out.append('| `%s` | %.1f | %.1f | %.3f |' % (nm, int(r['Calls'])/steps, float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6/steps))
open('/root/repo/profiles/%s_kernel_stats.md'%tag,'w').write('\n'.join(out)+'\n')

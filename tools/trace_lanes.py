"""Per-queue listing of one graph-replayed step from a rocprofv3 kernel trace (the step picked as tools/timeline.py
picks it): every launch >= min_us with its queue, start, duration and workgroup count -- which chain each kernel is on.
usage: trace_lanes.py KERNEL_TRACE.csv [step] [min_us]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    wg = (int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])) // max(
        1, int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']))
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], wg, r['Queue_Id']))
ev.sort()
ends = [i for i, x in enumerate(ev) if 'FusedOptimizer' in x[2] or 'adam_flat_kernel' in x[2]]
gaps = [(a, b) for a, b in zip(ends, ends[1:]) if b - a > 100]
pick = int(sys.argv[2]) if len(sys.argv) > 2 else 5
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
lo, hi = gaps[pick][0] + 1, gaps[pick][1] + 1
step = ev[lo:hi]
t0 = step[0][0]
queues = sorted({x[4] for x in step}, key=lambda q: min(x[0] for x in step if x[4] == q))
col = {q: i for i, q in enumerate(queues)}
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n).replace('void ', '')
    n = re.sub(r'at::native::', '', n)
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n)
    return n.split('(')[0][:44]
print('queues in order of first use:', queues)
for s, e, n, wg, q in step:
    if (e - s) / 1e3 >= min_us:
        print('%7.3f %7.3f  q%-2d %s%-44s %5d wg' % ((s - t0) / 1e6, (e - s) / 1e6, col[q], '    ' * col[q], short(n), wg))

"""Per-millisecond picture of one step from a rocprofv3 kernel trace of bench.py: which kernels are on
the GPU in each 1-ms bin (device time inside the bin, workgroups per launch) and how many kernels run
side by side.  usage: timeline.py KERNEL_TRACE.csv [bin_ms] [step]
`step` indexes the gaps between optimizer launches: with bench.py --steps 3 --warmup 2 the first
three are the harness's eager warm-up steps, 3..7 graph replays (5 = the first timed one), the last
three the eager probe steps behind the timed region (-1, the default, is one of those: host-bound)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
bin_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
ev = []
for r in rows:
    name = r['Kernel_Name']
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    wg = 1
    try:
        wg = (int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])) // max(
            1, int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']))
    except (KeyError, ValueError):
        pass
    ev.append((s, e, name, wg))
ev.sort()
# a step ends with the fused Adam launch; take the last complete step
ends = [i for i, x in enumerate(ev) if 'FusedOptimizer' in x[2] or 'adam_flat_kernel' in x[2]]        # (the fused Adam launches only: _foreach_mul is a multi_tensor_apply too)
gaps = [(a, b) for a, b in zip(ends, ends[1:]) if b - a > 100]        # (an optimizer step is a few launches)
if not gaps:
    sys.exit('no full step between two optimizer launches in the trace')
pick = int(sys.argv[3]) if len(sys.argv) > 3 else -1
lo, hi = gaps[pick][0] + 1, gaps[pick][1] + 1
step = ev[lo:hi]
t0, t1 = step[0][0], max(x[1] for x in step)
print('step: %d launches, wall %.2f ms, device time %.2f ms' % (len(step), (t1 - t0) / 1e6, sum(e - s for s, e, _, _ in step) / 1e6))


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n).replace('void ', '')
    n = re.sub(r'at::native::', '', n)
    return n.split('(')[0][:46]


# how much of the wall no kernel at all is on the device (launch gaps of the chain), and the chain's longest idle gaps
cover, cur_s, cur_e, idle = 0, step[0][0], step[0][1], []
for s, e, n, _ in step[1:]:
    if s > cur_e:
        cover += cur_e - cur_s
        idle.append((s - cur_e, cur_e - t0, short(n)))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
cover += cur_e - cur_s
print('device busy (any kernel) %.2f ms of %.2f ms wall; %d idle gaps, total %.2f ms; gaps >= 20 us: %d (%.2f ms)' % (
    cover / 1e6, (t1 - t0) / 1e6, len(idle), sum(g for g, _, _ in idle) / 1e6,
    sum(1 for g, _, _ in idle if g >= 20000), sum(g for g, _, _ in idle if g >= 20000) / 1e6))
for g, at, n in sorted(idle, reverse=True)[:12]:
    print('   idle %.1f us at %.2f ms, before %s' % (g / 1e3, at / 1e6, n))

nb = int((t1 - t0) / 1e6 / bin_ms) + 1
for b in range(nb):
    a, z = t0 + b * bin_ms * 1e6, t0 + (b + 1) * bin_ms * 1e6
    acc = collections.defaultdict(lambda: [0.0, 0])
    for s, e, n, wg in step:
        o = min(e, z) - max(s, a)
        if o > 0:
            k = short(n)
            acc[k][0] += o
            acc[k][1] = max(acc[k][1], wg)
    tot = sum(v[0] for v in acc.values())
    top = sorted(acc.items(), key=lambda kv: -kv[1][0])[:4]
    print('%5.1f ms  x%.2f  %s' % (b * bin_ms, tot / (bin_ms * 1e6),
                                   ' | '.join('%s %.2f (%d wg)' % (k, v[0] / 1e6, v[1]) for k, v in top)))

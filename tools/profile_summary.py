"""usage: python tools/profile_summary.py TAG "title"  -- gpurun_out/TAG_{bench.json,stats/} -> profiles/TAG_*"""
import csv, json, os, re, shutil, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
tag, title = sys.argv[1], sys.argv[2]
src = os.path.join(R, 'gpurun_out')
bench = json.loads(open(os.path.join(src, tag + '_bench.json')).read().strip().splitlines()[-1])
shutil.copy(os.path.join(src, tag + '_bench.json'), os.path.join(R, 'profiles', tag + '_bench.json'))
stats = None
for d, _, fs in os.walk(os.path.join(src, tag + '_stats')):
    for f in fs:
        if f.endswith('kernel_stats.csv'):
            stats = os.path.join(d, f)
shutil.copy(stats, os.path.join(R, 'profiles', tag + '_kernel_stats.csv'))
rows = list(csv.DictReader(open(stats)))
rf, cb = bench['roofline'], bench.get('cpu_baseline')
out = ['# ' + title, '',
       '`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py ...` (tools/prof_bench.sh / prof_cfg2.sh; MI355X; workload: %s).' % bench['config']['workload'][:60],
       'bench.py on the same box, same commit (`%s_bench.json`): %.3f ms/step = %.0f sequences/s; dominant kernel `%s`'
       % (tag, bench['ms_per_step'], bench['value'], rf['kernel']),
       '%.3f ms/launch by HIP events -> %.1f TFLOP/s = %.3f of the %.1f TFLOP/s peak (%s operands).'
       % (rf['launch_ms'], rf['achieved'], rf['frac'], rf['peak'], rf.get('operands', '')),
       ('CPU oracle on the same box: %.3f sequences/s.' % cb['value']) if cb else 'CPU baseline: see the full bench line.', '',
       '| kernel | calls | total ms | avg us | % |', '|---|---|---|---|---|']
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n); n = n.replace('at::native::', '', 1)
    return n[:96]
for r in rows[:40]:
    out.append('| `%s` | %s | %.2f | %.1f | %s |' % (short(r['Name']), r['Calls'], int(r['TotalDurationNs']) / 1e6,
                                                     float(r['AverageNs']) / 1e3, r['Percentage']))
open(os.path.join(R, 'profiles', tag + '_kernel_stats.md'), 'w').write('\n'.join(out) + '\n')
print('\n'.join(out[:16]))

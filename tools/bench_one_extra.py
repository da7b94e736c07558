"""GPU box: one of bench.py's ride-along configurations alone (cfg3_f32 has no --config of its own).
usage: python tools/bench_one_extra.py cfg3_f32|cfg4_f32|cfg5_f32|cfg2|cfg4 [steps] [warmup]; SPANS=1: the largest library calls"""
import argparse, json, os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
import bench
name = sys.argv[1]
cfg = {'cfg3_f32': bench.Cfg3F32, 'cfg4_f32': bench.Cfg4F32, 'cfg5_f32': bench.Cfg5F32, 'cfg2': bench.Cfg2, 'cfg4': bench.Cfg4}[name]
a = argparse.Namespace(gpus=1, steps=int(sys.argv[2]) if len(sys.argv) > 2 else 3, warmup=int(sys.argv[3]) if len(sys.argv) > 3 else 2,
                       batch=0, config='cfg3',
                       no_cpu_baseline=True, no_extra=True, eager=False)
bench.ANNEAL = os.environ.get('ANNEAL') == '1'
r = bench.run(cfg, a, 1, 0, torch.device('cuda:0'), graph=not name.endswith('_f32') and os.environ.get('EAGER') != '1')
print(json.dumps({k: r[k] for k in ('ms_per_step', 'value')}), r['config'].get('loss'))
if os.environ.get('SPANS') == '1':
    calls = r.get('calls_ms_per_step') or {}
    for k, v in sorted(calls.items(), key=lambda kv: -kv[1])[:25]:
        print('%10.3f ms  %s' % (v, k))
    print('library calls summed: %.1f ms of %.1f' % (sum(calls.values()), r['ms_per_step']))

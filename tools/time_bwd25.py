"""GPU box: time of the K=25 sweep kernels in the cfg2 step (eager, HIP events)."""
import os, sys, json, subprocess
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
out = subprocess.run([sys.executable, os.path.join(R, 'bench.py'), '--no-cpu-baseline', '--steps', '5', '--warmup', '2'],
                     capture_output=True, text=True).stdout
d = json.loads(out.strip().splitlines()[-1])
k = d['calls_ms_per_step']
print(os.environ.get('TAG', ''), 'ms/step', d['ms_per_step'], 'bwd25', k.get('sweep_bwd[P=3,K=25,rev]'), 'fwd25', k.get('sweep_fwd[P=3,K=25,rev]'))

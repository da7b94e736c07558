cd $GRAFT_REPO_ROOT
python tools/repro_replay_op.py cfg4 > gpurun_out/r04d_repro_cfg4.txt 2>&1
cat gpurun_out/r04d_repro_cfg4.txt
python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "step or zfilter or forward" 2>&1 | tail -5
for v in 1 0 1 0; do MDMM_JOINT_DECODE=$v python bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('JOINT=$v', d['ms_per_step'], d['config']['loss'], d['config']['replay_matches_eager'])"; done

for lib in "" ab_fwdbound.so; do
  if [ -n "$lib" ]; then export MDMM_LIB=$PWD/multimodal-dmm_amd/mdmm/lib/$lib; else unset MDMM_LIB; fi
  TAG=${lib:-base} python tools/bench_sweep.py K=1 inv=1 rev=0 n=10 2>&1 | grep sweep_f
  TAG=${lib:-base} python tools/bench_sweep.py K=1 inv=0 rev=1 n=10 2>&1 | grep sweep_f
done

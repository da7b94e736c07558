#!/bin/bash
# diagnostic build of libmdmm_hip.so with in-kernel cycle stamps (-DWIDE_STAMPS) -> mdmm/lib/ab_stamps/
set -e
cd "$(dirname "$0")/../multimodal-dmm_amd/csrc"
make -s
d=../mdmm/lib/ab_stamps; rm -rf $d; mkdir -p $d/.b
for f in *.hip; do cp ../mdmm/lib/.build/${f%.hip}.o $d/.b/; done
for f in sweep_wide sweep_wide_bwd4; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DWIDE_STAMPS -c -o $d/.b/$f.o $f.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $d/libmdmm_hip.so $d/.b/*.o
rm -rf $d/.b

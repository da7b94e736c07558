#!/bin/bash
# A/B build: tools/build_variant.sh NAME FILE.hip "-DFLAG ..." -> mdmm/lib/ab_NAME/libmdmm_hip.so (only FILE recompiled)
set -e
name=$1; file=$2; flags=$3
cd "$(dirname "$0")/../multimodal-dmm_amd/csrc"
make -s -j8
d=../mdmm/lib/ab_$name; rm -rf $d; mkdir -p $d/.b
for f in *.hip; do cp ../mdmm/lib/.build/${f%.hip}.o $d/.b/; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -c -o $d/.b/${file%.hip}.o $file
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $d/libmdmm_hip.so $d/.b/*.o
rm -rf $d/.b

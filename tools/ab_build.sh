#!/bin/bash
# usage: tools/ab_build.sh NAME [extra hipcc flags]  -> multimodal-dmm_amd/mdmm/lib/ab_NAME.so from the current csrc (A/B kernel experiments)
cd "$(dirname "$0")/../multimodal-dmm_amd/csrc" && n=$1 && shift && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -shared -o ../mdmm/lib/ab_$n.so *.hip

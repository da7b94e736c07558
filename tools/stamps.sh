#!/bin/bash
# GPU box: build the -DMDMM_STAMPS diagnostic library and print the phase table
cd "${GRAFT_REPO_ROOT:-/root/repo}"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMDMM_STAMPS -shared -o gpurun_out/libmdmm_stamps.so multimodal-dmm_amd/csrc/*.hip || exit 1
python3 tools/stamps.py "$@"

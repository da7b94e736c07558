#!/bin/bash
# GPU box: workgroups per CU of the conv up / down kernels (tools/time_conv_parts.py, N = 20480)
for v in 2 3 4 6; do echo "up per CU $v: $(MDMM_CONV_UP_PER_CU=$v python tools/time_conv_parts.py N=20480 2>&1 | grep -E 'Deconv|Conv' | sed -E 's/colsum\[[0-9]+\] [0-9.]+ ms//; s/conv_wgrad\[S=[0-9]+\] [0-9.]+ ms//' | tr '\n' '|')"; done
for v in 1 2 3 4; do echo "down per CU $v: $(MDMM_CONV_DOWN_PER_CU=$v python tools/time_conv_parts.py N=20480 2>&1 | grep -E 'Deconv|Conv' | sed -E 's/colsum\[[0-9]+\] [0-9.]+ ms//; s/conv_wgrad\[S=[0-9]+\] [0-9.]+ ms//' | tr '\n' '|')"; done

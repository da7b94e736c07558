#!/bin/bash
# GPU box: conv weight-gradient kernels at other grid sizes (tools/time_conv_parts.py)
for n in 20480 10240; do
for g in 768 512 384 256 192 128; do echo "N=$n S=8 grid $g: $(MDMM_CONV_WGRAD_GRID8=$g python tools/time_conv_parts.py N=$n 2>&1 | grep 'Deconv 64->32' | sed 's/.*conv_wgrad/conv_wgrad/')"; done
for g in 512 384 256; do echo "N=$n S>=16 grid $g: $(MDMM_CONV_WGRAD_GRID=$g python tools/time_conv_parts.py N=$n 2>&1 | grep -E 'Deconv 32->16|Deconv 16->3' | sed 's/.*conv_wgrad/conv_wgrad/' | tr '\n' ' ')"; done
done

#!/bin/bash
# Ablation builds of the fused INT8 kernel (conv_i8_fused.hip.h, DWPW_ABL bits), made before the GPU call:
#   for a in 1 2 4; do make -C superpoint-stereo-visual-odometry_amd -j8 BUILD=build_abl$a OUT=variants/abl$a EXTRA=-DDWPW_ABL=$a; done
echo "== full"; python tools/perop_int8.py mbv1 2>&1 | grep -E "conv:3|conv:5|conv:7|conv:9 |conv:11|conv:13|net"
for a in 1 2 4; do echo "== ABL $a"; PEROP_LIB=superpoint-stereo-visual-odometry_amd/variants/abl$a/libspvo.so python tools/perop_int8.py mbv1 2>&1 | grep -E "conv:3|conv:5|conv:7|conv:9 |conv:11|conv:13|net"; done
python -m pytest tests/test_gpu_network.py -m gpu -x -q -k "int8" 2>&1 | tail -2

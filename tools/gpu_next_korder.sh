#!/bin/bash
# Prepared for the next GPU session (not yet run): A/B of the channel-slice-major ("tap-minor") K order of the split-precision
# forward kernels against the tap-major default -- time per launch (tools/ab_x3.py prints the max difference against the
# first configuration; the two orders differ in the last bits only) and the memory-side traffic of both on the member-batched
# res-block launch (profiles/r02_i_x3w_pmc.txt has the tap-major figure: 9.1x the activation bytes).
set -u
O=gpurun_out/korder
mkdir -p $O
export TMPDIR=/tmp
CG_TEST_EXPERIMENTAL=1 timeout -k 5 60 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -k "tile_configuration or wide_tile" < /dev/null > $O/tiles.log 2>&1; tail -2 $O/tiles.log
EXTRA_SHAPES="res16,16,64,64,256,256,3,1,1,0;res32,32,64,64,256,256,3,1,1,0;up1,16,64,64,256,128,3,1,1,1;dec128,16,128,128,128,128,3,1,1,0;d3,16,64,64,128,256,4,2,1,0;c64,16,256,256,64,64,3,1,1,0" \
  timeout -k 5 120 python tools/ab_x3.py 16,22,13,23,1,24,2,25,26,27 12 < /dev/null > $O/ab_korder.txt 2>&1; cat $O/ab_korder.txt | cut -c1-250
timeout -k 5 150 bash tools/pmc_x3w.sh korder/pmc_cfg22 22 < /dev/null > $O/pmc22.log 2>&1; tail -4 $O/pmc22.log | cut -c1-300
# weight-gradient kernel: is it bound by operand traffic? (128x128 -> 256x128 cut the operand bytes per MFMA by 25 % and the time by 20-25 %)
timeout -k 5 150 bash tools/pmc_kernel.sh korder/pmc_wgrad128 conv_wgrad_x3t tools/prof_wgrad_x3.py 0 < /dev/null > $O/pmc_wgrad128.log 2>&1; tail -4 $O/pmc_wgrad128.log | cut -c1-300
timeout -k 5 150 bash tools/pmc_kernel.sh korder/pmc_wgrad256 conv_wgrad_x3t tools/prof_wgrad_x3.py 1 < /dev/null > $O/pmc_wgrad256.log 2>&1; tail -4 $O/pmc_wgrad256.log | cut -c1-300
# the experimental 256x256 LDS-DMA weight-gradient tile against the two register-staged ones
timeout -k 5 100 python tools/ab_thin.py wgrad wide < /dev/null > $O/ab_wgrad_wide.txt 2>&1; cat $O/ab_wgrad_wide.txt | cut -c1-250

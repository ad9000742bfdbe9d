#!/bin/bash
# round-2 GPU call 5: forward-error diagnostics, staggered wide kernel A/B
set -u
O=gpurun_out/r02_e
mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/diag_forward.py male2female_council_folder.yaml 128 2 > $O/diag_fwd_m2f.txt 2>&1
timeout 200 python tools/diag_forward.py anime2face_council_folder.yaml 128 1 > $O/diag_fwd_anime.txt 2>&1
cat $O/diag_fwd_m2f.txt | cut -c1-140 | tail -32
timeout 120 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider -k "every_tile_configuration" > $O/pytest_tiles.log 2>&1; tail -3 $O/pytest_tiles.log
EXTRA_SHAPES="res16,16,64,64,256,256,3,1,1,0;up1,16,64,64,256,128,3,1,1,1;dec128,16,128,128,128,128,3,1,1,0;d3,32,64,64,128,256,4,2,1,0;dc3,64,128,128,128,256,4,2,1,0;res32,32,64,64,256,256,3,1,1,0" timeout 300 python tools/ab_x3.py 1,13,16,18,17,19 0 > $O/ab_x3.txt 2>&1
tail -9 $O/ab_x3.txt

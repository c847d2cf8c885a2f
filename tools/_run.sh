mkdir -p gpurun_out/r03a
{ timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "block_rotation_groupings and 21" 2>&1 | tail -3; GS="1 4 21" bash tools/blkg_ab.sh;  BATCH=4096 GS="1 4 21" bash tools/blkg_ab.sh; } 2>&1 | tee gpurun_out/r03a/blk_v4.txt

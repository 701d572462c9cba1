#!/bin/bash
O=$PWD/gpurun_out/r05; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -k "image or img or bev" 2>&1 | tail -2
BENCH_ARGS="--refresh-every 0 --image --witness 256" bash tools/exp_env_ab.sh 2 "new:" "base:HOPE_AMD_LIB=$PWD/hope_amd/libhope_env_base.so" > $O/ab_image_list_wave.txt 2>&1; cat $O/ab_image_list_wave.txt
bash tools/_tl_img.sh > /dev/null 2>&1; grep -E "k_bev|k_kinematics" $O/timeline_image.txt | tail -12

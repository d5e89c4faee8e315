#!/bin/bash
# Same stage order and variables as the reference's scripts/exp_shapley.sh:2-9.  Pass --synthetic through
# EXTRA when no datasets/checkpoints are present; prefix with "torchrun --nproc-per-node N" via
# LAUNCH to shard every stage over N GPUs.
model="pointnet";
dataset="shapenet";
device_id=0;
LAUNCH=${LAUNCH:-python}
EXTRA=${EXTRA:-}
$LAUNCH final_shapley_value.py --model=$model --dataset=$dataset --device_id=$device_id $EXTRA
$LAUNCH final_trans_center_enum_all.py --model=$model --dataset=$dataset --device_id=$device_id $EXTRA
$LAUNCH final_rotate_center_enum_all.py --model=$model --dataset=$dataset --device_id=$device_id $EXTRA
$LAUNCH final_scale_center_enum_all.py --model=$model --dataset=$dataset --device_id=$device_id $EXTRA
$LAUNCH final_smoothness_center_enum_all.py --model=$model --dataset=$dataset --device_id=$device_id $EXTRA

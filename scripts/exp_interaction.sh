#!/bin/bash
# Stages 2-3 of the reference's scripts/exp_interaction.sh:2-7.  Stage 1 (final_gen_pair.py, host
# NumPy sampling of pairs/contexts) is a "next" row: its outputs
# (region_pair_list.npy, ratio*_context_list.npy, <mode>_adv/transform_params.npy) must exist.
model="pointnet";
dataset="shapenet";
device_id=0;
LAUNCH=${LAUNCH:-python}
EXTRA=${EXTRA:-}
$LAUNCH final_point_binary_interaction_logits.py --model=$model --dataset=$dataset --device_id=$device_id $EXTRA
$LAUNCH final_cal_interactions.py --model=$model --dataset=$dataset --device_id=$device_id $EXTRA

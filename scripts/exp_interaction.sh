#!/bin/bash
# Same stage order and variables as the reference's scripts/exp_interaction.sh:2-7 (it needs the
# <mode>_all/ artefacts of scripts/exp_shapley.sh, as in the reference).
model="pointnet";
dataset="shapenet";
device_id=0;
LAUNCH=${LAUNCH:-python}
EXTRA=${EXTRA:-}
python final_gen_pair.py --model=$model --dataset=$dataset --device_id=$device_id $EXTRA
$LAUNCH final_point_binary_interaction_logits.py --model=$model --dataset=$dataset --device_id=$device_id $EXTRA
$LAUNCH final_cal_interactions.py --model=$model --dataset=$dataset --device_id=$device_id $EXTRA

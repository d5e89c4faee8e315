#!/bin/bash
# Same stage order and variables as the reference's scripts/exp_interaction.sh:2-7 (it needs the
# <mode>_all/ artefacts of scripts/exp_shapley.sh, as in the reference).
model="pointnet";
dataset="shapenet";
device_id=0;
LAUNCH=${LAUNCH:-python}
EXTRA=${EXTRA:-}
# final_gen_pair.py draws pairs / contexts from ONE host generator that runs on from cloud to cloud: under a multi-rank
# LAUNCH rank 0 does its work and the others wait at its barrier, so that every stage runs under the same launcher.
$LAUNCH final_gen_pair.py --model=$model --dataset=$dataset --device_id=$device_id $EXTRA
$LAUNCH final_point_binary_interaction_logits.py --model=$model --dataset=$dataset --device_id=$device_id $EXTRA
$LAUNCH final_cal_interactions.py --model=$model --dataset=$dataset --device_id=$device_id $EXTRA

#!/usr/bin/env python3
"""Drop-in entry point of the pre-stage final_save_fps.py:57-82: FPS region centres of the 30 clouds
-> fps_<dataset>_1024_32_index_final30.npy."""
from interpret_quality_amd.shapley_stage import build_parser, finish_args, save_fps

if __name__ == "__main__":
    args = build_parser("pointnet").parse_args()
    finish_args(args)
    print(save_fps(args))

#!/usr/bin/env python3
"""Drop-in entry point of the pre-stage final_save_fps.py:57-82: FPS region centres of the 30 clouds
-> fps_<dataset>_1024_32_index_final30.npy."""
from interpret_quality_amd.shapley_stage import build_parser, finish_args, save_fps

# the reference's module-level names, importable from here as from the reference's script
from interpret_quality_amd.shapley_stage import (farthest_point_sample, save_fps)  # noqa: F401,E402

if __name__ == "__main__":
    args = build_parser("pointnet").parse_args()
    finish_args(args)
    print(save_fps(args))

#!/usr/bin/env python3
"""Drop-in entry point with the reference's script name and flags (final_rotate_center_enum_all.py:70-98).
Thin driver: all logic lives in interpret_quality_amd/, all arithmetic in libiq_hip.so."""
from interpret_quality_amd.pose_sweep import main_rotate

# the reference's module-level names, importable from here as from the reference's script
from interpret_quality_amd.pose_sweep import (rotate_xyz, generate_rotate_angle, print_rotate_info, save_rotate_info, ANGLE_THRESHOLD, NUM_GRID_ENUM_ROTATE)  # noqa: F401,E402

if __name__ == "__main__":
    main_rotate()

#!/usr/bin/env python3
"""Drop-in entry point with the reference's script name and flags (final_scale_center_enum_all.py:44-73).
Thin driver: all logic lives in interpret_quality_amd/, all arithmetic in libiq_hip.so."""
from interpret_quality_amd.pose_sweep import main_scale

# the reference's module-level names, importable from here as from the reference's script
from interpret_quality_amd.pose_sweep import (scale_pc, generate_scale, print_scale_info, save_scale_info, SCALE_UPPER, SCALE_LOWER, NUM_GRID_ENUM_SCALE)  # noqa: F401,E402

if __name__ == "__main__":
    main_scale()

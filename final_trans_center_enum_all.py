#!/usr/bin/env python3
"""Drop-in entry point with the reference's script name and flags (final_trans_center_enum_all.py:56-84).
Thin driver: all logic lives in interpret_quality_amd/, all arithmetic in libiq_hip.so."""
from interpret_quality_amd.pose_sweep import main_trans

# the reference's module-level names, importable from here as from the reference's script
from interpret_quality_amd.pose_sweep import (translate_pc, generate_trans_vector, print_trans_info, save_trans_info, TRANS_DIST_THRESHOLD, NUM_GRID_ENUM_TRANS)  # noqa: F401,E402

if __name__ == "__main__":
    main_trans()

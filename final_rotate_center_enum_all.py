#!/usr/bin/env python3
"""Drop-in entry point with the reference's script name and flags (final_rotate_center_enum_all.py:70-98).
Thin driver: all logic lives in interpret_quality_amd/, all arithmetic in libiq_hip.so."""
from interpret_quality_amd.pose_sweep import main_rotate

if __name__ == "__main__":
    main_rotate()

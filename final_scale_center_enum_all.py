#!/usr/bin/env python3
"""Drop-in entry point with the reference's script name and flags (final_scale_center_enum_all.py:44-73).
Thin driver: all logic lives in interpret_quality_amd/, all arithmetic in libiq_hip.so."""
from interpret_quality_amd.pose_sweep import main_scale

if __name__ == "__main__":
    main_scale()

#!/usr/bin/env python3
"""Drop-in entry point with the reference's script name and flags (final_smoothness_center_enum_all.py:393-424).
Thin driver: all logic lives in interpret_quality_amd/, all arithmetic in libiq_hip.so."""
from interpret_quality_amd.smoothness import main

if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Drop-in entry point with the reference's script name and flags (final_shapley_value.py:177-211).
Thin driver: all logic lives in interpret_quality_amd/, all arithmetic in libiq_hip.so."""
from interpret_quality_amd.shapley_stage import main

# the reference's module-level names, importable from here as from the reference's script
from interpret_quality_amd.shapley_stage import (cal_region_id, cal_norm_factor, generate_all_orders, mask_data, save_shapley, shap_sampling, test, main)  # noqa: F401,E402

if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Drop-in entry point with the reference's script name and flags (final_cal_interactions.py:102-144).
Thin driver: all logic lives in interpret_quality_amd/, all arithmetic in libiq_hip.so."""
from interpret_quality_amd.interaction import main_cal

# the reference's module-level names, importable from here as from the reference's script
from interpret_quality_amd.interaction import (compute_order_interaction, cal_interaction_all_orders, cal_interaction)  # noqa: F401,E402

if __name__ == "__main__":
    main_cal()

#!/usr/bin/env python3
"""Drop-in entry point with the reference's script name and flags (final_point_binary_interaction_logits.py:140-180).
Thin driver: all logic lives in interpret_quality_amd/, all arithmetic in libiq_hip.so."""
from interpret_quality_amd.interaction import main_logits

# the reference's module-level names, importable from here as from the reference's script
from interpret_quality_amd.interaction import (compute_order_interaction_logits, save_logits_all_orders, save_logits)  # noqa: F401,E402

if __name__ == "__main__":
    main_logits()

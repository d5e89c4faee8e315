#!/usr/bin/env python3
"""Drop-in entry point with the reference's script name and flags (final_point_binary_interaction_logits.py:140-180).
Thin driver: all logic lives in interpret_quality_amd/, all arithmetic in libiq_hip.so."""
from interpret_quality_amd.interaction import main_logits

if __name__ == "__main__":
    main_logits()

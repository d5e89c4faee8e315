#!/usr/bin/env python3
"""Drop-in entry point with the reference's script name and flags (final_gen_pair.py:323-374).
Thin driver: all logic lives in interpret_quality_amd/, all arithmetic in libiq_hip.so."""
from interpret_quality_amd.gen_pair import main

# the reference's module-level names, importable from here as from the reference's script
from interpret_quality_amd.gen_pair import (gen_context, save_context, gen_pred_label, save_pred_label, gen_pair_single_region, save_pair_single_region, check_adv_success, gen_pair_random, save_pair_random)  # noqa: F401,E402

if __name__ == "__main__":
    main()

"""Reference import path ``tools.final_common`` (tools/final_common.py:11-174) -> interpret_quality_amd."""
from interpret_quality_amd.final_common import (cal_reward, get_reward, mask_data_batch,  # noqa: F401
                                                shap_sampling_all_regions_batch)
from interpret_quality_amd.pose_sweep import test  # noqa: F401

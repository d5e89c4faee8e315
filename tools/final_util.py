"""Reference import path ``tools.final_util`` (tools/final_util.py) -> interpret_quality_amd.final_util.
Training-only helpers of the reference (cal_loss, rot_angle_axis) are out of scope and not provided."""
from interpret_quality_amd.final_util import *  # noqa: F401,F403
from interpret_quality_amd.final_util import (IOStream, ball_query, cal_rank, get_folder_name_list, load_model, mkdir,  # noqa: F401
                                              set_interaction_batch_size, set_model_args, set_random, set_shapley_batch_size,
                                              square_distance, square_distance_np)

"""Drop-in for the reference's final_data_shapley.py (same class / function names); see
interpret_quality_amd/data_shapley.py."""
from interpret_quality_amd.data_shapley import (ModelNet_Loader_Shapley_test, ShapeNetDataset_Shapley_test,  # noqa: F401
                                                farthest_point_sample_np, make_dataset_modelnet10)

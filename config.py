"""Batch-size knobs of the Shapley / interaction drivers (same keys and meaning as the reference's
config.py:2-17): permutations (x (R+1) clouds) per forward batch for the Shapley sweeps, contexts
(x 4 clouds) per forward batch for the interaction logits.  The reference sized them for 24 GB
GPUs; on the HIP path they only bound how many materialised clouds the non-PointNet models see per
launch - PointNet never materialises clouds, so for it the knob only keeps the reference's
``num_samples // batch`` semantics."""
CONFIG = {
    "shapley_batch_size": {
        "pointnet2": 5,
        "pointnet": 50,
        "dgcnn": 5,
        "gcnn": 10,
        "pointconv": 20
    },
    "interaction_batch_size": {
        "pointnet2": 25,
        "pointnet": 100,
        "dgcnn": 25,
        "gcnn": 50,
        "pointconv": 100
    }
}

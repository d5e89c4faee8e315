"""Batch-size knobs of the Shapley / interaction drivers (same keys and meaning as the reference's
config.py:2-17): permutations (x (R+1) clouds) per forward batch for the Shapley sweeps, contexts
(x 4 clouds) per forward batch for the interaction logits.  The reference sized them for 24 GB
GPUs; on the HIP path they only bound how many materialised clouds the non-PointNet models see per
launch - PointNet never materialises clouds, so for it the knob only keeps the reference's
``num_samples // batch`` semantics."""
# "strict_batch_cap" (additive key, also IQ_STRICT_BATCH=1): False = the knobs are a FLOOR - launches take as many coalitions as
# there are (rows are independent in eval mode, so the results do not depend on the batching; the HIP kernels want tens of
# thousands of workgroups).  True = the knobs are a CAP as in the reference: no launch sees more than
# shapley_batch_size x (R+1) resp. interaction_batch_size x 4 coalitions, so they bound device memory again.
CONFIG = {
    "strict_batch_cap": False,
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

"""Coalition counters of this process (the unit BASELINE.json's metric counts: one coalition = one masked forward pass of
one cloud).  ``coalitions``: rows in the reference's row order - what tools/final_common.py:88-93 and
final_point_binary_interaction_logits.py:45-56 would push through the network; ``evaluated``: the distinct clouds the device
ran (equal sets are equal clouds, final_common.distinct_coalitions).  tools/sweep.py reads the difference around each unit."""
COUNTS = {"coalitions": 0, "evaluated": 0}


def add(coalitions, evaluated=None):
    COUNTS["coalitions"] += int(coalitions)
    COUNTS["evaluated"] += int(coalitions if evaluated is None else evaluated)


def snapshot():
    return dict(COUNTS)


def since(before):
    return {k: COUNTS[k] - before[k] for k in COUNTS}

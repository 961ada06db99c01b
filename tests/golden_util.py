import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_cases(fname):
    """npz with keys '<case>__<field>' -> {case: {field: array}}"""
    z = np.load(os.path.join(GOLDEN, fname), allow_pickle=False)
    cases = {}
    for key in z.files:
        case, field = key.split("__", 1)
        cases.setdefault(case, {})[field] = z[key]
    return cases

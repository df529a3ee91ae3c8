"""Seed helper used for identical-seed parity (reference: engine/tools/utils.py:26-32)."""
import random

import numpy as np
import torch


def set_random_seed(seed):
    """Seed python, numpy and torch (CPU + every GPU) generators."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def makedirs(path, verbose=False):
    import os
    os.makedirs(path, exist_ok=True)

"""The pieces of cosyvoice/utils/common.py callers touch directly."""
import random

import numpy as np
import torch


def set_all_random_seed(seed):
    """utils/common.py:153-157."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)

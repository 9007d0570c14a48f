"""The two glorot helpers the PHC stack borrows (reference: phc/quaternion/inits.py:12-17)."""
import math

import torch


def glorot_normal(tensor: torch.Tensor):
    return torch.nn.init.xavier_normal_(tensor, gain=math.sqrt(2))


def glorot_uniform(tensor: torch.Tensor):
    return torch.nn.init.xavier_uniform_(tensor, gain=math.sqrt(2))

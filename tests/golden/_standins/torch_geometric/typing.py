from typing import Optional, Tuple
import torch

Adj = torch.Tensor
Size = Optional[Tuple[int, int]]
OptTensor = Optional[torch.Tensor]

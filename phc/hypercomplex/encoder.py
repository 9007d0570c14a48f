"""Atom / bond feature encoders (reference: phc/hypercomplex/encoder.py):
categorical (or dense) input features -> [*, phm_dim, out_dim]."""
from typing import Union

import torch

from phc.quaternion.encoder import IntegerEncoder
from phc_gnn_amd import ops as _engine


def _make_component_encoder(input_dims, out_dim, combine):
    if isinstance(input_dims, list):
        return IntegerEncoder(input_dims=input_dims, out_dim=out_dim, combine=combine)
    if isinstance(input_dims, int):
        return torch.nn.Linear(in_features=input_dims, out_features=out_dim, bias=True)
    raise ValueError(f"`input_dims` must be an int or a list, got {type(input_dims)}")


class PHMEncoder(torch.nn.Module):
    """phm_dim independent encoders, one per hypercomplex component (ref :7-41)."""

    def __init__(self, out_dim: int, input_dims: Union[list, int], phm_dim: int, combine: str = "sum"):
        super().__init__()
        self.input_dims = input_dims
        self.out_dim = out_dim
        self.phm_dim = phm_dim
        self.combine = combine
        self.encoders = torch.nn.ModuleList([_make_component_encoder(input_dims, out_dim, combine)
                                             for _ in range(phm_dim)])
        self.reset_parameters()

    def reset_parameters(self):
        for enc in self.encoders:
            enc.reset_parameters()

    @property
    def engine_tables(self):
        """[component][feature] embedding weights when this encoder is the integer/sum
        kind the engine evaluates in one kernel, else None."""
        if isinstance(self.input_dims, list) and self.combine == "sum":
            return [[emb.weight for emb in enc.embeddings] for enc in self.encoders]
        return None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        tables = self.engine_tables
        if tables is not None:
            # n*F embedding lookups + adds + stack  ->  one gather-sum kernel reading the tables in place
            flat = _engine.encode(x, _engine.tableset_for(self, tables), tables)          # [N, n*k]
            return flat.view(flat.shape[0], self.phm_dim, self.out_dim)
        return torch.stack([enc(x) for enc in self.encoders], dim=1)       # [*, phm_dim, out_dim]

    def __repr__(self):
        return '{}(in_dims={}, out_dim={},  phm_dim={}, combine="{}")'.format(
            self.__class__.__name__, self.input_dims, self.out_dim, self.phm_dim, self.combine)


class NaivePHMEncoder(torch.nn.Module):
    """One encoder whose output is replicated over the components (ref :45-77)."""

    def __init__(self, out_dim: int, input_dims: Union[list, int], phm_dim: int, combine: str = "sum"):
        super().__init__()
        self.input_dims = input_dims
        self.out_dim = out_dim
        self.phm_dim = phm_dim
        self.combine = combine
        self.encoder = _make_component_encoder(input_dims, out_dim, combine)
        self.reset_parameters()

    def reset_parameters(self):
        self.encoder.reset_parameters()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        enc = self.encoder(x)
        return enc.unsqueeze(1).expand(enc.shape[0], self.phm_dim, *enc.shape[1:]).contiguous()

    def __repr__(self):
        return '{}(in_dims={}, out_dim={}, phm_dim={}, combine="{}")'.format(
            self.__class__.__name__, self.input_dims, self.out_dim, self.phm_dim, self.combine)

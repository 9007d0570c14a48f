"""Integer-feature encoder (reference: phc/quaternion/encoder.py:9-60): one
embedding table per categorical feature, embeddings summed (or concatenated)."""
import math

import torch


class IntegerEncoder(torch.nn.Module):
    def __init__(self, out_dim: int, input_dims: list, combine: str = "sum") -> None:
        super().__init__()
        assert combine in ["sum", "concat"]
        self.combine = combine
        self.out_dim = out_dim
        self.input_dims = input_dims
        self.embeddings = torch.nn.ModuleList([torch.nn.Embedding(v, out_dim) for v in input_dims])
        self.reset_parameters()

    def reset_parameters(self):
        for emb in self.embeddings:
            torch.nn.init.xavier_uniform_(emb.weight.data, gain=math.sqrt(2))

    def get_number_of_params(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def print_number_of_params(self):
        return f"Module {self.__class__.__name__} has {self.get_number_of_params()} trainable parameters."

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() == 1:
            x = x.unsqueeze(1)
        parts = [self.embeddings[f](x[:, f]) for f in range(x.size(1))]
        if self.combine == "sum":
            out = parts[0]
            for p in parts[1:]:
                out = out + p
            return out
        return torch.cat(parts, dim=1)

    def __repr__(self):
        return "{}(out_dim={}, input_dims={}, combine='{}')".format(self.__class__.__name__, self.out_dim,
                                                                    self.input_dims, self.combine)

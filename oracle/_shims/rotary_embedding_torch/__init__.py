"""Stand-in for rotary-embedding-torch (lucidrains), 'lang' frequencies."""
import torch
from torch import nn


def _rotate_half(x):
    x = x.reshape(*x.shape[:-1], -1, 2)
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).reshape(*x.shape[:-2], -1)


class RotaryEmbedding(nn.Module):
    def __init__(self, dim, theta=10000):
        super().__init__()
        # (the published package: nn.Parameter(freqs, requires_grad=learned_freq) with learned_freq=False by default)
        self.freqs = nn.Parameter(1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim)), requires_grad=False)

    def rotate_queries_or_keys(self, t, seq_dim=-2):
        n = t.shape[seq_dim]
        pos = torch.arange(n, device=t.device, dtype=self.freqs.dtype)
        ang = torch.einsum("i,j->ij", pos, self.freqs).repeat_interleave(2, dim=-1)
        rot = ang.shape[-1]
        tl, tr = t[..., :rot], t[..., rot:]
        tl = tl * ang.cos() + _rotate_half(tl) * ang.sin()
        return torch.cat((tl, tr), dim=-1)

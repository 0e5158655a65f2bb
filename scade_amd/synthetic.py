"""Synthetic inputs of the benchmark configurations: seeded ray
batches in the reference's ray_batch row layout.  Product-side helper so that bench.py does not
need anything from oracle/ outside its cpu_baseline leg."""
from __future__ import annotations

import torch


def synthetic_rays(n_rays: int, seed: int = 0, near: float = 0.1, far: float = 5.0) -> torch.Tensor:
    """[n_rays, 11] = o(3) d(3) near far viewdir(3): o = 0.1 N(0,1), d = viewdir = unit N(0,1)
    (run_scade_scannet.py:627-639 row layout)."""
    g = torch.Generator().manual_seed(seed)
    o = 0.1 * torch.randn(n_rays, 3, generator=g)
    d = torch.randn(n_rays, 3, generator=g)
    d = d / torch.norm(d, dim=-1, keepdim=True)
    nf = torch.tensor([near, far]).expand(n_rays, 2)
    return torch.cat([o, d, nf, d], -1).contiguous()

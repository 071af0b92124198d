"""Torch formulations of what csrc/gbp_hitmap.h computes (test infrastructure: the kernels are held to these on shapes the reference
fixture does not cover, and these to the imported reference's own numbers -- tests/test_hitmap_gpu.py)."""
import numpy as np
import torch


def statistics_torch(hitmap, log_mean_prior, half_width):
    """Mean and 5 / 50 / 95 % points of log10 conductivity per depth cell from the hit map [B, n_value, n_depth] (the reference derives the
    same from its Histogram2D posterior)."""
    B, nv, nz = hitmap.shape                                      # stored value-major, depth fastest
    centres = (torch.arange(nv, dtype=torch.float64, device=hitmap.device) + 0.5) / nv * (2.0 * half_width) - half_width
    h = hitmap.transpose(1, 2).to(torch.float64)                  # [B, nz, nv]
    tot = h.sum(dim=2).clamp(min=1.0)
    shift = (log_mean_prior / np.log(10.0))[:, None]
    mean = (h * centres).sum(dim=2) / tot + shift
    cdf = torch.cumsum(h, dim=2) / tot[:, :, None]
    pct = []
    for q in (0.05, 0.5, 0.95):
        idx = (cdf < q).sum(dim=2).clamp(max=nv - 1)
        pct.append(centres[idx] + shift)
    return mean, pct



def runs_torch(hitmap):
    """(ptr int64 [B + 1], start int32, value) of the rows of ``hitmap`` flattened: a run starts at cell 0 and at every change of value."""
    hm = hitmap.flatten(1)
    edge = torch.ones_like(hm, dtype=torch.bool)
    edge[:, 1:] = hm[:, 1:] != hm[:, :-1]
    nz = torch.nonzero(edge)                                      # [runs, 2] row-major: sorted by row, then by flat position
    ptr = torch.zeros(hm.shape[0] + 1, dtype=torch.int64, device=hm.device)
    ptr[1:] = torch.cumsum(torch.bincount(nz[:, 0], minlength=hm.shape[0]), 0)
    return ptr, nz[:, 1].to(torch.int32), hm[nz[:, 0], nz[:, 1]]



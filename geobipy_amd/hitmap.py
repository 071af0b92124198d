"""What leaves the device of a block's conductivity-depth hit maps (csrc/gbp_hitmap.h): per-depth statistics for the survey summary and
the maps in run-length form for the results containers -- one streaming kernel each instead of transposes, cumulative sums and a
``nonzero`` over 9e8 cells.  The torch formulations are kept (``*_torch``) as what the tests hold the kernels to; they are not a fallback: the
product entries refuse tensors that are not on the device."""
import numpy as np
import torch

from . import _lib


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


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


def statistics(hitmap, log_mean_prior, half_width):
    """``statistics_torch`` as one kernel (gbp_hitmap_statistics): the percentile cells are the same cells, the mean agrees to rounding."""
    if hitmap.device.type != "cuda":
        raise _lib.NativeLibraryError("hitmap.statistics runs on the device (gbp_hitmap_statistics); statistics_torch is the test reference, not a fallback")
    B, nv, nz = hitmap.shape
    hm = hitmap.contiguous()
    assert hm.dtype == torch.int32
    lmp = log_mean_prior.to(torch.float64).contiguous()
    out = torch.empty((4, B, nz), dtype=torch.float64, device=hm.device)
    with torch.cuda.device(hm.device):
        _lib.check(_lib.load().gbp_hitmap_statistics(B, nv, nz, hm.data_ptr(), lmp.data_ptr(), float(half_width), out[0].data_ptr(), out[1].data_ptr(),
                                                     out[2].data_ptr(), out[3].data_ptr(), _stream(hm.device)))
    return out[0], [out[1], out[2], out[3]]


def runs_torch(hitmap):
    """(ptr int64 [B + 1], start int32, value) of the rows of ``hitmap`` flattened: a run starts at cell 0 and at every change of value."""
    hm = hitmap.flatten(1)
    edge = torch.ones_like(hm, dtype=torch.bool)
    edge[:, 1:] = hm[:, 1:] != hm[:, :-1]
    nz = torch.nonzero(edge)                                      # [runs, 2] row-major: sorted by row, then by flat position
    ptr = torch.zeros(hm.shape[0] + 1, dtype=torch.int64, device=hm.device)
    ptr[1:] = torch.cumsum(torch.bincount(nz[:, 0], minlength=hm.shape[0]), 0)
    return ptr, nz[:, 1].to(torch.int32), hm[nz[:, 0], nz[:, 1]]


def runs(hitmap):
    """``runs_torch`` as two passes of one kernel (gbp_hitmap_runs: count, prefix, write) -- identical output."""
    if hitmap.device.type != "cuda":
        raise _lib.NativeLibraryError("hitmap.runs runs on the device (gbp_hitmap_runs); runs_torch is the test reference, not a fallback")
    hm = hitmap.flatten(1).contiguous()
    assert hm.dtype == torch.int32
    B, M = hm.shape
    dev = hm.device
    lib = _lib.load()
    ptr = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    if B == 0:
        return ptr, torch.empty(0, dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.gbp_hitmap_runs(B, M, hm.data_ptr(), ptr[1:].data_ptr(), None, None, None, _stream(dev)))
        torch.cumsum(ptr[1:], 0, out=ptr[1:])
        n = int(ptr[-1])
        start = torch.empty(n, dtype=torch.int32, device=dev)
        value = torch.empty(n, dtype=torch.int32, device=dev)
        _lib.check(lib.gbp_hitmap_runs(B, M, hm.data_ptr(), None, ptr.data_ptr(), start.data_ptr(), value.data_ptr(), _stream(dev)))
    return ptr, start, value

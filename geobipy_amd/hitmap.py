"""What leaves the device of a block's conductivity-depth hit maps (csrc/gbp_hitmap.h): per-depth statistics for the survey summary and
the maps in run-length form for the results containers -- one streaming kernel each instead of transposes, cumulative sums and a
``nonzero`` over 9e8 cells.  The statistics are the reference's ``Histogram.mean`` / ``Histogram.percentile`` along the value axis of
the conductivity-depth posterior (statistics/Histogram.py:262-284, 369-401 -> mesh/Mesh.py:80-113, 173-215), in log10: pinned to values
the imported reference computed (tests/golden/make_hitmap_stats.py -> hitmap_stats.npz, tests/test_hitmap_gpu.py).  There is no
fallback: the entries refuse tensors that are not on the device (the torch formulations the kernels were first held to live in
tests/hitmap_reference.py)."""
import torch

from . import _lib


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def statistics(hitmap, log_mean_prior, half_width):
    """Mean and 5 / 50 / 95 % points of log10 conductivity per depth cell from the hit maps [B, n_value, n_depth] (value-major, depth
    fastest) in one kernel (gbp_hitmap_statistics): sum(count x cell centre) / total + the prior mean, and the centre of the first cell
    whose cumulative share reaches the percentile (an empty column: the prior mean / the last cell, as the reference's)."""
    if hitmap.device.type != "cuda":
        raise _lib.NativeLibraryError("hitmap.statistics runs on the device (gbp_hitmap_statistics); there is no host fallback")
    B, nv, nz = hitmap.shape
    hm = hitmap.contiguous()
    assert hm.dtype == torch.int32
    lmp = log_mean_prior.to(torch.float64).contiguous()
    out = torch.empty((4, B, nz), dtype=torch.float64, device=hm.device)
    with torch.cuda.device(hm.device):
        _lib.check(_lib.load().gbp_hitmap_statistics(B, nv, nz, hm.data_ptr(), lmp.data_ptr(), float(half_width), out[0].data_ptr(), out[1].data_ptr(),
                                                     out[2].data_ptr(), out[3].data_ptr(), _stream(hm.device)))
    return out[0], [out[1], out[2], out[3]]


def runs(hitmap):
    """(ptr int64 [B + 1], start int32, value) of the rows of ``hitmap`` flattened -- a run starts at cell 0 and at every change of
    value -- as two passes of one kernel (gbp_hitmap_runs: count, prefix, write)."""
    if hitmap.device.type != "cuda":
        raise _lib.NativeLibraryError("hitmap.runs runs on the device (gbp_hitmap_runs); there is no host fallback")
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

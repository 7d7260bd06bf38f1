"""Data-parallel calibration: merge the per-rank observer statistics with RCCL over xGMI.

The reference has no distributed runtime (one process, one device).  Calibration
shards embarrassingly over samples and every observer statistic is an associative reduction, so:

* one process per GPU (``torch.distributed``, backend ``nccl`` == RCCL on ROCm; ``gloo`` on CPU for
  the unit tests), each rank runs the calibration forward passes on its own shard of the batches;
* at the end of each calibration phase the ranks merge with ONE all-reduce per reduction kind
  over ONE flat buffer that concatenates the statistics of every observer of the graph:
      phase 1  running ranges      MIN over [mins ; -maxs]          (a few KB)
               percentile sums     SUM                              (3 floats per observer)
      phase 2  histograms          SUM over int32 [n_observers x bins]   (~1 MB for ResNet-50)
  Payloads are latency-bound on xGMI (7 links x ~153 GB/s per GPU), so the number of collectives,
  not their size, is what is minimised -- never one collective per tensor.
* integer SUM and float MIN/MAX are order independent: the merged statistics, and therefore the
  rendered scales, are bit-identical to a single-GPU calibration over the union of the shards, and
  identical on every rank (no broadcast of results needed).
"""
from typing import List, Optional, Sequence

import torch


def is_distributed(group=None) -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def merge_observers(observers: Sequence, group=None) -> int:
    """All-reduce, in place, every buffer the observers declare ``reducible()``.  Returns the number
    of collectives issued (0 when not distributed)."""
    if not is_distributed(group):
        return 0
    import torch.distributed as dist
    mins: List[torch.Tensor] = []     # reduced with MIN ('max' buffers are negated into this list)
    maxs: List[torch.Tensor] = []
    sums: dict = {}                   # dtype -> list of buffers
    for ob in observers:
        for buf, kind in ob.reducible():
            if buf is None: continue
            if kind == 'min': mins.append(buf)
            elif kind == 'max': maxs.append(buf)
            elif kind == 'sum': sums.setdefault(buf.dtype, []).append(buf)
            else: raise ValueError(f'unknown reduction {kind}')
    issued = 0
    if mins or maxs:
        flat = torch.cat([b.reshape(-1) for b in mins] + [-b.reshape(-1) for b in maxs])
        dist.all_reduce(flat, op=dist.ReduceOp.MIN, group=group)
        issued += 1
        pos = 0
        for b in mins:
            n = b.numel(); b.copy_(flat[pos: pos + n].reshape(b.shape)); pos += n
        for b in maxs:
            n = b.numel(); b.copy_((-flat[pos: pos + n]).reshape(b.shape)); pos += n
    for dtype, bufs in sums.items():
        flat = torch.cat([b.reshape(-1) for b in bufs])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        issued += 1
        pos = 0
        for b in bufs:
            n = b.numel(); b.copy_(flat[pos: pos + n].reshape(b.shape)); pos += n
    return issued


def shard_batches(batches: Sequence, rank: Optional[int] = None, world_size: Optional[int] = None) -> list:
    """Round-robin shard of the calibration batches for this rank."""
    import torch.distributed as dist
    if rank is None: rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    if world_size is None:
        world_size = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    return [b for i, b in enumerate(batches) if i % world_size == rank]

"""Data-parallel calibration: merge the per-rank observer statistics with RCCL over xGMI.

The reference has no distributed runtime (one process, one device).  Calibration
shards embarrassingly over samples and every observer statistic is an associative reduction, so:

* one process per GPU (``torch.distributed``, backend ``nccl`` == RCCL on ROCm; ``gloo`` on CPU for
  the unit tests), each rank runs the calibration forward passes on its own shard of the batches;
* at the end of each calibration phase the ranks merge with ONE all-reduce per reduction kind
  over ONE flat buffer that concatenates the statistics of every observer of the graph:
      phase 1  running ranges      MIN over [mins ; -maxs]          (a few KB)
               percentile sums     SUM                              (3 floats per observer)
               FP8 'floating'      SUM of the 7 per-candidate squared errors + a count (doubles)
               isotone             all-gather of the [rows, 2] top-2 pairs (a set, not a reduction)
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


last_merge_stats: dict = {}     # what the most recent merge_observers() moved (bench.py reports it)


def _check_layout(dist, lengths: List[int], device, group, largest_count: int = 0) -> int:
    """Every rank must bring the same flat layout: an observer that saw no batch on one rank (fewer
    batches than ranks, uneven shards) declares nothing, and all-reducing buffers of different length
    hangs or corrupts.  ONE tiny MAX all-reduce over [len, -len, ...] detects it on every rank.  The same
    collective carries the largest int32 count any rank holds (returned: the maximum over ranks), which
    decides whether the int32 SUM can overflow."""
    probe = torch.tensor([v for n in lengths for v in (n, -n)] + [int(largest_count)], dtype=torch.int64, device=device)
    dist.all_reduce(probe, op=dist.ReduceOp.MAX, group=group)
    got = probe.tolist()
    if any(got[2 * i] != -got[2 * i + 1] for i in range(len(lengths))):
        raise RuntimeError('merge_observers: the ranks declare different statistics layouts '
                           f'(this rank: {lengths}, max over ranks: {got[0::2]}); every rank must observe at '
                           'least one batch with every observer before a merge (calib_steps >= world_size)')
    return int(got[-1])


def merge_observers(observers: Sequence, group=None, even_if_single_rank: bool = False) -> int:
    """All-reduce, in place, every buffer the observers declare ``reducible()``.  Returns the number
    of data collectives issued (0 when not distributed); sizes and time land in ``last_merge_stats``.
    ``even_if_single_rank``: issue the collectives on a 1-rank group too (a no-op on the data; it proves the
    backend -- RCCL -- loads and accepts every (dtype, reduction) pair used, tests/test_gpu_rccl.py)."""
    global last_merge_stats
    if not is_distributed(group):
        import torch.distributed as dist
        if not (even_if_single_rank and dist.is_available() and dist.is_initialized()):
            last_merge_stats = {}
            return 0
    import time
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
    gathers = [(ob, ob.gatherable()) for ob in observers if hasattr(ob, 'gatherable')]
    gathers = [(ob, bufs) for ob, bufs in gathers if bufs]
    t0 = time.perf_counter()
    sum_dtypes = (torch.int32, torch.float32, torch.float64, torch.int64)
    for dt in sums:
        if dt not in sum_dtypes: raise ValueError(f'merge_observers: unsupported sum dtype {dt}')
    lengths = [sum(b.numel() for b in mins), sum(b.numel() for b in maxs)] + \
              [sum(b.numel() for b in sums.get(dt, [])) for dt in sum_dtypes]
    lengths.append(sum(len(bufs) for _, bufs in gathers))
    some = (mins + maxs + [b for v in sums.values() for b in v] + [b for _, bufs in gathers for b in bufs])
    backend = dist.get_backend(group)
    device = some[0].device if some else torch.device('cuda' if backend == 'nccl' else 'cpu')
    # int32 histograms (the reference's counter type, sort.cu:91-165): world_size ranks x the largest count any of them
    # holds bounds every merged bin -- below 2^31 the int32 SUM cannot wrap; otherwise the buffers are summed in int64 and
    # a bin that really passes 2^31 raises instead of wrapping silently (the reference wraps in one process, too)
    i32 = [b for b in (sums.get(torch.int32) or []) if b.numel()]
    largest = int(torch.stack([b.max() for b in i32]).max()) if i32 else 0          # one device -> host copy
    largest = _check_layout(dist, lengths, device, group, largest)
    widen_i32 = largest * dist.get_world_size(group) >= 2 ** 31
    issued = 0
    moved = {}
    if mins or maxs:
        flat = torch.cat([b.reshape(-1) for b in mins] + [-b.reshape(-1) for b in maxs])
        dist.all_reduce(flat, op=dist.ReduceOp.MIN, group=group)
        issued += 1
        moved['min_f32_bytes'] = flat.numel() * flat.element_size()
        pos = 0
        for b in mins:
            n = b.numel(); b.copy_(flat[pos: pos + n].reshape(b.shape)); pos += n
        for b in maxs:
            n = b.numel(); b.copy_((-flat[pos: pos + n]).reshape(b.shape)); pos += n
    for dtype in sum_dtypes:
        bufs = sums.get(dtype)
        if not bufs: continue
        flat = torch.cat([b.reshape(-1) for b in bufs])
        if dtype == torch.int32 and widen_i32:
            wide = flat.to(torch.int64)
            dist.all_reduce(wide, op=dist.ReduceOp.SUM, group=group)
            if int(wide.max()) >= 2 ** 31:
                raise OverflowError('merge_observers: a merged histogram bin reaches 2^31 counts; int32 histograms cannot hold '
                                    f'it (largest bin {int(wide.max())}).  Calibrate with fewer samples per observer or more bins.')
            flat = wide.to(torch.int32)
            moved['sum_int32_widened'] = 1
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        issued += 1
        moved[f'sum_{str(dtype).split(".")[-1]}_bytes'] = flat.numel() * flat.element_size()
        pos = 0
        for b in bufs:
            n = b.numel(); b.copy_(flat[pos: pos + n].reshape(b.shape)); pos += n
    if gathers:
        # row sets (isotone top-2 pairs): ONE all-gather of the row counts, ONE of the rows padded to the largest count
        world = dist.get_world_size(group)
        flat_bufs = [b for _, bufs in gathers for b in bufs]
        rows = torch.tensor([b.shape[0] for b in flat_bufs], dtype=torch.int64, device=device)
        all_rows = [torch.empty_like(rows) for _ in range(world)]
        dist.all_gather(all_rows, rows, group=group)
        width = [int(b.shape[1]) for b in flat_bufs]
        cap = [int(max(int(r[k]) for r in all_rows)) for k in range(len(flat_bufs))]
        mine = torch.cat([torch.cat([b.reshape(b.shape[0], width[k]).float(),        # (a rank may bring ZERO rows: no -1 here)
                                     torch.zeros(cap[k] - b.shape[0], width[k], dtype=torch.float32, device=device)]).reshape(-1)
                          for k, b in enumerate(flat_bufs)])
        everyone = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine, group=group)
        issued += 2
        moved['gather_f32_bytes'] = mine.numel() * 4 * world
        merged, pos = [[] for _ in flat_bufs], 0
        for k in range(len(flat_bufs)):
            n = cap[k] * width[k]
            for r in range(world):
                merged[k].append(everyone[r][pos: pos + n].reshape(cap[k], width[k])[: int(all_rows[r][k])])
            pos += n
        k = 0
        for ob, bufs in gathers:
            ob.take_gathered([torch.cat(merged[k + i]).to(bufs[i].dtype) for i in range(len(bufs))])
            k += len(bufs)
    if some and some[0].is_cuda: torch.cuda.synchronize(some[0].device)
    last_merge_stats = {'collectives': issued, 'ms': (time.perf_counter() - t0) * 1e3, 'world_size': dist.get_world_size(group),
                        'backend': backend, **moved}
    return issued


def shard_batches(batches: Sequence, rank: Optional[int] = None, world_size: Optional[int] = None) -> list:
    """Round-robin shard of the calibration batches for this rank."""
    import torch.distributed as dist
    if rank is None: rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    if world_size is None:
        world_size = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    return [b for i, b in enumerate(batches) if i % world_size == rank]

"""Calibration observers over the HIP kernels.

Mirror of ppq/quantization/observer/{base,range,floating,__init__}.py: same class names, same
``observe(value)`` / ``render_quantization_config()`` protocol, same ``OBSERVER_TABLE`` keys, same
two-phase behaviour of the histogram observers, and the same host arithmetic for turning ranges /
histograms into ``scale`` / ``offset`` (``minmax_to_scale_offset`` is restated line by line).

What is different is where the statistics live and how they are reduced (MI355X-first):

* every observer keeps its running statistics in small DEVICE buffers that the kernels accumulate
  into (``MinMax_T/C``, ``Histogram_T`` / ``Histogram_Asymmetric_T``, ``Quantile``): one streaming
  pass per observed tensor, no transpose/flatten copies, no per-batch host synchronisation (the
  reference appends a tensor per batch and synchronises in every render, range.py:86-98, 113-114);
* the searches run on the device, batched over all observers of a graph
  (:func:`render_observers`): one ``KLLosses`` / ``MseSearch`` launch and ONE device->host copy per
  calibration phase instead of one per tensor (range.py:226, 465);
* the buffers are flat and additive, so data-parallel calibration merges them with one all-reduce
  per phase (ppq_amd/distributed.py).

Each observer also works stand-alone (``render_quantization_config()`` with no batching), which is
what PPQ's own ``RuntimeCalibrationPass`` does when these classes are registered in PPQ's
``OBSERVER_TABLE``.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import weakref

import numpy as np
import torch

from .core import (OBSERVER_FLOATING_MSE_FETCHES, OBSERVER_ISOTONE_OBSERVER_AXIS, OBSERVER_KL_HIST_BINS, OBSERVER_KL_HIST_BINS_MANUL_OVERRIDE,
                   OBSERVER_MIN_SCALE, OBSERVER_MIN_SCALE_MANUL_OVERRIDE, OBSERVER_MSE_HIST_BINS,
                   OBSERVER_PERCENTILE, OBSERVER_PERCENTILE_MANUL_OVERRIDE, is_initial, set_activated)
from .core import QuantizationProperty as P
from .core import RoundingPolicy
from .ffi import CUDA, quantile_hint
from .round import ppq_numerical_round, ppq_round_to_power_of_2


# ------------------------------------------------------------------------------- host arithmetic
def minmax_to_scale_offset(min_val: float, max_val: float, config,
                           scale_threshold: float = OBSERVER_MIN_SCALE) -> Tuple[float, float]:
    """ppq/quantization/observer/range.py:22-75, line by line.

    ``min_val`` / ``max_val`` may be Python floats (per-tensor render, ``.item()``) or numpy float32
    scalars (per-channel render, ``.cpu().numpy()``); like the reference under numpy >= 2 the
    subtraction and the offset division then happen in float32."""
    if OBSERVER_MIN_SCALE_MANUL_OVERRIDE in config.detail:
        scale_threshold = config.detail[OBSERVER_MIN_SCALE_MANUL_OVERRIDE]
    scale, offset = 1, 0
    if min_val > 0: min_val = 0
    if max_val < 0: max_val = 0
    if config.policy.has_property(P.ASYMMETRICAL):
        range = float(max_val - min_val)
        scale = range / (config.quant_max - config.quant_min)
        scale = max(scale, scale_threshold)
        offset = ppq_numerical_round(float(-min_val / scale))
    elif config.policy.has_property(P.SYMMETRICAL):
        range = 2 * float(max(abs(max_val), abs(min_val)))
        scale = range / (config.quant_max - config.quant_min)
        scale = max(scale, scale_threshold)
        offset = 0
    else:
        raise TypeError('Tensor Min Max Observer Excepts either ASYMMETRICAL or SYMMETRICAL quantization config.')
    if config.policy.has_property(P.POWER_OF_2):
        scale = ppq_round_to_power_of_2(scale, policy=RoundingPolicy.ROUND_UP)
    return scale, offset


def minmax_to_scale_offset_channels(mins: np.ndarray, maxs: np.ndarray, config,
                                    scale_threshold: float = OBSERVER_MIN_SCALE) -> Tuple[list, list]:
    """``[minmax_to_scale_offset(a, b, config) for a, b in zip(mins, maxs)]`` for float32 per-channel ranges -- what the
    per-channel render loops over (range.py:122-129; 26 560 channels for ResNet-50's weights) -- evaluated on whole arrays.
    Same arithmetic, operation by operation, as the scalar function sees it with numpy float32 scalars under numpy >= 2:
    the clamps to 0, ``max_val - min_val`` and ``-min_val / scale`` in FLOAT32 (NEP 50: the Python-float scale is cast to the
    array's type), everything else in double, the offset rounded half-even on the exact value (``Decimal.quantize`` ==
    ``rint`` there).  POWER_OF_2 configs keep the scalar loop (``math.log2`` vs ``numpy.log2`` may differ in the last place).
    Bit-identical to the loop: tests/test_host_cpu.py::test_vectorised_channel_render_equals_the_scalar_loop."""
    if config.policy.has_property(P.POWER_OF_2) or mins.dtype != np.float32 or maxs.dtype != np.float32:
        pairs = [minmax_to_scale_offset(a, b, config, scale_threshold) for a, b in zip(mins, maxs)]
        return [p[0] for p in pairs], [p[1] for p in pairs]
    if OBSERVER_MIN_SCALE_MANUL_OVERRIDE in config.detail:
        scale_threshold = config.detail[OBSERVER_MIN_SCALE_MANUL_OVERRIDE]
    zero = np.float32(0)
    lo = np.where(mins > 0, zero, mins)                       # NaN compares false: stays, as in `if min_val > 0`
    hi = np.where(maxs < 0, zero, maxs)
    levels = config.quant_max - config.quant_min
    if config.policy.has_property(P.ASYMMETRICAL):
        with np.errstate(over='ignore', divide='ignore', invalid='ignore'):
            scale = (hi - lo).astype(np.float64) / levels      # float32 subtraction (may overflow to inf, as the scalar does), double division
            scale = np.where(scale_threshold > scale, scale_threshold, scale)      # max(scale, threshold): the first wins ties / NaN
            offset = np.rint(((-lo) / scale.astype(np.float32)).astype(np.float64))
        return scale.tolist(), [int(v) for v in offset.tolist()]
    if config.policy.has_property(P.SYMMETRICAL):
        a, b = np.abs(hi), np.abs(lo)
        m = np.where(b > a, b, a)                              # max(abs(max_val), abs(min_val)): the first unless the second is greater
        scale = 2 * m.astype(np.float64) / levels
        scale = np.where(scale_threshold > scale, scale_threshold, scale)
        return scale.tolist(), [0] * int(scale.size)
    raise TypeError('Tensor Min Max Observer Excepts either ASYMMETRICAL or SYMMETRICAL quantization config.')


_DEFERRED_SETS: Optional[list] = None      # render_observers: collect, then ONE host-to-device copy


def _set_per_tensor(config, scale: float, offset: float, device) -> None:
    if _DEFERRED_SETS is not None:
        _DEFERRED_SETS.append((config, scale, offset, device))
        return
    config.scale = torch.tensor([scale], dtype=torch.float32, device=device).squeeze(0)
    config.offset = torch.tensor([offset], dtype=torch.float32, device=device).squeeze(0)
    set_activated(config)


def _flush_deferred_sets(pending: list) -> None:
    """Upload every (scale, offset) pair rendered in this round with one copy per device and hand each
    config its own 0-d float32 tensors (views of the uploaded buffer; same values as the per-config
    torch.tensor(...) of the reference, range.py:113-114)."""
    by_dev: Dict[object, list] = {}
    for item in pending: by_dev.setdefault(torch.device(item[3]), []).append(item)
    for dev, items in by_dev.items():
        # per-tensor items contribute (scale, offset), per-channel items their C scales then their C offsets
        chunks = [np.asarray(v, dtype=np.float32).reshape(-1) for _, s, o, _ in items for v in (s, o)]
        flat = torch.from_numpy(np.concatenate(chunks)).to(dev)
        pos = 0
        for k, (config, s, _, _) in enumerate(items):
            if isinstance(s, (list, tuple)):                        # per channel: float32[C] views (range.py:130-131)
                C = len(s)
                config.scale, config.offset = flat[pos: pos + C], flat[pos + C: pos + 2 * C]
                pos += 2 * C
            else:                                                   # per tensor: 0-d views
                config.scale, config.offset = flat[pos], flat[pos + 1]
                pos += 2
            set_activated(config)


class ObservationQueue:
    """Defers the per-tensor statistics kernels of one forward so ONE multi-tensor launch serves them all
    (``CUDA.MinMax_T_Slots_Multi`` / ``CUDA.Histogram_*_T_Rows_Multi``): a calibration forward of
    ResNet-50 observes 72 activation tensors of 3..100 MB, and launched one by one every kernel pays
    ~5 us of launch / fill / drain latency on top of its streaming time and ~7 us of host time.

    Contract: an observed tensor must not be modified in place between ``observe`` and ``flush``
    (the pass flushes at the end of every forward; executor ops return new tensors).  The queue keeps
    the tensors alive until then; ``max_pending_bytes`` bounds that (4 GiB by default -- one launch per
    forward of ResNet-50 at batch 32, 2.15 GB; measured in situ: 0.5 / 1 / 4 GiB thresholds give
    4.0 / 4.7 / 5.0 TB/s for the histogram launch)."""
    def __init__(self, max_pending_bytes: int = 4 << 30):
        self._minmax = []                  # (tensor, slots)
        self._minmax_c = []                # (tensor, channel_axis, mins, maxs, fresh)
        self._hist = {}                    # (asymmetric, bins, device) -> [(tensor, rows, p0, p1)]
        self._quantile = {}                # (q, device) -> [(tensor, dest)]
        self._bytes = 0
        self._max = max_pending_bytes
        self.launches = 0
        self.recorder: Optional[list] = None   # RuntimeCalibrationPass(reuse_activations=True): (observer, tensor) of phase 1

    def __len__(self):
        return (len(self._minmax) + len(self._minmax_c) + sum(len(v) for v in self._hist.values())
                + sum(len(v) for v in self._quantile.values()))

    def _grow(self, value) -> None:
        self._bytes += value.numel() * 4
        if self._bytes > self._max: self.flush()

    def add_minmax(self, value: torch.Tensor, slots: torch.Tensor) -> None:
        self._minmax.append((value, slots))
        self._grow(value)

    def add_minmax_c(self, value: torch.Tensor, channel_axis: int, mins: torch.Tensor, maxs: torch.Tensor, fresh: bool) -> None:
        """Per-channel running range (``CUDA.MinMax_C_Multi``); ``fresh``: mins / maxs are uninitialised and this item
        qualifies for being overwritten (``CUDA.minmax_c_fresh_ok``)."""
        # one buffer, one job per launch: a second observation of the same observer must not share a launch with a pending
        # `fresh` (overwriting) job of it -- the two would race on the buffer
        if any(m.data_ptr() == mins.data_ptr() for _, _, m, _, _ in self._minmax_c): self.flush()
        self._minmax_c.append((value, channel_axis, mins, maxs, bool(fresh)))
        self._grow(value)

    def add_hist(self, value: torch.Tensor, rows: torch.Tensor, asymmetric: bool, p0: float, p1: float = 0.0) -> None:
        self._hist.setdefault((bool(asymmetric), rows.shape[1], value.device), []).append((value, rows, p0, p1))
        self._grow(value)

    def add_quantile(self, value: torch.Tensor, q: float, hint: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Returns the float32[2] (max-side, min-side) result tensor; it is filled at the next flush.
        ``hint``: the observer's threshold hint (ffi.quantile_hint), see include/ppq_hip.h ppqhip_quantile_t."""
        dest = torch.empty(2, dtype=torch.float32, device=value.device)
        self._quantile.setdefault((float(q), value.device), []).append((value, dest, hint))
        self._grow(value)
        return dest

    def flush(self) -> None:
        if self._quantile:
            pending, self._quantile = self._quantile, {}
            for (q, _), items in pending.items():
                # one hint belongs to ONE job of a sequence: an observer that observed twice before this flush hands its
                # hint to its LAST tensor only (the tails of two jobs would write the 8 hint words unsynchronised; results
                # never depend on a hint, but a torn one costs the exact passes -- ADVICE r3)
                hints, seen = [h for _, _, h in items], set()
                for k in range(len(hints) - 1, -1, -1):
                    if hints[k] is None: continue
                    key = hints[k].data_ptr()
                    if key in seen: hints[k] = None
                    else: seen.add(key)
                CUDA.Quantile_Multi([v for v, _, _ in items], q, [d for _, d, _ in items], hints)
                self.launches += 1
        if self._minmax:
            by_dev = {}
            for v, sl in self._minmax: by_dev.setdefault(v.device, []).append((v, sl))
            self._minmax = []
            for items in by_dev.values():
                CUDA.MinMax_T_Slots_Multi([v for v, _ in items], [sl for _, sl in items])
                self.launches += 1
        if self._minmax_c:
            by_dev = {}
            for it in self._minmax_c: by_dev.setdefault(it[0].device, []).append(it)
            self._minmax_c = []
            for dev, items in by_dev.items():
                CUDA.MinMax_C_Multi([i[0] for i in items], [i[1] for i in items], [i[2] for i in items],
                                    [i[3] for i in items], [i[4] for i in items])
                self.launches += 1
        if self._hist:
            pending, self._hist = self._hist, {}
            for (asym, _, _), items in pending.items():
                vs, rows = [i[0] for i in items], [i[1] for i in items]
                if asym:
                    CUDA.Histogram_Asymmetric_T_Rows_Multi([i[2] for i in items], [i[3] for i in items], vs, rows)
                else:
                    CUDA.Histogram_T_Rows_Multi(vs, rows, [i[2] for i in items])
                self.launches += 1
        self._bytes = 0


# Observers driven by a pass that renders them ONE BY ONE (PPQ's own RuntimeCalibrationPass after
# install_plugins_into_ppq(observers=True): optim/calibration.py renders operation by operation) pay a device synchronisation
# per observer: 316 of them per ResNet-50 render, 68 ms per pass.  With SIBLING_PREFETCH on, the first
# stand-alone render that finds its host values missing fetches them for EVERY live observer in the same state at once (steps
# 1-2 of render_observers: one copy of all ranges, one search launch + one copy per group of histograms); the siblings'
# own renders then finish on host values.  Results are the same numbers; an observer that observes again drops what was
# prefetched for it (observe() invalidates), so a sibling of another, still collecting pass is never rendered stale.
SIBLING_PREFETCH = False
_LIVE_OBSERVERS: "weakref.WeakSet" = weakref.WeakSet()
_PREFETCHING = False


def _prefetch_siblings(ob) -> None:
    global _PREFETCHING
    if not SIBLING_PREFETCH or _PREFETCHING: return
    _PREFETCHING = True
    try:
        def device_of(o):
            t = getattr(o, '_hist', None) if getattr(o, '_hist', None) is not None else getattr(o, '_range', None)
            return None if t is None else t.device
        dev = device_of(ob)
        # range and histogram observers only (the FP8 'floating' observer's batched search keeps no per-observation invalidation)
        sibs = [o for o in list(_LIVE_OBSERVERS) if isinstance(o, TorchMinMaxObserver) and is_initial(o._quant_cfg) and device_of(o) == dev]
        if ob not in sibs: sibs.append(ob)
        _prefetch(sibs)
    finally:
        _PREFETCHING = False


class BaseTensorObserver:
    """ppq/quantization/observer/base.py:9-30."""
    queue: Optional[ObservationQueue] = None      # set by RuntimeCalibrationPass(batch_observations=True)

    def __init__(self, watch_on, quant_cfg):
        self._watch_on = watch_on
        self._quant_cfg = quant_cfg
        _LIVE_OBSERVERS.add(self)

    def _drain(self) -> None:
        """Statistics are about to be read: make sure nothing of this pass is still queued."""
        if self.queue is not None and len(self.queue): self.queue.flush()

    def observe(self, value):
        raise NotImplementedError('Implement this function first.')

    def render_quantization_config(self):
        raise NotImplementedError('Implement this function first.')

    def __str__(self) -> str:
        return ('PPQ Tensor Observer (' + self.__class__.__name__ + ') mount on variable ' +
                getattr(self._watch_on, 'name', str(self._watch_on)) + ' observing algorithm: ' +
                str(self._quant_cfg.observer_algorithm))

    def report(self) -> str:
        return ''

    # ---- batching / distributed hooks (no twin in the reference) --------------------------------
    def pending_range(self) -> Optional[torch.Tensor]:
        """Device buffer holding the running range this render needs on the host (or None)."""
        return None

    def take_range(self, host: np.ndarray) -> None:
        """Receive the host copy of ``pending_range()`` (fetched for all observers at once)."""

    def reducible(self) -> List[Tuple[torch.Tensor, str]]:
        """Device buffers + reduction ('min' | 'max' | 'sum') that merge shards of a data-parallel
        calibration; applied in place before rendering."""
        return []

    def gatherable(self) -> List[torch.Tensor]:
        """2-D device buffers [rows, k] whose ROWS must be collected from every rank -- statistics that are a set, not a
        reduction (the isotone observer's top-2 pairs).  ``take_gathered`` receives, per buffer, the concatenation over
        ranks in rank order, identical on every rank."""
        return []

    def take_gathered(self, merged: List[torch.Tensor]) -> None:
        pass


_RANGE_SEEDS: Dict[object, tuple] = {}


def _range_seed(device) -> tuple:
    """([+inf, -inf], the same repeated for every min/max slot) resident on `device`, built once."""
    hit = _RANGE_SEEDS.get(device)
    if hit is None:
        one = torch.tensor([float('inf'), float('-inf')], dtype=torch.float32, device=device)
        hit = _RANGE_SEEDS[device] = (one, one.repeat(CUDA.minmax_slots(), 1).contiguous())
    return hit


class TorchMinMaxObserver(BaseTensorObserver):
    """range.py:78-137.  Running min/max live on the device: float32[2] (per tensor) or
    float32[C] x 2 (per channel), accumulated by the minmax kernels."""
    def __init__(self, watch_on, quant_cfg):
        super().__init__(watch_on, quant_cfg)
        self._range: Optional[torch.Tensor] = None      # [2] = (min, max)   or   [2, C]
        self._slots: Optional[torch.Tensor] = None      # per-tensor: [minmax_slots, 2], one slot per workgroup
        self._slots_dirty = False
        self._queued_c = False                          # a per-channel observation sits in the ObservationQueue
        self._host_range: Optional[np.ndarray] = None
        self._observed = False

    @ torch.no_grad()
    def observe(self, value: torch.Tensor):
        assert isinstance(value, torch.Tensor), 'TorchMinMaxObserver can only deal with torch Tensor values'
        assert value.numel() > 0, (f'You are observing an empty tensor({getattr(self._watch_on, "name", "")}).')
        if not is_initial(self._quant_cfg): return
        cfg = self._quant_cfg
        if cfg.policy.has_property(P.PER_TENSOR):
            if self._range is None:
                seed = _range_seed(value.device)                 # device-side clones: no host-to-device copy per observer
                self._range = seed[0].clone()
                self._slots = seed[1].clone()
            if self.queue is not None and value.is_cuda: self.queue.add_minmax(value, self._slots)
            else: CUDA.MinMax_T_Slots(value, self._slots)    # no per-batch reduction kernel
            self._slots_dirty = True
        elif cfg.policy.has_property(P.PER_CHANNEL):
            # parameter-shaped tensors (channel axis outermost: one row per channel) join the forward's ONE multi-tensor
            # launch; an activation [N, C, ...] keeps its own launch -- the single-tensor kernel groups several short rows
            # of a channel per wave, the multi kernel takes one wave per row
            queued = (self.queue is not None and value.is_cuda and CUDA.minmax_c_outer_is_one(value, cfg.channel_axis))
            fresh = False
            if self._range is None:
                C = value.shape[cfg.channel_axis]
                self._range = torch.empty((2, C), dtype=torch.float32, device=value.device)
                # a queued first observation of a weight-shaped tensor lets the multi launch OVERWRITE the range: no seeding
                fresh = queued and CUDA.minmax_c_fresh_ok(value, cfg.channel_axis)
                if not fresh: self._range[0].fill_(float('inf')); self._range[1].fill_(float('-inf'))
            if queued:
                self.queue.add_minmax_c(value, cfg.channel_axis, self._range[0], self._range[1], fresh)
                self._queued_c = True
            else: CUDA.MinMax_C(value, cfg.channel_axis, self._range[0], self._range[1])
        else:
            raise TypeError('Min-max Observer only work with per-tensor or per-channel quantize policy.')
        self._observed = True
        self._host_range = None

    def _fold(self) -> None:
        """Fold the per-workgroup slots into the running [min, max] (one tiny launch, at render)."""
        if self._queued_c:
            self._drain()
            self._queued_c = False
        if self._slots_dirty:
            self._drain()
            CUDA.MinMax_Slots_Finish(self._slots, self._range)
            self._slots_dirty = False

    def pending_range(self):
        if not (is_initial(self._quant_cfg) and self._observed): return None
        self._fold()
        return self._range

    def take_range(self, host: np.ndarray) -> None:
        self._host_range = host

    def reducible(self):
        if self._range is None: return []
        self._fold()
        if self._range.ndim == 1: return [(self._range[0:1], 'min'), (self._range[1:2], 'max')]
        return [(self._range[0], 'min'), (self._range[1], 'max')]

    def _range_on_host(self) -> np.ndarray:
        if not self._observed:
            raise ValueError('Can not render quantization config yet, Observer data collator is empty. '
                             'Invoke observe() function before render config.')
        if self._host_range is None: _prefetch_siblings(self)
        if self._host_range is None:
            self._fold()
            self._host_range = self._range.cpu().numpy()
        return self._host_range

    def render_quantization_config(self):
        if not is_initial(self._quant_cfg): return
        cfg = self._quant_cfg
        r = self._range_on_host()
        device = self._range.device
        if cfg.policy.has_property(P.PER_TENSOR):
            scale, offset = minmax_to_scale_offset(min_val=float(r[0]), max_val=float(r[1]), config=cfg)
            _set_per_tensor(cfg, scale, offset, device)
        elif cfg.policy.has_property(P.PER_CHANNEL):
            # (the reference loops over numpy float32 scalars, range.py:125-129; the same arithmetic on whole arrays)
            scales, offsets = minmax_to_scale_offset_channels(np.ascontiguousarray(r[0]), np.ascontiguousarray(r[1]), cfg)
            if _DEFERRED_SETS is not None:                      # render_observers: one host-to-device copy for all configs
                _DEFERRED_SETS.append((cfg, scales, offsets, device))
                return
            cfg.scale = torch.tensor(scales, dtype=torch.float32, device=device)
            cfg.offset = torch.tensor(offsets, dtype=torch.float32, device=device)
            set_activated(cfg)
        else:
            raise TypeError('Min-max Observer only work with per-tensor or per-channel quantize policy.')


class TorchHistObserver(TorchMinMaxObserver):
    """range.py:140-309 (KL).  Phase 'Detecting Minmax' -> phase 'Collating Hist' -> KL search."""
    def __init__(self, watch_on, quant_cfg, hist_bins: int = OBSERVER_KL_HIST_BINS):
        self._phase = 'Detecting Minmax'
        self._hist = None
        self._rows = None            # [hist_rows, bins]: one accumulator row per workgroup, folded at render
        self._rows_dirty = False
        self._hist_scale = None
        self._min = None
        self._max = None
        self._losses: Optional[np.ndarray] = None
        if OBSERVER_KL_HIST_BINS_MANUL_OVERRIDE in quant_cfg.detail:
            hist_bins = quant_cfg.detail[OBSERVER_KL_HIST_BINS_MANUL_OVERRIDE]
        self._hist_bins = hist_bins
        super().__init__(watch_on, quant_cfg)

    def observe(self, value: torch.Tensor):
        if not is_initial(self._quant_cfg): return
        assert value.numel() > 0, (f'You are observing an empty tensor({getattr(self._watch_on, "name", "")}).')
        if self._phase == 'Detecting Minmax':
            if self.queue is not None and self.queue.recorder is not None and value.is_cuda:
                self.queue.recorder.append((self, value))
            return super().observe(value)
        elif self._phase == 'Collating Hist':
            self._losses = None                              # (a search result prefetched for an earlier state of the histogram)
            if getattr(self, '_best', None) is not None: self._best = None
            if self._hist is None:
                self._hist = torch.zeros(size=(self._hist_bins,), dtype=torch.int32, device=value.device)
                if self._hist_bins <= 16384:
                    self._rows = torch.zeros(size=(CUDA.hist_rows(), self._hist_bins), dtype=torch.int32,
                                             device=value.device)
            if self._quant_cfg.policy.has_property(P.ASYMMETRICAL):
                if self._rows is not None:
                    if self.queue is not None: self.queue.add_hist(value, self._rows, True, self._min, self._max)
                    else: CUDA.Histogram_Asymmetric_T_Rows(self._min, self._max, tensor=value, rows=self._rows)
                    self._rows_dirty = True
                else:
                    CUDA.Histogram_Asymmetric_T(self._min, self._max, tensor=value, histogram=self._hist)
            elif self._quant_cfg.policy.has_property(P.SYMMETRICAL):
                if self._rows is not None:
                    if self.queue is not None: self.queue.add_hist(value, self._rows, False, self._hist_scale)
                    else: CUDA.Histogram_T_Rows(tensor=value, rows=self._rows, scale=self._hist_scale)
                    self._rows_dirty = True
                else:
                    CUDA.Histogram_T(tensor=value, histogram=self._hist, scale=self._hist_scale)
            else:
                raise TypeError('Quantization Property is invalid, expect either ASYMMETRICAL or SYMMETRICAL config here.')

    def pending_range(self):
        return super().pending_range() if self._phase == 'Detecting Minmax' else None

    def _fold_hist(self) -> None:
        """Sum the per-workgroup rows into the histogram (one launch, at render) and release them."""
        if self._rows_dirty:
            self._drain()
            CUDA.Histogram_Rows_Finish(self._rows, self._hist)
            self._rows_dirty = False
            # the rows may have been allocated while a side stream was current (async_observe): tell the
            # caching allocator that the finish kernel on THIS stream still reads them before they go back
            if self._rows.is_cuda: self._rows.record_stream(torch.cuda.current_stream(self._rows.device))
            self._rows = None

    def histogram(self) -> Optional[torch.Tensor]:
        self._fold_hist()
        return self._hist

    def reducible(self):
        if self._phase == 'Detecting Minmax': return super().reducible()
        return [(self.histogram(), 'sum')] if self._hist is not None else []

    # ---- search ------------------------------------------------------------------------------
    search_kind = 'kl'

    def search_key(self):
        """Observers with equal keys are searched by one batched launch."""
        return ('kl', self._hist_bins, self._quant_cfg.num_of_bits)

    def take_losses(self, losses: np.ndarray) -> None:
        self._losses = losses

    def hist_to_scale_offset(self, histogram: torch.Tensor, hist_bins: int, hist_scale: float, config,
                             scale_threshold: float = OBSERVER_MIN_SCALE) -> Tuple[float, int]:
        """range.py:190-282: the divergences come from the KLLosses kernel; selection, scale formula,
        threshold and power-of-2 rounding are the reference's host code."""
        if config.policy.has_property(P.ASYMMETRICAL):
            raise PermissionError('KL observer is not designed for ASYMMETRICAL quantization')
        if OBSERVER_MIN_SCALE_MANUL_OVERRIDE in config.detail:
            scale_threshold = config.detail[OBSERVER_MIN_SCALE_MANUL_OVERRIDE]
        quant_bins = 2 ** (config.num_of_bits - 1)
        if self._losses is None: _prefetch_siblings(self)
        if self._losses is None:
            self._losses = CUDA.KLLosses(histogram, config.num_of_bits).cpu().numpy()[0]
        losses = [{'kl': float(kl), 'bin_range': (j + 1) * quant_bins} for j, kl in enumerate(self._losses)]
        best_bin_range = sorted(losses, key=lambda x: x['kl'])[0]['bin_range']     # stable: first minimum
        scale, offset = (best_bin_range / self._hist_bins) * hist_scale * (self._hist_bins / quant_bins), 0
        scale = max(scale, scale_threshold)
        if config.policy.has_property(P.POWER_OF_2):
            scale = ppq_round_to_power_of_2(scale, policy=RoundingPolicy.ROUND_HALF_UP)
        return scale, offset

    def render_quantization_config(self):
        if not is_initial(self._quant_cfg): return
        if not self._quant_cfg.policy.has_property(P.PER_TENSOR):
            raise ValueError('Hist observer can only apply with per-tensor quantization config.')
        if self._phase == 'Detecting Minmax':
            r = self._range_on_host()
            min_val, max_val = float(r[0]), float(r[1])
            if self._quant_cfg.policy.has_property(P.SYMMETRICAL):
                hist_range = float(max(abs(max_val), abs(min_val)))
            else:
                hist_range = max_val - min_val
            self._min = min_val
            self._max = max_val
            self._hist_scale = hist_range / self._hist_bins
            self._phase = 'Collating Hist'
        elif self._phase == 'Collating Hist':
            if self._hist is None:
                raise ValueError('Can not render quantization config yet, histogram is empty. '
                                 'Invoke observe() function before render config.')
            scale, offset = self.hist_to_scale_offset(histogram=self.histogram(), hist_bins=self._hist_bins,
                                                      hist_scale=self._hist_scale, config=self._quant_cfg)
            _set_per_tensor(self._quant_cfg, scale, offset, self._hist.device)


class ChannelwiseKLObserver(TorchHistObserver):
    """SURVEY section 8f-4 -- a capability the reference lacks ('kl' refuses PER_CHANNEL configs,
    range.py:288-289): the two-phase KL calibration PER CHANNEL.  Phase 1 = per-channel min/max
    (MinMax_C), phase 2 = one histogram per channel binned with that channel's own hist_scale
    (Histogram_C_Scales), render = the batched KLLosses search over the C histograms and the
    reference's scale formula channel by channel.  By construction channel c gets exactly the scale the
    per-tensor 'kl' observer renders for the slice of channel c.  Registered as 'kl_channel'."""
    def __init__(self, watch_on, quant_cfg, hist_bins: int = OBSERVER_KL_HIST_BINS):
        super().__init__(watch_on, quant_cfg, hist_bins)
        if not quant_cfg.policy.has_property(P.PER_CHANNEL) or not quant_cfg.policy.has_property(P.SYMMETRICAL):
            raise TypeError('ChannelwiseKLObserver needs a symmetrical per-channel config.')
        self._scales_dev: Optional[torch.Tensor] = None

    def observe(self, value: torch.Tensor):
        if not is_initial(self._quant_cfg): return
        assert value.numel() > 0, (f'You are observing an empty tensor({getattr(self._watch_on, "name", "")}).')
        if self._phase == 'Detecting Minmax':
            return TorchMinMaxObserver.observe(self, value)
        C = value.shape[self._quant_cfg.channel_axis]
        if self._hist is None:
            self._hist = torch.zeros(size=(C, self._hist_bins), dtype=torch.int32, device=value.device)
        CUDA.Histogram_C_Scales(value, self._quant_cfg.channel_axis, self._hist, self._scales_dev)

    def histogram(self): return self._hist

    def render_quantization_config(self):
        cfg = self._quant_cfg
        if not is_initial(cfg): return
        if self._phase == 'Detecting Minmax':
            r = self._range_on_host()                                      # [2, C]
            hs = [float(max(abs(float(hi)), abs(float(lo)))) / self._hist_bins for lo, hi in zip(r[0], r[1])]
            self._hist_scale = hs
            self._scales_dev = torch.tensor(hs, dtype=torch.float32, device=self._range.device)
            self._phase = 'Collating Hist'
            return
        if self._hist is None:
            raise ValueError('Can not render quantization config yet, histogram is empty. '
                             'Invoke observe() function before render config.')
        losses = CUDA.KLLosses(self._hist, cfg.num_of_bits).cpu().numpy()  # [C, candidates], one launch
        scales = []
        for c in range(self._hist.shape[0]):
            self._losses = losses[c]
            scale, _ = self.hist_to_scale_offset(self._hist[c], self._hist_bins, self._hist_scale[c], cfg)
            scales.append(scale)
        self._losses = None
        cfg.scale = torch.tensor(scales, dtype=torch.float32, device=self._hist.device)
        cfg.offset = torch.zeros(len(scales), dtype=torch.float32, device=self._hist.device)
        set_activated(cfg)


class TorchPercentileObserver(BaseTensorObserver):
    """range.py:312-403.  Per batch: the two order statistics of ``CUDA.Quantile`` (index rule of the
    reference's CUDA path, sort.cu:13-19); render = mean over batches -> minmax_to_scale_offset."""
    def __init__(self, watch_on, quant_cfg):
        super().__init__(watch_on, quant_cfg)
        if OBSERVER_PERCENTILE_MANUL_OVERRIDE not in quant_cfg.detail:
            self._percentile = OBSERVER_PERCENTILE
        else: self._percentile = quant_cfg.detail[OBSERVER_PERCENTILE_MANUL_OVERRIDE]
        self._percentile_collector = []
        self._sum: Optional[torch.Tensor] = None     # float32 [3] = (sum of max-quantiles, sum of min-quantiles, count)
        self._hint: Optional[torch.Tensor] = None    # the filter thresholds that worked for the previous batch (device int32[8])

    @ torch.no_grad()
    def observe(self, value: torch.Tensor):
        if not is_initial(self._quant_cfg): return
        assert value is not None, ('You are observing an Empty Tensor. '
                                   '(This Error is usually due to you have a wrong Quantizer configuration.)')
        assert value.numel() > 0, (f'You are observing an empty tensor({getattr(self._watch_on, "name", "")}).')
        assert isinstance(value, torch.Tensor), 'TorchMinMaxObserver can only deal with torch Tensor values'
        if self._quant_cfg.policy.has_property(P.PER_TENSOR):
            if value.is_cuda and (self._hint is None or self._hint.device != value.device):
                self._hint = quantile_hint(value.device)
            if self.queue is not None and value.is_cuda:
                self._percentile_collector.append(self.queue.add_quantile(value, self._percentile, self._hint).view(1, -1))
            else:
                self._percentile_collector.append(CUDA.Quantile_Hinted(value, self._percentile, self._hint).view(1, -1))
        elif self._quant_cfg.policy.has_property(P.PER_CHANNEL):
            raise PermissionError('Percentile observer can not deal with per channel quantization.')
        else:
            raise TypeError('Min-max Observer only work with per-tensor or per-channel quantize policy.')

    def _fold(self):
        self._drain()
        if self._percentile_collector:
            stacked = torch.cat(self._percentile_collector, dim=0).float()
            part = torch.cat([stacked.sum(dim=0), stacked.new_tensor([float(stacked.shape[0])])])
            self._sum = part if self._sum is None else self._sum + part
            self._percentile_collector = []

    def reducible(self):
        self._fold()
        return [(self._sum, 'sum')] if self._sum is not None else []

    def render_quantization_config(self):
        if not is_initial(self._quant_cfg): return
        if self._quant_cfg.policy.has_property(P.PER_TENSOR):
            single_shard = self._sum is None
            if single_shard and len(self._percentile_collector) == 0:
                raise ValueError('Can not render quantization config yet, Observer data collator is empty. '
                                 'Invoke observe() function before render config.')
            self._drain()
            if single_shard:      # exactly the reference's expression, range.py:369
                device = self._percentile_collector[-1].device
                mean = torch.cat(self._percentile_collector, dim=0).float().mean(dim=0).cpu()
            else:                 # merged shards: sum of per-batch quantiles / number of batches
                self._fold()
                device = self._sum.device
                mean = (self._sum[:2] / self._sum[2]).cpu()
            scale, offset = minmax_to_scale_offset(min_val=mean[1].item(), max_val=mean[0].item(),
                                                   config=self._quant_cfg)
            _set_per_tensor(self._quant_cfg, scale, offset, device)
        elif self._quant_cfg.policy.has_property(P.PER_CHANNEL):
            raise PermissionError('Percentile observer can not deal with per channel quantization.')
        else:
            raise TypeError('Min-max Observer only work with per-tensor or per-channel quantize policy.')


class TorchMSEObserver(TorchHistObserver):
    """range.py:406-520: histogram-accelerated MSE.  The candidate sweep runs in the MseSearch kernel
    with the float32 loss of csrc/cpu/hist_mse.cc (the loss the reference uses when its kernels are
    enabled, range.py:423-425); the winning (start, end) goes through minmax_to_scale_offset here."""
    search_kind = 'mse'

    def __init__(self, watch_on, quant_cfg, bins: int = OBSERVER_MSE_HIST_BINS):
        super().__init__(watch_on, quant_cfg)
        self._hist_bins = bins
        self._best: Optional[np.ndarray] = None

    def search_key(self):
        cfg = self._quant_cfg
        return ('mse', self._hist_bins, cfg.quant_min, cfg.quant_max, cfg.policy.has_property(P.SYMMETRICAL))

    def take_best(self, best: np.ndarray) -> None:
        self._best = best

    def compute_mse_loss(self, histogram: list, start: int, step: int, end: int) -> float:
        return CUDA.compute_mse_loss(histogram=histogram, start=start, step=step, end=end)

    def hist_to_scale_offset(self, histogram: torch.Tensor, hist_bins: int, hist_scale: float, config,
                             scale_threshold: float = OBSERVER_MIN_SCALE) -> Tuple[float, int]:
        if OBSERVER_MIN_SCALE_MANUL_OVERRIDE in config.detail:
            scale_threshold = config.detail[OBSERVER_MIN_SCALE_MANUL_OVERRIDE]
        if config.policy.has_property(P.PER_CHANNEL):
            raise PermissionError('Torch Mse observer do not support PER_CHANNEL policy now, please wait.')
        symmetrical = config.policy.has_property(P.SYMMETRICAL)
        if self._best is None: _prefetch_siblings(self)
        if self._best is None:
            dev = histogram.device
            self._best = CUDA.MseSearch(histogram.reshape(1, -1),
                                        torch.tensor([hist_scale], dtype=torch.float64, device=dev),
                                        torch.tensor([self._min], dtype=torch.float64, device=dev),
                                        config.quant_min, config.quant_max, symmetrical).cpu().numpy()[0]
        best_start, best_end = int(self._best[0]), int(self._best[1])
        if not symmetrical:
            range_min, range_max = (best_start * hist_scale) + self._min, (best_end * hist_scale) + self._min
        else:
            range_min, range_max = -(best_end * hist_scale), (best_end * hist_scale)
        return minmax_to_scale_offset(range_min, range_max, config, scale_threshold)


class ChannelwiseMSEObserver(TorchMSEObserver):
    """SURVEY section 8f-4 -- a capability the reference lacks (TorchMSEObserver raises on PER_CHANNEL configs,
    range.py:496-497): the two-phase histogram MSE calibration PER CHANNEL, symmetrical or asymmetrical.
    Phase 1 = per-channel min/max (MinMax_C); phase 2 = one histogram per channel binned with that channel's
    own range (Histogram_C_Scales / Histogram_Asymmetric_C_Ranges); render = ONE batched MseSearch launch over
    the [C, bins] histograms, then the reference's minmax_to_scale_offset channel by channel.  By
    construction channel c gets exactly the (scale, offset) the per-tensor 'mse' observer renders for the slice
    of channel c.  Registered as 'mse_channel'."""
    def __init__(self, watch_on, quant_cfg, bins: int = OBSERVER_MSE_HIST_BINS):
        super().__init__(watch_on, quant_cfg, bins)
        if not quant_cfg.policy.has_property(P.PER_CHANNEL):
            raise TypeError('ChannelwiseMSEObserver needs a per-channel config.')
        self._mins_dev = self._maxs_dev = self._scales_dev = None

    def observe(self, value: torch.Tensor):
        if not is_initial(self._quant_cfg): return
        assert value.numel() > 0, (f'You are observing an empty tensor({getattr(self._watch_on, "name", "")}).')
        if self._phase == 'Detecting Minmax':
            return TorchMinMaxObserver.observe(self, value)
        cfg = self._quant_cfg
        C = value.shape[cfg.channel_axis]
        if self._hist is None:
            self._hist = torch.zeros(size=(C, self._hist_bins), dtype=torch.int32, device=value.device)
        if cfg.policy.has_property(P.ASYMMETRICAL):
            CUDA.Histogram_Asymmetric_C_Ranges(value, cfg.channel_axis, self._hist, self._mins_dev, self._maxs_dev)
        else:
            CUDA.Histogram_C_Scales(value, cfg.channel_axis, self._hist, self._scales_dev)

    def histogram(self): return self._hist

    def reducible(self):
        if self._phase == 'Detecting Minmax': return TorchMinMaxObserver.reducible(self)
        return [(self._hist, 'sum')] if self._hist is not None else []

    def render_quantization_config(self):
        cfg = self._quant_cfg
        if not is_initial(cfg): return
        sym = cfg.policy.has_property(P.SYMMETRICAL)
        if self._phase == 'Detecting Minmax':
            r = self._range_on_host()                                      # [2, C]
            self._min = [float(v) for v in r[0]]
            self._max = [float(v) for v in r[1]]
            # range.py:294-301, channel by channel
            self._hist_scale = [(float(max(abs(hi), abs(lo))) if sym else (hi - lo)) / self._hist_bins
                                for lo, hi in zip(self._min, self._max)]
            dev = self._range.device
            self._mins_dev = torch.tensor(self._min, dtype=torch.float32, device=dev)
            self._maxs_dev = torch.tensor(self._max, dtype=torch.float32, device=dev)
            self._scales_dev = torch.tensor(self._hist_scale, dtype=torch.float32, device=dev)
            self._phase = 'Collating Hist'
            return
        if self._hist is None:
            raise ValueError('Can not render quantization config yet, histogram is empty. '
                             'Invoke observe() function before render config.')
        dev = self._hist.device
        best = CUDA.MseSearch(self._hist, torch.tensor(self._hist_scale, dtype=torch.float64, device=dev),
                              torch.tensor(self._min, dtype=torch.float64, device=dev), cfg.quant_min, cfg.quant_max,
                              sym).cpu().numpy()                          # [C, 4]: one launch, one D2H
        scale_threshold = cfg.detail.get(OBSERVER_MIN_SCALE_MANUL_OVERRIDE, OBSERVER_MIN_SCALE)
        scales, offsets = [], []
        for c in range(self._hist.shape[0]):
            start, end, hs = int(best[c][0]), int(best[c][1]), self._hist_scale[c]
            if sym: range_min, range_max = -(end * hs), (end * hs)                          # range.py:512-516
            else: range_min, range_max = (start * hs) + self._min[c], (end * hs) + self._min[c]
            s_, o_ = minmax_to_scale_offset(range_min, range_max, cfg, scale_threshold)
            scales.append(s_); offsets.append(o_)
        cfg.scale = torch.tensor(scales, dtype=torch.float32, device=dev)
        cfg.offset = torch.tensor(offsets, dtype=torch.float32, device=dev)
        set_activated(cfg)


class ConstantObserver(BaseTensorObserver):
    """observer/floating.py:11-48: scale 1, offset 0."""
    def __init__(self, watch_on, quant_cfg):
        super().__init__(watch_on, quant_cfg)
        self._value_shape = None
        self._value_device = None

    @ torch.no_grad()
    def observe(self, value: torch.Tensor):
        if not is_initial(self._quant_cfg): return
        self._value_shape = value.shape
        self._value_device = value.device

    def render_quantization_config(self):
        if not is_initial(self._quant_cfg): return
        device = self._value_device
        cfg = self._quant_cfg
        if cfg.policy.has_property(P.FLOATING):
            if cfg.policy.has_property(P.PER_TENSOR):
                _set_per_tensor(cfg, 1.0, 0.0, device)
            elif cfg.policy.has_property(P.PER_CHANNEL):
                num = self._value_shape[cfg.channel_axis]
                cfg.scale = torch.ones(num, dtype=torch.float32, device=device)
                cfg.offset = torch.zeros(num, dtype=torch.float32, device=device)
                set_activated(cfg)
        else:
            raise TypeError('This Observer is designed for floating quantization.')


def tensor_random_fetch(tensor: torch.Tensor, num_of_fetches: int = 1024) -> torch.Tensor:
    """ppq/utils/fetch.py:32-50 (unseeded variant, the one DirectMSEObserver uses)."""
    tensor = tensor.flatten()
    indexer = torch.randint(low=0, high=tensor.numel(), size=[num_of_fetches], device=tensor.device)
    return tensor.index_select(dim=0, index=indexer)


def channel_random_fetch(tensor: torch.Tensor, fetchs_per_channel: int = 1024, channel_axis: int = 0) -> torch.Tensor:
    """ppq/utils/fetch.py:53-84."""
    tensor = tensor.transpose(0, channel_axis).flatten(start_dim=1)
    indexer = torch.randint(low=0, high=tensor.shape[-1], size=[fetchs_per_channel], device=tensor.device)
    return tensor.index_select(dim=-1, index=indexer)


class DirectMSEObserver(BaseTensorObserver):
    """observer/floating.py:51-143 ('floating'): picks the FP8 scale among
    {2^-7, 2^-5, 2^-3, 1, 4, 16, 64} by the MSE of the fake-quantised random fetches."""
    SCALE_CANDIDATES = [.0078125, .03125, .125, 1.0, 4.0, 16.0, 64.0]

    def __init__(self, watch_on, quant_cfg):
        super().__init__(watch_on, quant_cfg)
        if not quant_cfg.policy.has_property(P.FLOATING):
            raise TypeError('MSE Floating Observer is designed for floating quantization.')
        if not quant_cfg.policy.has_property(P.POWER_OF_2):
            raise TypeError('MSE Floating Observer is designed for power-of-2 quantization.')
        self._collector = []
        self._fetches = OBSERVER_FLOATING_MSE_FETCHES

    @ torch.no_grad()
    def observe(self, value: torch.Tensor):
        if not is_initial(self._quant_cfg): return
        cfg = self._quant_cfg
        if cfg.policy.has_property(P.PER_CHANNEL):
            if getattr(self._watch_on, 'is_parameter', False):
                value = torch.transpose(value, dim0=0, dim1=cfg.channel_axis)
                self._collector.append(torch.flatten(value, start_dim=1))
            else:
                self._collector.append(channel_random_fetch(value, fetchs_per_channel=self._fetches,
                                                            channel_axis=cfg.channel_axis))
        if cfg.policy.has_property(P.PER_TENSOR):
            self._collector.append(tensor_random_fetch(value, num_of_fetches=self._fetches))

    def search_item(self):
        """(values [rows, row_len], exponent, mantissa, clip_min, clip_max) for CUDA.FloatScaleSearch: one row per
        channel, or one row holding everything a per-tensor config collected."""
        cfg = self._quant_cfg
        if cfg.policy.has_property(P.PER_CHANNEL): fp = torch.cat(self._collector, dim=-1).contiguous()
        else: fp = torch.cat(self._collector, dim=0).reshape(1, -1).contiguous()
        return (fp, cfg.exponent_bits, cfg.mantissa_bits, cfg.quant_min, cfg.quant_max)

    def take_squared_errors(self, sse: torch.Tensor, row_len: int) -> None:
        """sse: float64 [rows, candidates] of this observer's rows (from a batched FloatScaleSearch)."""
        per_channel = self._quant_cfg.policy.has_property(P.PER_CHANNEL)
        self._sse = sse if per_channel else sse.reshape(-1)
        self._count = torch.full([sse.shape[0] if per_channel else 1], float(row_len), dtype=torch.float64, device=sse.device)

    def take_best(self, index) -> None:
        """index: the arg-min candidate of every row, already on the host (render_observers fetches all at once)."""
        self._best = list(index)

    def _squared_errors(self):
        """Per candidate: (sum of squared fake-quant errors in float64, number of values) -- additive over ranks."""
        item = self.search_item()
        r = int(getattr(self._quant_cfg.rounding, 'value', self._quant_cfg.rounding))
        self.take_squared_errors(CUDA.FloatScaleSearch([item], self.SCALE_CANDIDATES, r), item[0].shape[1])
        return self._sse, self._count

    def reducible(self):
        """Data-parallel calibration: every rank fetched from its own batches; the per-candidate squared
        errors are sums, so 7 doubles (+ a count) per config travel in the phase's one SUM all-reduce and every rank
        picks the same scale -- the arg-min over the union of all ranks' fetches."""
        if not is_initial(self._quant_cfg) or not self._collector: return []
        self._sse, self._count = self._squared_errors()
        return [(self._sse, 'sum'), (self._count, 'sum')]

    def render_quantization_config(self):
        if not is_initial(self._quant_cfg): return
        if not self._collector:
            raise PermissionError('Observer collector is empty, you should invoke observe function'
                                  ' before render quantization config.')
        cfg = self._quant_cfg
        r = int(getattr(cfg.rounding, 'value', cfg.rounding))
        e, m, lo, hi = cfg.exponent_bits, cfg.mantissa_bits, cfg.quant_min, cfg.quant_max
        best = getattr(self, '_best', None)
        if best is not None:                                                # render_observers searched every config in one launch
            dev = self._collector[0].device
            cfg.scale = torch.tensor([self.SCALE_CANDIDATES[i] for i in best], dtype=torch.float32, device=dev)
            cfg.offset = torch.zeros(len(best), dtype=torch.float32, device=dev)
            self._best = self._sse = None
            set_activated(cfg)
            return
        merged = getattr(self, '_sse', None)
        if merged is not None:                                              # after a data-parallel merge
            mean = merged / (self._count.unsqueeze(-1) if merged.ndim == 2 else self._count)
            dev = merged.device
            if cfg.policy.has_property(P.PER_CHANNEL):
                index = torch.argmin(mean, dim=-1).cpu().tolist()
                cfg.scale = torch.tensor([self.SCALE_CANDIDATES[i] for i in index], dtype=torch.float32, device=dev)
                cfg.offset = torch.zeros(len(index), dtype=torch.float32, device=dev)
            else:
                best_scale = sorted(zip(mean.cpu().tolist(), self.SCALE_CANDIDATES))[0][1]
                cfg.scale = torch.tensor([best_scale], dtype=torch.float32, device=dev)
                cfg.offset = torch.zeros(1, dtype=torch.float32, device=dev)
            self._sse = None
            set_activated(cfg)
            return
        if cfg.policy.has_property(P.PER_CHANNEL):
            fp = torch.cat(self._collector, dim=-1).contiguous()           # [C, fetches * batches]
            C = fp.shape[0]
            zeros = torch.zeros(C, dtype=torch.float32, device=fp.device)
            losses = []
            for scale in self.SCALE_CANDIDATES:
                s = torch.full([C], scale, dtype=torch.float32, device=fp.device)
                qt = CUDA.FloatingQuantize_C(fp, s, zeros, 0, e, m, lo, hi, r)
                losses.append(torch.mean(torch.square(qt - fp), dim=-1, keepdim=True))
            index = torch.argmin(torch.cat(losses, dim=-1), dim=-1).cpu().tolist()
            cfg.scale = torch.tensor([self.SCALE_CANDIDATES[i] for i in index], dtype=torch.float32, device=fp.device)
            cfg.offset = zeros
        else:
            fp = torch.cat(self._collector, dim=0).contiguous()
            zero = torch.zeros(1, dtype=torch.float32, device=fp.device)
            losses = []
            for scale in self.SCALE_CANDIDATES:
                s = torch.full([1], scale, dtype=torch.float32, device=fp.device)
                qt = CUDA.FloatingQuantize_T(fp, s, zero, e, m, lo, hi, r)
                losses.append(torch.mean(torch.square(qt - fp)).reshape(1))
            host = torch.cat(losses).cpu().tolist()                        # one D2H for the 7 candidates
            best_scale = sorted(zip(host, self.SCALE_CANDIDATES))[0][1]
            cfg.scale = torch.tensor([best_scale], dtype=torch.float32, device=fp.device)
            cfg.offset = zero
        set_activated(cfg)


class TorchIsotoneObserver(BaseTensorObserver):
    """OBSERVER_TABLE['isotone'] (observer/order.py:12-103): an order-preserving scale for classification outputs.  With L1 > L2
    the two largest entries of a row, argmax survives quantisation when L2 / (quant_max - .51) < scale < 2 (L1 - max(L2, 0));
    the scale inside the most rows' intervals wins, smallest such scale first.

    ``observe`` keeps one [rows, 2] device buffer per batch (torch.topk, as the reference -- including its use of the ORIGINAL
    axis on the already flattened [rows, classes] view, order.py:49-52).  ``render`` is ONE device-to-host copy and a sort-based
    sweep over all 2 x rows interval end points at once (numpy; the reference walks a Python list of tuples): end points are
    ordered by (value, closing-before-opening) exactly as its ``sorted()`` orders ``(value, 'max') < (value, 'min')``, the
    running depth is a cumulative sum and the winner its first maximum.  All arithmetic is float32, like the numpy scalars the
    reference iterates over, so the rendered scale is bit-identical (tests/golden/isotone.npz).
    Data parallel: the pairs are a set -- ``merge_observers`` all-gathers them (``gatherable``)."""
    def __init__(self, watch_on, quant_cfg):
        super().__init__(watch_on, quant_cfg)
        self._pairs: List[torch.Tensor] = []
        self.axis = quant_cfg.detail.get(OBSERVER_ISOTONE_OBSERVER_AXIS, -1)      # order.py:16-19 (implicit axis -1)

    @ torch.no_grad()
    def observe(self, value: torch.Tensor):
        if not is_initial(self._quant_cfg): return
        assert isinstance(value, torch.Tensor), 'IsotoneObserver can only deal with torch Tensor values'
        assert value.numel() > 0, f'You are observing an empty tensor({getattr(self._watch_on, "name", "?")}).'
        policy = self._quant_cfg.policy
        if policy.has_property(P.PER_CHANNEL):
            raise TypeError('Isotone Observer is not designed for channelwise quantization.')
        if not policy.has_property(P.PER_TENSOR):
            raise TypeError('Isotone Observer only work with per-tensor or per-channel quantize policy.')
        rows = value
        if rows.ndim > 1: rows = rows.transpose(self.axis, -1).flatten(0, -2)
        top2 = torch.topk(rows, k=2, dim=self.axis, largest=True, sorted=True).values
        self._pairs.append(top2.reshape(1, -1) if top2.ndim <= 1 else top2)

    def gatherable(self):
        if not is_initial(self._quant_cfg) or not self._pairs: return []
        self._pairs = [torch.cat(self._pairs, dim=0).contiguous()]
        return [self._pairs[0]]

    def take_gathered(self, merged):
        self._pairs = [merged[0]]

    def render_quantization_config(self):
        cfg = self._quant_cfg
        if not is_initial(cfg): return
        device = self._pairs[-1].device
        pairs = torch.cat(self._pairs, dim=0).cpu().numpy().astype(np.float32, copy=False)
        l1, l2 = pairs[:, 0], pairs[:, 1]
        if cfg.policy.has_property(P.SYMMETRICAL): l1, l2 = np.abs(l1), np.abs(l2)
        f32 = np.float32
        lower = np.maximum(l2 / f32(cfg.quant_max - .51), f32(0))               # L2 must not be clipped
        upper = f32(2) * (l1 - np.maximum(l2, f32(0)))                          # L1 and L2 must land in different levels
        usable = (upper > lower) & (l1 > 0)

        def _tensor(v): return torch.tensor([v], dtype=torch.float32, device=device).squeeze(0)
        if not usable.any():
            # no row can be separated: min-max on [0, L1 of the LAST row] (order.py:81-91)
            scale, offset = minmax_to_scale_offset(min_val=0, max_val=l1[-1], config=cfg)
            cfg.scale, cfg.offset = _tensor(scale), _tensor(offset)
            set_activated(cfg)
            return
        ends = np.concatenate([upper[usable], lower[usable]])
        opens = np.concatenate([np.zeros(int(usable.sum()), np.int8), np.ones(int(usable.sum()), np.int8)])
        order = np.lexsort((opens, ends))                                       # by value; an interval closes before another opens
        depth = np.cumsum(np.where(opens[order] == 1, 1, -1))
        best = int(np.argmax(depth))                                            # first maximum = the smallest best scale
        cfg.scale, cfg.offset = _tensor(ends[order][best]), _tensor(0)
        set_activated(cfg)
        self.s_candidates = [(v, 'min' if o else 'max') for v, o in zip(ends[order].tolist(), opens[order].tolist())]


# observer/__init__.py:15-23
OBSERVER_TABLE = {
    'minmax': TorchMinMaxObserver,
    'kl': TorchHistObserver,
    'kl_channel': ChannelwiseKLObserver,       # extension, not in the reference's table
    'percentile': TorchPercentileObserver,
    'mse': TorchMSEObserver,
    'mse_channel': ChannelwiseMSEObserver,     # extension, not in the reference's table
    'isotone': TorchIsotoneObserver,
    'constant': ConstantObserver,
    'floating': DirectMSEObserver,
}


def register_calibration_observer(algorithm: str, observer: type) -> None:
    """ppq/lib/extension.py:76-93: put a user's observer class into OBSERVER_TABLE (an existing entry is replaced without
    warning, as there).  The factory looks algorithms up in lower case (observer/__init__.py:28-38), so the key is stored that
    way.  The reference demands a subclass of ``OperationObserver`` here although the table holds TENSOR observers
    (``build_observer`` instantiates the entry with ``(watch_on, quant_cfg)``): this mirror asks for what the table needs, a
    ``BaseTensorObserver`` subclass."""
    if not isinstance(observer, type):
        raise TypeError(f'You can only register an observer CLASS as custimized ppq observer, however {type(observer)} is given. ')
    if not issubclass(observer, BaseTensorObserver):
        raise TypeError('Regitsing observer must be a subclass of BaseTensorObserver.')
    OBSERVER_TABLE[str(algorithm).lower()] = observer


class TensorObserverFactroy:
    """observer/__init__.py:25-37."""
    def __init__(self) -> None:
        raise NotImplementedError('Observer Factory can not be initialized, use TensorObserverFactroy.build_observer instead.')

    @ classmethod
    def build_observer(cls, variable, config) -> BaseTensorObserver:
        algorithm = str(config.observer_algorithm.lower())
        if algorithm not in OBSERVER_TABLE:
            raise ValueError(f'Observer type not understand, Except one of {OBSERVER_TABLE.keys()}, '
                             f'while {str(algorithm)} was given.')
        return OBSERVER_TABLE[algorithm](watch_on=variable, quant_cfg=config)


class CalibrationHook:
    """observer/__init__.py:40-73 -- the QuantOPRuntimeHook the executor fires around every
    quantable operation (executor/base.py:76-102).

    ``stream`` (optional, set by RuntimeCalibrationPass) moves the observer kernels to a side HIP
    stream: observing only READS the tensor, so the statistics kernels (HBM/latency bound, a few
    microseconds each) overlap with the next convolution of the graph (compute bound) instead of
    sitting on the critical path.  The side stream waits for the producer, the caching allocator is
    told about the cross-stream use (record_stream), and rendering joins the streams."""
    def __init__(self, operation, observer_table: Dict[object, BaseTensorObserver], stream=None) -> None:
        self._hook_to = operation
        self._operation = operation
        self._observer_table = observer_table
        self.stream = stream

    def _observe_all(self, values: list, quant_configs: list) -> None:
        todo = [(v, self._observer_table[c]) for v, c in zip(values, quant_configs) if c in self._observer_table]
        if not todo: return
        if self.stream is None or not todo[0][0].is_cuda:
            for v, ob in todo: ob.observe(v)
            return
        self.stream.wait_stream(torch.cuda.current_stream(todo[0][0].device))
        with torch.cuda.stream(self.stream):
            capturing = torch.cuda.is_current_stream_capturing()
            for v, ob in todo:
                ob.observe(v)
                if not capturing: v.record_stream(self.stream)     # inside a graph the fork/join orders it

    def pre_forward_hook(self, inputs: list, quant_inputs: list, quant_configs: list) -> list:
        self._observe_all(inputs, quant_configs)
        return quant_inputs

    def post_forward_hook(self, outputs: list, quant_outputs: list, quant_configs: list) -> list:
        self._observe_all(outputs, quant_configs)
        return quant_outputs

    def render_quantization_config(self):
        if self.stream is not None: torch.cuda.current_stream().wait_stream(self.stream)
        for _, observer in self._observer_table.items():
            observer.render_quantization_config()
            observer.report()

    def __str__(self) -> str:
        return ''.join([observer.__str__() + '\n' for _, observer in self._observer_table.items()])


class OperationObserver:
    """observer/__init__.py:75-124."""
    def __init__(self, operation, monitor_parameter: bool = True, monitor_outputs: bool = True,
                 monitor_inputs: bool = True) -> None:
        if not hasattr(operation, 'config'):
            raise TypeError(f'Only QuantableOP instance can apply an Observer, while {type(operation)} was given.')
        self._operation = operation
        self._hook = self.build_hook(monitor_parameter=monitor_parameter, monitor_outputs=monitor_outputs,
                                     monitor_inputs=monitor_inputs)

    def render_quantization_config(self):
        self.hook.render_quantization_config()

    def build_hook(self, monitor_parameter: bool, monitor_outputs: bool, monitor_inputs: bool) -> CalibrationHook:
        observer_table = {}
        for var, config in zip(self._operation.inputs, self._operation.config.input_quantization_config):
            if is_initial(config):
                if var in self._operation.parameters and monitor_parameter:
                    observer_table[config] = TensorObserverFactroy.build_observer(var, config)
                elif monitor_inputs:
                    observer_table[config] = TensorObserverFactroy.build_observer(var, config)
        if monitor_outputs:
            for var, config in zip(self._operation.outputs, self._operation.config.output_quantization_config):
                if is_initial(config):
                    observer_table[config] = TensorObserverFactroy.build_observer(var, config)
        return CalibrationHook(operation=self._operation, observer_table=observer_table)

    @ property
    def hook(self) -> CalibrationHook:
        return self._hook

    def observers(self) -> List[BaseTensorObserver]:
        return list(self._hook._observer_table.values())

    def report(self) -> str:
        return str(self._hook)


# ------------------------------------------------------------------------------ batched render
def render_observers(observers: Sequence[BaseTensorObserver]) -> None:
    """Render many observers with a bounded number of kernel launches and host synchronisations:

    1. every pending running range is fetched with ONE device->host copy;
    2. histogram observers that are ready to search are grouped by (kind, bins, levels ...) and each
       group is searched by ONE launch (KLLosses / MseSearch) and ONE device->host copy;
    3. each observer's own ``render_quantization_config()`` then finishes on the host values.

    Results are identical to rendering the observers one by one."""
    observers = [ob for ob in observers if is_initial(ob._quant_cfg)]
    _prefetch(observers)
    # 3. host finish; the per-tensor (scale, offset) results travel to the device in one copy
    global _DEFERRED_SETS
    _DEFERRED_SETS = pending = []
    try:
        for ob in observers:
            ob.render_quantization_config()
    finally:
        _DEFERRED_SETS = None
        _flush_deferred_sets(pending)


def _prefetch(observers: Sequence[BaseTensorObserver]) -> None:
    """Steps 1-2 of render_observers: every device-to-host copy and every search the renders of `observers` will need."""
    # 1. ranges
    pend = [(ob, ob.pending_range()) for ob in observers if getattr(ob, '_host_range', None) is None]
    pend = [(ob, r) for ob, r in pend if r is not None]
    if pend:
        flat = torch.cat([r.reshape(-1) for _, r in pend]).cpu().numpy()
        pos = 0
        for ob, r in pend:
            n = r.numel()
            ob.take_range(flat[pos: pos + n].reshape(tuple(r.shape)))
            pos += n
    # 2. searches
    groups: Dict[tuple, List[TorchHistObserver]] = {}
    for ob in observers:
        if isinstance(ob, TorchHistObserver) and ob._phase == 'Collating Hist' and ob._hist is not None:
            if ob._losses is not None or getattr(ob, '_best', None) is not None: continue        # searched already
            if ob._quant_cfg.policy.has_property(P.PER_TENSOR):
                groups.setdefault(ob.search_key(), []).append(ob)
    for key, obs in groups.items():
        hists = torch.stack([ob.histogram() for ob in obs])
        if key[0] == 'kl':
            if any(ob._quant_cfg.policy.has_property(P.ASYMMETRICAL) for ob in obs): continue   # raises at render
            losses = CUDA.KLLosses(hists, key[2]).cpu().numpy()
            for ob, l in zip(obs, losses): ob.take_losses(l)
        else:
            dev = hists.device
            hs = torch.tensor([ob._hist_scale for ob in obs], dtype=torch.float64, device=dev)
            mn = torch.tensor([ob._min for ob in obs], dtype=torch.float64, device=dev)
            best = CUDA.MseSearch(hists, hs, mn, key[2], key[3], key[4]).cpu().numpy()
            for ob, b in zip(obs, best): ob.take_best(b)
    # 2b. FP8 'floating' observers: ONE FloatScaleSearch launch for every config of the graph (per rounding policy) and
    #     ONE device->host copy of the arg-min indices, instead of 7 x 4 launches + a synchronisation per config
    fobs = [ob for ob in observers if isinstance(ob, DirectMSEObserver) and ob._collector]
    if fobs:
        by_round: Dict[int, list] = {}
        for ob in fobs:
            if getattr(ob, '_sse', None) is None:                            # (a data-parallel merge already left merged sums)
                by_round.setdefault(int(getattr(ob._quant_cfg.rounding, 'value', ob._quant_cfg.rounding)), []).append(ob)
        for r, obs in by_round.items():
            items = [ob.search_item() for ob in obs]
            sse = CUDA.FloatScaleSearch(items, DirectMSEObserver.SCALE_CANDIDATES, r)
            pos = 0
            for ob, it in zip(obs, items):
                rows = it[0].shape[0]
                ob.take_squared_errors(sse[pos: pos + rows], it[0].shape[1]); pos += rows
        means = [(ob._sse.reshape(-1, len(DirectMSEObserver.SCALE_CANDIDATES)) / ob._count.reshape(-1, 1)) for ob in fobs]
        best = torch.argmin(torch.cat(means), dim=-1).cpu().tolist()
        pos = 0
        for ob, m in zip(fobs, means):
            ob.take_best(best[pos: pos + m.shape[0]]); pos += m.shape[0]

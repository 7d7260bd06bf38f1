"""PPQ's ``USING_CUDA_KERNEL = False`` PyTorch-CPU path -- TEST / BASELINE INFRASTRUCTURE ONLY.

The reference's default (ppq/core/config.py:4) runs this path on the host: plain torch ops on all
host threads.  BASELINE.json's north star asks for exactly this to be timed next to the HIP kernels.
Each function below restates the torch expressions of the reference line by line (citations relative
to /root/reference); nothing here is imported by ``ppq_amd``.  When a staged copy of the reference is
importable (``oracle/_ref/ppq_stage``, tools/stage_reference.py -- git-ignored, never committed) the
functions of :mod:`oracle.reference_import` time the reference's own classes instead and bench.py
labels the number ``"reference"``; this module is the ``"reference-path"`` fallback and is pinned to
the reference by tests/test_oracle_golden.py (same tensors through both, equal results).

    fq_linear_t / fq_linear_c   qfunction/linear.py:27-32, 75-81  (ppq_tensor_round = torch.round for
                                ROUND_HALF_EVEN, utils/round.py:97-105)
    minmax_t / minmax_c         observer/range.py:91-98
    hist_sym / hist_asym        observer/range.py:175-188  (torch.histc)
    kl_scale                    observer/range.py:190-282 + measure/statistic.py:3-12
    mse_scale                   observer/range.py:422-520  (pure-Python loss loop)
    percentile                  observer/range.py:338-346  (torch.kthvalue)
"""
import time
from typing import Dict, List, Tuple

import torch


# ------------------------------------------------------------------------------------ fake quant
def fq_linear_t(tensor: torch.Tensor, scales: torch.Tensor, offsets: torch.Tensor, quant_min: int, quant_max: int):
    """qfunction/linear.py:27-32."""
    tensor = torch.round(tensor / scales) + offsets
    tensor = torch.clamp(tensor, quant_min, quant_max)
    return (tensor - offsets) * scales


def fq_linear_c(tensor: torch.Tensor, scales: torch.Tensor, offsets: torch.Tensor, channel_axis: int,
                quant_min: int, quant_max: int):
    """qfunction/linear.py:75-81."""
    shape = [1 if axis != channel_axis else -1 for axis in range(tensor.ndim)]
    scale, offset = scales.view(shape), offsets.view(shape)
    tensor = torch.round(tensor / scale) + offset
    tensor = torch.clamp(tensor, quant_min, quant_max)
    return (tensor - offset) * scale


# ------------------------------------------------------------------------------------- observers
def minmax_t(value: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """observer/range.py:91-92."""
    return value.min().reshape(shape=[1, ]), value.max().reshape(shape=[1, ])


def minmax_c(value: torch.Tensor, channel_axis: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """observer/range.py:93-98."""
    channelwise_view = value.transpose(dim0=0, dim1=channel_axis).unsqueeze(-1)
    channelwise_view = torch.flatten(channelwise_view, start_dim=1)
    return (torch.min(channelwise_view, dim=1, keepdim=True)[0], torch.max(channelwise_view, dim=1, keepdim=True)[0])


def hist_sym(value: torch.Tensor, hist: torch.Tensor, hist_scale: float) -> None:
    """observer/range.py:186-187: hist += histc(abs(value), bins, 0, hist_scale * bins)."""
    bins = hist.numel()
    hist += torch.histc(torch.abs(value), bins, min=0, max=hist_scale * bins).int()


def hist_asym(value: torch.Tensor, hist: torch.Tensor, vmin: float, vmax: float) -> None:
    """observer/range.py:178-179."""
    hist += torch.histc(value, hist.numel(), min=vmin, max=vmax).int()


def percentile(value: torch.Tensor, q: float = 0.9999) -> Tuple[torch.Tensor, torch.Tensor]:
    """observer/range.py:338-346 (CPU branch): two kthvalue calls on the flattened tensor."""
    numel = value.numel()
    min_idx, max_idx = int(numel * (1 - q)), int(numel * q)
    min_idx = max(0, min_idx) + 1
    max_idx = min(max_idx, numel - 1) + 1
    flat = value.flatten()
    return torch.kthvalue(flat, k=max_idx, dim=0)[0].view(1, -1), torch.kthvalue(flat, k=min_idx, dim=0)[0].view(1, -1)


def torch_KL_divergence(hist: torch.Tensor, ref_hist: torch.Tensor, eps=1e-30) -> float:
    """measure/statistic.py:3-12."""
    if hist.ndim != 1 or ref_hist.ndim != 1:
        raise ValueError('Only 1 dimension tensor can compute KL divergence with another tensor.')
    return torch.dot(hist.double(), torch.log10(hist.double() + eps) - torch.log10(ref_hist.double() + eps)).item()


def kl_scale(histogram: torch.Tensor, hist_bins: int, hist_scale: float, num_of_bits: int = 8,
             scale_threshold: float = 1e-8) -> float:
    """observer/range.py:226-276 (per-tensor symmetrical, no power-of-2)."""
    histogram = histogram.to('cpu').float().clone()
    losses, quant_bins = [], 2 ** (num_of_bits - 1)
    histogram[: int(hist_bins * .002)] = 0
    histogram[int(hist_bins * .002)] = 1
    hist_sum = torch.sum(histogram)
    for bin_range in range(quant_bins, hist_bins + quant_bins - 1, quant_bins):
        p_hist = torch.zeros(size=(bin_range, ), dtype=torch.float)
        p_hist[: bin_range].copy_(histogram[: bin_range])
        p_hist[bin_range - 1] += torch.sum(histogram[bin_range:])
        p_hist = p_hist / hist_sum
        expand_ratio = int(bin_range / quant_bins)
        q_hist = histogram[: bin_range].clone()
        q_hist = q_hist.reshape((quant_bins, expand_ratio))
        positive_map = q_hist > 0
        positive_cnt = positive_map.sum(axis=1, keepdim=True)
        positive_cnt[positive_cnt == 0] = 1
        q_hist = torch.div(q_hist.sum(axis=1, keepdim=True), positive_cnt)
        q_hist = q_hist.repeat([1, expand_ratio])
        q_hist = q_hist * positive_map
        q_hist = q_hist / torch.sum(q_hist)
        q_hist = q_hist.flatten()
        losses.append({'kl': torch_KL_divergence(p_hist, q_hist), 'bin_range': bin_range})
    best_bin_range = sorted(losses, key=lambda x: x['kl'])[0]['bin_range']
    scale = (best_bin_range / hist_bins) * hist_scale * (hist_bins / quant_bins)
    return max(scale, scale_threshold)


def compute_mse_loss(histogram: list, start: int, step: int, end: int) -> float:
    """observer/range.py:431-454: the pure-Python loss the reference runs when its kernels are off."""
    num_of_elements = sum(histogram)
    loss = 0
    for idx, bin in enumerate(histogram):
        if idx < start:
            error = ((start - idx - 1) + 0.5)
        elif idx > end:
            error = ((idx - end) + 0.5)
        else:
            l_idx = (idx - start) % step
            r_idx = step - l_idx - 1
            if l_idx == r_idx:
                error = (l_idx + 0.25)
            else:
                l_err = (l_idx + 0.5)
                r_err = (r_idx + 0.5)
                error = min(l_err, r_err)
        loss += (bin * error * error) / num_of_elements
    return loss


def mse_range(histogram: torch.Tensor, hist_bins: int, hist_scale: float, vmin: float, quant_min: int, quant_max: int,
              symmetrical: bool, start_stride: int = 8) -> Tuple[float, float]:
    """observer/range.py:465-518: the candidate sweep; returns (range_min, range_max)."""
    histogram = histogram.to('cpu').float()
    num_of_quant_levels = (quant_max - quant_min) + 1
    losses = []
    if not symmetrical:
        for start in range(0, hist_bins, start_stride):
            if (start * hist_scale) + vmin > 0: break
            for step in range(1, hist_bins // num_of_quant_levels + 1):
                end = start + num_of_quant_levels * step
                if end > (hist_bins + num_of_quant_levels): break
                loss = compute_mse_loss(histogram=histogram.tolist(), start=start, step=step, end=end)
                losses.append({'mse': loss, 'start': start, 'end': end})
        best = sorted(losses, key=lambda x: x['mse'])[0]
        return (best['start'] * hist_scale) + vmin, (best['end'] * hist_scale) + vmin
    for step in range(1, hist_bins // num_of_quant_levels + 1):
        end = num_of_quant_levels * step
        if end > (hist_bins + num_of_quant_levels): break
        loss = compute_mse_loss(histogram=histogram.tolist(), start=0, step=step, end=end)
        losses.append({'mse': loss, 'end': end})
    best = sorted(losses, key=lambda x: x['mse'])[0]
    return -(best['end'] * hist_scale), (best['end'] * hist_scale)


# ------------------------------------------------------------------------- whole calibration loop
def calibrate(graph, batches: List[torch.Tensor], bins: int = 2048) -> Dict[str, float]:
    """RuntimeCalibrationPass('kl') of a harness graph on the host exactly as the reference's CPU executor
    would run it (optim/calibration.py:124-213, executor/torch.py:499-570): per forward every ACTIVATED
    per-channel weight config is fake-quantised (fq_linear_c), phase 1 collects min / max per batch
    (minmax_t) and renders the range, phase 2 accumulates torch.histc histograms, render = kl_scale.
    Returns {observed variable name: scale}; the graph's configs are not modified."""
    from ppq_amd.harness import _forward          # dense ops (torch CPU); no quantization code in there
    from .cpu_calibration import _is_fp32, _is_initial, _per_channel, _sym
    from . import ppq_oracle as O
    weights = {}
    for op in graph.operations.values():
        if not hasattr(op, 'config'): continue
        for cfg, var in op.config_with_variable:
            if var.is_parameter and not _is_fp32(cfg) and _per_channel(cfg):
                w = var.value.detach().cpu()
                mins, maxs = minmax_c(w, cfg.channel_axis)                       # range.py:93-98 + :117-129
                so = [O.minmax_to_scale_offset(float(a), float(b), cfg.quant_min, cfg.quant_max, _sym(cfg), f32_inputs=True)
                      for a, b in zip(mins.flatten().numpy(), maxs.flatten().numpy())]
                weights[var.name] = (w, torch.tensor([s for s, _ in so], dtype=torch.float32),
                                     torch.tensor([o for _, o in so], dtype=torch.float32), cfg)
    observed = {}
    for op in graph.operations.values():
        if not hasattr(op, 'config'): continue
        for cfg, var in op.config_with_variable:
            if not var.is_parameter and _is_initial(cfg): observed[var.name] = cfg
    mins = {n: [] for n in observed}; maxs = {n: [] for n in observed}
    hists = {n: torch.zeros(bins, dtype=torch.int32) for n in observed}
    hist_scale = {}

    def observe(name, t, phase):
        if phase == 1:
            a, b = minmax_t(t); mins[name].append(a); maxs[name].append(b)
        else: hist_sym(t, hists[name], hist_scale[name])

    def run(phase):
        for batch in batches:
            values = {next(iter(graph.inputs)): batch.cpu()}
            for op in graph.operations.values():
                xs = []
                for v in op.inputs:
                    if v.is_parameter:
                        if v.name in weights:
                            w, s, o, cfg = weights[v.name]
                            xs.append(fq_linear_c(w, s, o, cfg.channel_axis, cfg.quant_min, cfg.quant_max))
                        else: xs.append(v.value.detach().cpu())
                    else: xs.append(values[v.name])
                for v, x in zip(op.inputs, xs):
                    if v.name in observed and not v.is_parameter and v.source_op is None: observe(v.name, x, phase)
                y = _forward(op, xs)
                values[op.outputs[0].name] = y
                if op.outputs[0].name in observed: observe(op.outputs[0].name, y, phase)

    with torch.no_grad():
        run(1)
        for n in observed:
            mn = torch.min(torch.cat(mins[n], dim=0)).item(); mx = torch.max(torch.cat(maxs[n], dim=0)).item()
            hist_scale[n] = float(max(abs(mx), abs(mn))) / bins                 # range.py:294-301
        run(2)
        return {n: kl_scale(hists[n], bins, hist_scale[n], observed[n].num_of_bits) for n in observed}


def timed_calibrate(graph, batches, bins=2048):
    t0 = time.perf_counter()
    scales = calibrate(graph, batches, bins)
    return time.perf_counter() - t0, scales


# --------------------------------------------------------------------------------- per-op table
def _median_ms(fn, warmup=1, runs=20, budget_s=4.0) -> float:
    for _ in range(warmup): fn()
    ts = []
    t_all = time.perf_counter()
    for _ in range(runs):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > budget_s and len(ts) >= 3: break
    ts.sort()
    return ts[len(ts) // 2] * 1e3


def op_table(bins: int = 2048, tensors=('A', 'B', 'Bx32')) -> list:
    """per-op median milliseconds and effective GB/s (algorithmic bytes) of the
    reference's CPU path on A = randn(1,3,224,224), B = randn(1,512,56,56), Bx32 (seed 0), all host threads."""
    shapes = {'A': (1, 3, 224, 224), 'B': (1, 512, 56, 56), 'Bx32': (32, 512, 56, 56)}
    g = torch.Generator().manual_seed(0)
    rows = []
    with torch.no_grad():
        for name in tensors:
            x = torch.randn(*shapes[name], generator=g)
            # Bx32 rotates over 6 copies (1.2 GB between two uses of a buffer), like tools/microbench.py on the GPU: a single
            # 205 MB tensor stays resident in the host's L3 and the row would report cache, not DRAM, bandwidth
            pool = [x] + ([x.clone() for _ in range(5)] if name == 'Bx32' else [])
            turn = [0]

            def nxt(pool=pool, turn=turn):
                turn[0] += 1
                return pool[turn[0] % len(pool)]
            n, C = x.numel(), x.shape[1]
            s1 = torch.tensor(float(x.abs().max()) * 2 / 255); o1 = torch.tensor(0.0)
            sc = torch.rand(C, generator=g) * 0.05 + 0.01; oc = torch.randint(0, 255, [C], generator=g).float()
            hs = float(x.abs().max()) / bins
            lo, hi = float(x.min()), float(x.max())
            hist = torch.zeros(bins, dtype=torch.int32)
            cases = [('fq_linear_t', 8, lambda: fq_linear_t(nxt(), s1, o1, -128, 127)),
                     ('fq_linear_c', 8, lambda: fq_linear_c(nxt(), sc, oc, 1, 0, 255)),
                     ('hist_sym_t', 4, lambda: hist_sym(nxt(), hist, hs)),
                     ('hist_asym_t', 4, lambda: hist_asym(nxt(), hist, lo, hi)),
                     ('minmax_t', 4, lambda: minmax_t(nxt())),
                     ('minmax_c', 4, lambda: minmax_c(nxt(), 1))]
            if name != 'Bx32': cases.append(('percentile', 4, lambda: percentile(x)))
            for op, bpe, fn in cases:
                ms = _median_ms(fn)
                rows.append({'op': op, 'tensor': name, 'ms': round(ms, 4), 'GBps': round(bpe * n / ms / 1e6, 2)})
        x = torch.randn(1, 512, 56, 56, generator=g)
        hist = torch.histc(x.abs(), bins, min=0, max=float(x.abs().max())).int()
        hs = float(x.abs().max()) / bins
        rows.append({'op': 'kl_search', 'tensor': f'int32[{bins}]', 'ms': round(_median_ms(lambda: kl_scale(hist, bins, hs), runs=5), 3)})
        ah = torch.histc(x, bins, min=float(x.min()), max=float(x.max())).int()
        ahs = (float(x.max()) - float(x.min())) / bins
        t0 = time.perf_counter(); mse_range(ah, bins, ahs, float(x.min()), 0, 255, False); t1 = time.perf_counter()
        rows.append({'op': 'mse_search_asym (pure Python)', 'tensor': f'int32[{bins}]', 'ms': round((t1 - t0) * 1e3, 1)})
    return rows

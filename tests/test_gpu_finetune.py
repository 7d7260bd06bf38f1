"""GPU tests of the training-based passes (LearnedStepSizePass, BiasCorrectionPass on BASELINE config 5's
policy).  Collected LAST (tests/conftest.py): what an optimizer ends up with after hundreds of Adam steps is not
reproducible across boxes (the activation-scale gradients are ~1e-8, the order of Adam's eps, and the vendor
convolutions pick algorithms per box), so nothing here asserts HOW MUCH a loss fell.  Asserted instead: the block
structure, the keep / withdraw contract (a block that ended worse is restored bit for bit, a kept block changed),
that nothing keeps requires_grad / .grad / a delegate behind, finiteness -- box-independent by construction.  The
arithmetic of one training step (tensors and gradients of the first optimizer step) is pinned against the
reference's own pass in tests/test_gpu_reference.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _int4_weights(graph):
    for op in graph.operations.values():                        # weights -> int4 [-8, 7]
        for cfg, var in op.config_with_variable:
            if var.is_parameter and cfg.state.value == 1:
                cfg.num_of_bits, cfg.quant_min, cfg.quant_max = 4, -8, 7


def _snapshot(graph):
    """Every tensor a finetuning pass may touch: parameters, scales, offsets."""
    snap = {}
    for op in graph.operations.values():
        for v in op.inputs:
            if v.is_parameter and isinstance(v.value, torch.Tensor): snap[('p', v.name)] = v.value.detach().clone()
        if hasattr(op, 'config'):
            for i, (c, v) in enumerate(op.config_with_variable):
                if isinstance(c.scale, torch.Tensor): snap[('s', op.name, i)] = c.scale.detach().clone()
                if isinstance(c.offset, torch.Tensor): snap[('o', op.name, i)] = c.offset.detach().clone()
    return snap


def _block_keys(block):
    keys = []
    for op in block.rps:
        for v in op.inputs:
            if v.is_parameter and isinstance(v.value, torch.Tensor): keys.append(('p', v.name))
        if hasattr(op, 'config'):
            for i, (c, v) in enumerate(op.config_with_variable):
                if c.dominated_by is not c: continue         # an OVERLAPPED config reads its root's scale: another op owns (and may train) it
                if isinstance(c.scale, torch.Tensor): keys.append(('s', op.name, i))
                if isinstance(c.offset, torch.Tensor): keys.append(('o', op.name, i))
    return keys


def _check_contract(graph, ex, before, blocks, report):
    """The keep / withdraw contract of training.py:812-826, block by block, and a clean exit."""
    after = _snapshot(graph)
    assert set(after) == set(before)
    assert [str(b) for b in blocks] == [r[0] for r in report]
    for block, (_, pre, post) in zip(blocks, report):
        keys = _block_keys(block)
        same = all(torch.equal(before[k], after[k]) for k in keys)
        assert np.isfinite(pre) and np.isfinite(post) and pre >= 0 and post >= 0, (str(block), pre, post)
        if post > pre: assert same, f'{block}: ended worse ({pre} -> {post}) but was not restored'
        elif pre > 0: assert not same, f'{block}: kept ({pre} -> {post}) but nothing was trained'
    for t in after.values(): assert torch.isfinite(t).all()
    for op in graph.operations.values():                       # nothing is left trainable
        for v in op.inputs:
            if v.is_parameter and isinstance(v.value, torch.Tensor): assert not v.value.requires_grad and v.value.grad is None, v.name
        if hasattr(op, 'config'):
            for c, v in op.config_with_variable:
                for t in (c.scale, c.offset):
                    if isinstance(t, torch.Tensor): assert not t.requires_grad and t.grad is None, (op.name, v.name)
    assert not ex._delegates


def test_learned_step_size_finetune_int4():
    """BASELINE config 5 in miniature: INT4 per-channel weights + INT8 activations on a small CNN, scales / weights
    trained through CuLSQ (forward fake-quant kernels, backward LSQ kernels) with the reference's default depth limit."""
    from ppq_amd import harness
    from ppq_amd.blocks import split_graph_into_blocks
    from ppq_amd.calibration import RuntimeCalibrationPass
    from ppq_amd.lsq import LearnedStepSizePass
    graph = harness.small_cnn_graph(seed=5, width=16)
    harness.quantize_graph(graph, 'minmax')
    _int4_weights(graph)
    ex = harness.TorchExecutor(graph, DEV)
    harness.ParameterQuantizePass().optimize(graph)
    g = torch.Generator().manual_seed(7)
    batches = [torch.rand(8, 3, 24, 24, generator=g).to(DEV) for _ in range(8)]
    RuntimeCalibrationPass().optimize(graph, dataloader=batches, executor=ex, calib_steps=8)
    before = _snapshot(graph)
    p = LearnedStepSizePass(steps=100, lr=1e-3)
    assert p.block_size == 5                                   # the reference's default (training.py:711)
    pre, post = p.optimize(graph, batches, ex)
    blocks = split_graph_into_blocks(graph, graph.topological_sort(), 5)
    assert len(blocks) >= 2
    _check_contract(graph, ex, before, blocks, p.report)
    assert 0 < post <= pre
    out = ex.forward(batches[0])[0]
    assert torch.isfinite(out).all()
    # the whole graph as ONE block is refused when there is more than one end point
    multi = harness.yolov6s_graph(seed=0)
    harness.quantize_graph(multi, 'minmax')
    with pytest.raises(ValueError):
        LearnedStepSizePass(steps=1, block_size=None).optimize(multi, batches, harness.TorchExecutor(multi, DEV))


def test_lsq_gamma_term_is_the_reference_one():
    """training.py:793-798: loss += gamma * mse(w, Q(w)) with Q(w) = PPQLinearQuantFunction, NOT detached and NOT the
    delegator: under its straight-through backward (dy passes unmasked, qfunction/linear.py:48-50) the two branches cancel
    exactly, so the term shows in the loss value and moves no weight.  (Rounds 1-2 detached Q(w), which pulled the weights
    onto the grid -- a divergence)"""
    from ppq_amd import LinearQuantizationConfig
    from ppq_amd.core import QuantizationStates
    from ppq_amd.qfunction import PPQuantFunction
    cfg = LinearQuantizationConfig(symmetrical=True, quant_min=-8, quant_max=7, num_of_bits=4, channel_axis=0)
    cfg.scale = torch.full([4], 0.1, device=DEV); cfg.offset = torch.zeros(4, device=DEV)
    cfg.state = QuantizationStates.ACTIVATED
    w = torch.linspace(-1.5, 1.5, 4 * 6, device=DEV).reshape(4, 6).requires_grad_(True)
    loss = torch.mean(torch.square(w - PPQuantFunction(w, cfg)))
    loss.backward()
    assert float(loss) > 0 and torch.all(w.grad == 0)


def _lsq_run(batches, group, steps=4):
    from ppq_amd import harness
    from ppq_amd.calibration import RuntimeCalibrationPass
    from ppq_amd.lsq import LearnedStepSizePass
    graph = harness.small_cnn_graph(seed=5, width=16)
    harness.quantize_graph(graph, 'minmax')
    for op in graph.operations.values():
        for cfg, var in op.config_with_variable:
            if var.is_parameter and cfg.state.value == 1: cfg.num_of_bits, cfg.quant_min, cfg.quant_max = 4, -8, 7
    ex = harness.TorchExecutor(graph, DEV)
    harness.ParameterQuantizePass().optimize(graph)
    g = torch.Generator().manual_seed(7)
    calib = [torch.rand(8, 3, 24, 24, generator=g).to(DEV) for _ in range(8)]       # same calibration on every rank
    RuntimeCalibrationPass(check_steps=False).optimize(graph, dataloader=calib, executor=ex, calib_steps=8)
    p = LearnedStepSizePass(steps=steps, lr=1e-2, optimizer=torch.optim.SGD, process_group=group, block_size=None)
    pre, post = p.optimize(graph, [b.to(DEV) for b in batches], ex)
    torch.cuda.synchronize()
    # numpy (pickled by value): torch tensors on an mp.Queue travel as shared-memory handles that die with the worker
    scales = [c.scale.detach().reshape(-1).cpu().numpy() for op in graph.operations.values()
              for c, v in op.config_with_variable if c.state.value == 4]
    weights = [v.value.detach().cpu().numpy() for v in graph.variables.values() if v.is_parameter and v.value.dim() == 4]
    return pre, post, scales, weights


def _lsq_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(77)
        full = [torch.rand(8, 3, 24, 24, generator=g) for _ in range(4)]
        shard = [b[rank * 4:(rank + 1) * 4] for b in full]                         # each rank: half of every batch
        q.put((rank, _lsq_run(shard, dist.group.WORLD)))
    finally:
        dist.destroy_process_group()


def test_data_parallel_lsq_equals_big_batch():
    """LearnedStepSizePass(process_group=...): two ranks, each finetuning on half of every batch with ONE
    flat gradient all-reduce per step, stay in lock step bit for bit and follow the trajectory of one
    process on the full batches."""
    import torch.multiprocessing as mp
    g = torch.Generator().manual_seed(77)
    full = [torch.rand(8, 3, 24, 24, generator=g) for _ in range(4)]
    pre, post, scales, weights = _lsq_run(full, None)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + 7) % 2000)
    procs = [ctx.Process(target=_lsq_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs: p.join(timeout=60)
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1]                       # averaged losses: same decision
    assert res[0][0] == pytest.approx(pre, rel=1e-4)
    # LSQ scales its step-size gradient by 1/sqrt(numel * qmax) (linear.cu:299,402), and an activation has
    # half the elements on each rank: the data-parallel trajectory is the average of per-rank LSQ gradients,
    # close to -- not identical with -- the big-batch one.  Weight gradients are plain means and agree.
    assert res[0][1] <= res[0][0]
    for r in (0, 1):
        for a, b in zip(res[r][2], scales): assert np.allclose(a, b, rtol=2e-2, atol=1e-7)
        for a, b in zip(res[r][3], weights): assert np.allclose(a, b, rtol=1e-2, atol=1e-4)
    for a, b in zip(res[0][2], res[1][2]): assert np.array_equal(a, b)             # ranks stay in lock step
    for a, b in zip(res[0][3], res[1][3]): assert np.array_equal(a, b)


def test_yolov6s_int4_blockwise_lsq_and_bias_correction():
    """BASELINE config 5 on the YOLOv6-s-like detector (56 convolutions, 17 M parameters, 6 outputs): INT4
    per-channel weights + INT8 activations, calibrated, then BiasCorrectionPass (block_size 1: 56 blocks) and the
    block-wise LearnedStepSizePass (block_size 5: 27 TrainableBlocks incl. the SPPF fan-out that closes at its
    Concat) through the HIP forward / LSQ-backward kernels.  Box-independent assertions only (see the module
    docstring); the loss figures are printed for the log."""
    from ppq_amd import harness
    from ppq_amd.bias_correction import BiasCorrectionPass
    from ppq_amd.blocks import split_graph_into_blocks
    from ppq_amd.calibration import RuntimeCalibrationPass
    from ppq_amd.lsq import LearnedStepSizePass
    graph = harness.yolov6s_graph(seed=3)
    harness.quantize_graph(graph, 'minmax')
    _int4_weights(graph)
    ex = harness.TorchExecutor(graph, DEV)
    harness.ParameterQuantizePass().optimize(graph)
    g = torch.Generator().manual_seed(9)
    batches = [torch.rand(2, 3, 160, 160, generator=g).to(DEV) for _ in range(8)]
    RuntimeCalibrationPass().optimize(graph, dataloader=batches, executor=ex, calib_steps=8)
    outs = list(graph.outputs)
    quantable = [o for o in graph.operations.values() if hasattr(o, 'config')]
    # dequantised operations compute with the parameters stored at quantisation time (IR/quantize.py:124-141): the ORIGINAL network
    for o in quantable: o.dequantize()
    fp_ref = [ex.forward(b, outs) for b in batches[:4]]
    for o in quantable: o.restore_quantize_state()

    def error_vs_original():
        qt = [ex.forward(b, outs) for b in batches[:4]]
        num = sum(float(torch.sum((q - f) ** 2)) for fs, qs in zip(fp_ref, qt) for f, q in zip(fs, qs))
        return num / sum(float(torch.sum(f ** 2)) for fs in fp_ref for f in fs)
    e0 = error_vs_original()
    before = _snapshot(graph)
    bc = BiasCorrectionPass(steps=8, block_size=1)
    bc.optimize(graph, dataloader=batches, executor=ex)
    after = _snapshot(graph)
    assert len(bc.report) == 56 and all(post <= pre for _, pre, post in bc.report)
    changed = [k for k in before if not torch.equal(before[k], after[k])]
    assert changed and all(k[0] == 'p' and before[k].dim() == 1 for k in changed)      # biases only
    e1 = error_vs_original()
    before = _snapshot(graph)
    lsq = LearnedStepSizePass(steps=20, lr=1e-4, block_size=5)
    pre, post = lsq.optimize(graph, batches, ex)
    blocks = split_graph_into_blocks(graph, graph.topological_sort(), 5)
    assert len(blocks) == len(lsq.report) == 27
    _check_contract(graph, ex, before, blocks, lsq.report)
    assert 0 < post <= pre
    e2 = error_vs_original()
    assert np.isfinite([e0, e1, e2]).all()
    print(f'output error vs the original network: calibrated {e0:.4f}, bias-corrected {e1:.4f}, LSQ {e2:.4f}; '
          f'block loss {pre:.4f} -> {post:.4f}; kept {sum(1 for _, a, b in lsq.report if b <= a)}/27')


def _lsq_variant(steps, **kw):
    from ppq_amd import harness
    from ppq_amd.calibration import RuntimeCalibrationPass
    from ppq_amd.lsq import LearnedStepSizePass
    torch.manual_seed(0)
    graph = harness.small_cnn_graph(seed=5, width=16)
    harness.quantize_graph(graph, 'minmax')
    _int4_weights(graph)
    ex = harness.TorchExecutor(graph, DEV)
    harness.ParameterQuantizePass().optimize(graph)
    g = torch.Generator().manual_seed(7)
    batches = [torch.rand(8, 3, 24, 24, generator=g).to(DEV) for _ in range(4)]
    RuntimeCalibrationPass(check_steps=False).optimize(graph, dataloader=batches, executor=ex, calib_steps=4)
    p = LearnedStepSizePass(steps=steps, lr=1e-3, **kw)
    p.optimize(graph, batches, ex)
    return graph, ex, p, _snapshot(graph)


def test_grouped_weight_launches_equal_the_per_tensor_lsq_path():
    """LSQWeightGroup (ONE forward launch + ONE backward launch for all weight delegators of a block) against the per-tensor
    CuLSQ_LC path, both eager, ONE optimizer step per block: every trained tensor agrees to float summation order of the scale
    gradients (the weights' grad_x is bit-identical, Adam's first step is lr * sign(g): identical unless a gradient is within
    rounding of zero), and the pass reports that it really grouped."""
    _g, _, p_ref, ref = _lsq_variant(1, group_weights=False, use_hip_graph=False)
    _, _, p_grp, grp = _lsq_variant(1, group_weights=True, use_hip_graph=False)
    assert p_ref.stats['grouped_weights'] == 0 and p_grp.stats['grouped_weights'] >= 2
    assert [r[0] for r in p_ref.report] == [r[0] for r in p_grp.report]
    for key in ref:                          # one Adam step moves an element by at most lr = 1e-3 (sign flips of ~0 gradients: 2 lr)
        assert (ref[key] - grp[key]).abs().max() <= 2.1e-3, key
    first = _block_keys(__import__('ppq_amd.blocks', fromlist=['x']).split_graph_into_blocks(_g, _g.topological_sort(), 5)[0])
    exact = sum(torch.equal(ref[k], grp[k]) for k in first)
    # same inputs, same gradients up to the vendor convolutions' run-to-run rounding: most tensors come out identical
    assert exact >= (len(first) + 1) // 2, f'first block: only {exact} of {len(first)} tensors identical'
    (n, a, b), (_, c, d) = p_ref.report[0], p_grp.report[0]
    assert abs(a - c) <= 1e-6 * max(a, 1e-12) + 1e-12, (n, a, c)               # the first block's pre-loss sees identical tensors


def test_grouped_activation_backward_equals_the_per_tensor_lsq_path():
    """LSQActivationGroup (one backward launch per activation delegator + ONE ppqhip_lsq_finish_multi per step) against CuLSQ_LT
    (main + finish launch per delegator), both eager with the weights grouped alike, 3 steps per block: the scale gradients
    are the SAME kernels summed in the same order, so the pass's trained tensors agree like two runs of the same path do --
    up to the vendor convolutions' run-to-run rounding (see the grouped-weight test): first pre-loss equal, most tensors of
    the first block identical, nothing off by more than lr-sized steps; and the pass reports that it grouped."""
    _g, _, p_ref, ref = _lsq_variant(3, group_activations=False, use_hip_graph=False)
    _, _, p_grp, grp = _lsq_variant(3, group_activations=True, use_hip_graph=False)
    assert p_ref.stats['grouped_activations'] == 0 and p_grp.stats['grouped_activations'] >= 2
    assert [r[0] for r in p_ref.report] == [r[0] for r in p_grp.report]
    for key in ref: assert (ref[key] - grp[key]).abs().max() <= 3 * 2.1e-3, key
    first = _block_keys(__import__('ppq_amd.blocks', fromlist=['x']).split_graph_into_blocks(_g, _g.topological_sort(), 5)[0])
    exact = sum(torch.equal(ref[k], grp[k]) for k in first)
    assert exact >= (len(first) + 1) // 2, f'first block: only {exact} of {len(first)} tensors identical'
    (n, a, b), (_, c, d) = p_ref.report[0], p_grp.report[0]
    assert abs(a - c) <= 1e-6 * max(a, 1e-12) + 1e-12, (n, a, c)


def test_hip_graph_replay_of_the_block_step_equals_eager_steps():
    """One block step captured as a HIP graph and replayed (use_hip_graph=True) against the same steps issued eagerly, 6 steps
    per block, grouped weights in both: the keep / withdraw contract holds, the graph path really replayed, and the trained
    tensors agree to the tolerance of lr-sized Adam steps on near-zero gradients (the capturable Adam evaluates the same
    formula with device-side step counts)."""
    from ppq_amd.blocks import split_graph_into_blocks
    from ppq_amd.lsq import LearnedStepSizePass
    graph_e, ex_e, p_e, eager = _lsq_variant(6, use_hip_graph=False)
    graph_g, ex_g, p_g, graphed = _lsq_variant(6, use_hip_graph=True)
    assert p_g.stats['graph_failures'] == 0 and p_g.graph_error is None and not p_g._graph_broken, (p_g.stats, p_g.graph_error)
    assert not hasattr(LearnedStepSizePass, '_graph_broken')             # per instance: one failed capture does not poison later passes
    assert p_g.stats['graph_blocks'] == len(p_g.report) and p_g.stats['graph_replays'] == 5 * len(p_g.report), p_g.stats
    assert p_e.stats['graph_blocks'] == 0 and p_e.stats['eager_steps'] == 6 * len(p_e.report)
    for key in eager:
        assert (eager[key] - graphed[key]).abs().max() <= 6 * 2.1e-3, key
    for k, ((n, a, b), (_, c, d)) in enumerate(zip(p_e.report, p_g.report)):
        # the first block starts from identical tensors
        if k == 0: assert abs(a - c) <= 1e-6 * max(a, 1e-12) + 1e-12, (n, a, c)
        assert np.isfinite(c) and np.isfinite(d)           # (later blocks: an earlier keep / withdraw decision may differ -- see the file header)
    out = ex_g.forward(torch.rand(2, 3, 24, 24, device=DEV))[0]
    assert torch.isfinite(out).all()
    for op in graph_g.operations.values():                      # nothing is left trainable after the graphed pass either
        for v in op.inputs:
            if v.is_parameter and isinstance(v.value, torch.Tensor): assert not v.value.requires_grad and v.value.grad is None
    assert not ex_g._delegates


def test_prefix_cache_block_inputs_equal_the_full_forward_collection():
    """blocks.PrefixCache (quantised block inputs computed incrementally, TorchExecutor.forward_cached) against collect()'s full
    forward from the graph inputs (training.py:224-298), block after block on the YOLOv6-s-like graph (fan-outs, Concat, 6
    outputs), with the block's weights and scales CHANGED between two blocks the way training changes them: every block input of
    every batch is bit for bit the full forward's.
    The vendor convolutions must be repeatable for that statement to be testable: MIOpen answers the first calls of a shape with
    whatever algorithm its search has reached (tools/prefix_diag.py: two IDENTICAL full forwards differed by up to 14 INT8 steps on a
    fresh box, then settled), so the executor is warmed first and deterministic algorithms are requested; a comparison is only
    counted when the full forward reproduces itself at that moment, and nearly all of them must be."""
    from ppq_amd import harness
    from ppq_amd.blocks import PrefixCache, collect, split_graph_into_blocks
    from ppq_amd.calibration import RuntimeCalibrationPass
    graph = harness.yolov6s_graph(seed=1)
    harness.quantize_graph(graph, 'minmax')
    _int4_weights(graph)
    ex = harness.TorchExecutor(graph, DEV)
    harness.ParameterQuantizePass().optimize(graph)
    g = torch.Generator().manual_seed(3)
    batches = [torch.rand(2, 3, 96, 96, generator=g).to(DEV) for _ in range(2)]
    RuntimeCalibrationPass(check_steps=False).optimize(graph, dataloader=batches, executor=ex, calib_steps=2)
    blocks = split_graph_into_blocks(graph, graph.topological_sort(), 5)
    assert len(blocks) >= 20
    was = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        for _ in range(3):
            for b in batches: ex.forward(b)                       # let MIOpen settle on its kernels for every shape of the graph
        prefix = PrefixCache(graph, ex, batches)
        fake_targets = [{} for _ in batches]
        compared = inconclusive = 0
        mismatched = []
        for k, block in enumerate(blocks):
            got = prefix.inputs_of(block)
            want, _ = collect(graph, block, ex, batches, fp_outputs=fake_targets)
            again, _ = collect(graph, block, ex, batches, fp_outputs=fake_targets)
            for a, b, c in zip(got, want, again):
                assert set(a) == set(b)
                for n in a:
                    if not torch.equal(b[n], c[n]): inconclusive += 1; continue       # the yardstick itself moved
                    compared += 1
                    if not torch.equal(a[n], b[n]): mismatched.append((k, str(block), n, float((a[n] - b[n]).abs().max())))
            with torch.no_grad():                                   # "train" the block: weights and activation scales move
                for op in block.rps:
                    for v in op.inputs:
                        if v.is_parameter and isinstance(v.value, torch.Tensor) and v.value.dim() == 4: v.value.mul_(1.0 + 0.01 * (k % 3))
                    if hasattr(op, 'config'):
                        for c, v in op.config_with_variable:
                            if not v.is_parameter and isinstance(c.scale, torch.Tensor) and c.state.value == 4 and c.dominated_by is c:
                                c.scale.mul_(1.02)
            prefix.invalidate(block)
    finally:
        torch.backends.cudnn.deterministic = was
    # a stale cache entry would show at every block behind the missed invalidation; a residual algorithm switch of the vendor
    # library between the cached and the fresh computation shows once or twice
    from conftest import record_parity_residue
    record_parity_residue('vendor_conv_nonrepeatable', 'test_prefix_cache_block_inputs_equal_the_full_forward_collection',
                          compared=compared, inconclusive=inconclusive, mismatched=len(mismatched), worst=[m[3] for m in mismatched])
    assert compared >= 40 and inconclusive <= compared // 4 and len(mismatched) <= 2, (compared, inconclusive, mismatched[:5])


def test_passive_bias_policy_rides_a_passive_delegator_through_lsq():
    """The integer platforms' bias policy (PPLQuantizer.py:54-66: 32-bit symmetric per-channel, PASSIVE_INIT) on the harness:
    ParameterQuantizePass leaves the bias alone, calibration runs with it unquantised, PassiveParameterQuantizePass
    (optim/parameters.py:13-153) gives it scale = weight scale x input scale and state PASSIVE, the executor then fake-quantises
    it through the per-channel kernel, and the LSQ pass puts a PASSIVE LSQDelegator on it (algorithm/training.py:329: trainable
    value with a backup, scale not trainable) -- the keep / withdraw contract holds for the biases too."""
    from ppq_amd import harness
    from ppq_amd.blocks import split_graph_into_blocks
    from ppq_amd.calibration import RuntimeCalibrationPass
    from ppq_amd.core import QuantizationStates as S
    from ppq_amd.lsq import LearnedStepSizePass
    from ppq_amd.parameters import PassiveParameterQuantizePass
    graph = harness.small_cnn_graph(seed=5, width=16)
    harness.quantize_graph(graph, 'minmax', passive_bias=True)
    ex = harness.TorchExecutor(graph, DEV)
    harness.ParameterQuantizePass().optimize(graph)
    g = torch.Generator().manual_seed(7)
    batches = [torch.rand(8, 3, 24, 24, generator=g).to(DEV) for _ in range(8)]
    biases = [(op, op.config.input_quantization_config[2]) for op in graph.operations.values()
              if hasattr(op, 'config') and op.type in ('Conv', 'Gemm') and len(op.inputs) == 3]
    assert len(biases) >= 3 and all(c.state == S.PASSIVE_INIT for _, c in biases)
    fp_bias_out = ex.forward(batches[0])[0].clone()
    RuntimeCalibrationPass().optimize(graph, dataloader=batches, executor=ex, calib_steps=8)
    assert all(c.state == S.PASSIVE_INIT for _, c in biases)
    p = PassiveParameterQuantizePass(); p.optimize(graph)
    assert p.unresolved == []
    for op, c in biases:
        i_cfg, w_cfg, _ = op.config.input_quantization_config
        assert c.state == S.PASSIVE and torch.equal(c.scale, w_cfg.scale * i_cfg.scale) and float(c.offset.abs().sum()) == 0
        assert c.scale.numel() == op.inputs[2].value.numel()
    out = ex.forward(batches[0])[0]
    assert torch.isfinite(out).all() and not torch.equal(out, fp_bias_out)
    before = _snapshot(graph)
    lsq = LearnedStepSizePass(steps=20, lr=1e-3)
    lsq.optimize(graph, batches, ex)
    _check_contract(graph, ex, before, split_graph_into_blocks(graph, graph.topological_sort(), 5), lsq.report)


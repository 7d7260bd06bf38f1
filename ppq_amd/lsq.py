"""LSQ (learned step size) fake-quant with gradients for scale -- the "next" row of the scope table.

Mirror of ppq/quantization/algorithm/training.py:17-90 (``CuLSQ_LT`` / ``CuLSQ_LC``) and :318-421
(``LSQDelegator``): forward = the fake-quant kernels, backward = ``CUDA.LinearQuantize_T_B / _C_B``
(``dx = dy * 1[in range]``, ``ds = sum(...) * rsqrt(n * (qmax - qmin))`` per tensor,
``rsqrt(n * qmax)`` per channel -- ppq/csrc/cuda/linear.cu:284-433).  A delegator instance can be
registered on PPQ's executor with ``TorchExecutor.register_quantize_delegate(config, delegator)``
(ppq/executor/torch.py:296-323); ``BlockwiseFinetune``-style passes then train scales through these
kernels.  HIP path only: a CPU tensor raises.
"""
from typing import List

import warnings

import torch
from torch.autograd import Function

from .core import QuantizationProperty as P
from .core import QuantizationStates, rounding_value, state_value
from .blocks import COMPUTING_OP, block_forward, torch_mean_square_error
from .calibration import QuantizationOptimizationPass
from .ffi import CUDA
from .qfunction import PPQLinearQuantFunction, PPQuantFunction, _as_1d


class CuLSQ_LT(Function):
    @ staticmethod
    def forward(ctx, tensor, scales, offsets, quant_min: int, quant_max: int, rounding) -> torch.Tensor:
        r = rounding_value(rounding)
        quantized = CUDA.LinearQuantize_T(tensor=tensor, scales=scales, offsets=offsets, minimum=quant_min,
                                          maximum=quant_max, rounding=r)
        ctx.save_for_backward(tensor, scales, offsets)
        ctx._quant_params = [quant_min, quant_max, r]
        return quantized

    @ staticmethod
    def backward(ctx, dy: torch.Tensor):
        dy = dy.contiguous()
        tensor, scales, offsets = ctx.saved_tensors
        quant_min, quant_max, rounding = ctx._quant_params
        dx, ds = CUDA.LinearQuantize_T_B(tensor, scales, offsets, dy, quant_min, quant_max, rounding)
        return dx, ds.reshape(scales.shape), None, None, None, None


class CuLSQ_LC(Function):
    @ staticmethod
    def forward(ctx, tensor, scales, offsets, channel_axis: int, quant_min: int, quant_max: int,
                rounding) -> torch.Tensor:
        r = rounding_value(rounding)
        quantized = CUDA.LinearQuantize_C(tensor=tensor, scales=scales, offsets=offsets, channel_axis=channel_axis,
                                          minimum=quant_min, maximum=quant_max, rounding=r)
        ctx.save_for_backward(tensor, scales, offsets)
        ctx._quant_params = [quant_min, quant_max, channel_axis, r]
        return quantized

    @ staticmethod
    def backward(ctx, dy: torch.Tensor):
        dy = dy.contiguous()
        tensor, scales, offsets = ctx.saved_tensors
        quant_min, quant_max, channel_axis, rounding = ctx._quant_params
        dx, ds = CUDA.LinearQuantize_C_B(tensor, scales, offsets, dy, quant_min, quant_max, channel_axis, rounding)
        return dx, ds.reshape(scales.shape), None, None, None, None, None


class _GroupedLSQ(Function):
    """``CuLSQ_LC`` for a weight that belongs to an :class:`LSQWeightGroup`: forward hands out the slice the group's ONE
    forward launch already filled, backward only stashes ``dy`` -- weight and scale are autograd LEAVES, so nothing further
    back needs their gradients during the sweep; :meth:`LSQWeightGroup.flush` computes all of them in ONE launch after it."""
    @ staticmethod
    def forward(ctx, tensor, scales, offsets, group, slot: int) -> torch.Tensor:
        ctx.group, ctx.slot = group, slot
        return group.outputs[slot].detach()                 # a fresh alias: no history from the previous step

    @ staticmethod
    def backward(ctx, dy: torch.Tensor):
        ctx.group.dys[ctx.slot] = dy.contiguous()
        return None, None, None, None, None


class LSQWeightGroup:
    """All per-channel weight delegators of one block as ONE forward launch (``ppqhip_fq_linear_multi``: the arena of
    ``ffi.LinearQuantizePlan`` re-filled from the current weights / scales at the start of every step) and ONE backward
    launch (``ppqhip_fq_linear_c_bwd_multi``) -- where the per-tensor path issues a forward kernel, a memset and a backward
    kernel per weight per step (algorithm/training.py:49-90 does so in the reference).  Values are those of ``CuLSQ_LC``
    weight by weight (tests/test_gpu_finetune.py).  The gradient buffers are allocated once and installed as ``.grad``
    (added to an existing one: the gamma term's straight-through gradient arrives through autograd), so a captured HIP graph
    of the step finds them at fixed addresses."""
    def __init__(self, members):
        from .ffi import LinearQuantizePlan
        self.members = members                              # [(delegator, config, var)]
        cfg0 = members[0][1]
        self.rounding = rounding_value(cfg0.rounding)
        self.plan = LinearQuantizePlan([(v.value, c.scale, c.offset, c.channel_axis, c.quant_min, c.quant_max)
                                        for _, c, v in members], rounding=self.rounding)
        self.gx = [torch.empty_like(v.value, memory_format=torch.contiguous_format) for _, _, v in members]
        self.gs = [torch.empty_like(c.scale) for _, c, _ in members]
        self.dys = [None] * len(members)
        self.outputs = None
        self.launches = 0
        for k, (d, _, _) in enumerate(members): d.group, d.slot = self, k

    @ staticmethod
    def eligible(delegator, config, var) -> bool:
        from .ffi import LinearQuantizePlan
        pol = config.policy
        return (var.is_parameter and isinstance(var.value, torch.Tensor) and var.value.is_cuda and var.value.is_leaf
                and var.value.dtype == torch.float32 and pol.has_property(P.LINEAR) and pol.has_property(P.PER_CHANNEL)
                and not pol.has_property(P.DYNAMIC) and not delegator.passive
                and isinstance(config.scale, torch.Tensor) and isinstance(config.offset, torch.Tensor) and config.scale.is_leaf
                and var.value.is_contiguous()
                and LinearQuantizePlan.accepts(var.value, config.scale, config.offset, config.channel_axis))

    @ classmethod
    def build(cls, delegators: dict) -> list:
        """One group per rounding policy over the eligible delegators of a block (at least two members: a lone weight gains
        nothing over its own launch)."""
        by_round = {}
        for cfg, d in delegators.items():
            if cls.eligible(d, cfg, d.var): by_round.setdefault(rounding_value(cfg.rounding), []).append((d, cfg, d.var))
        return [cls(m) for m in by_round.values() if len(m) >= 2]

    def prepare(self) -> None:
        """Start of a step: fake-quantise every member weight (ONE launch)."""
        self.outputs = self.plan.run()
        self.dys = [None] * len(self.members)
        self.launches += 1

    def flush(self) -> None:
        """End of the backward sweep: grad_x / grad_s of every member that received a ``dy`` (ONE launch), installed as
        the weights' and scales' ``.grad``."""
        live = [k for k, dy in enumerate(self.dys) if dy is not None]
        if not live: return
        m = self.members
        CUDA.LinearQuantize_C_B_Multi([m[k][2].value for k in live], [m[k][1].scale for k in live], [m[k][1].offset for k in live],
                                      [self.dys[k] for k in live], [m[k][1].quant_min for k in live],
                                      [m[k][1].quant_max for k in live], [m[k][1].channel_axis for k in live], self.rounding,
                                      grad_xs=[self.gx[k] for k in live], grad_ss=[self.gs[k] for k in live])
        self.launches += 1
        for k in live:
            for leaf, g in ((m[k][2].value, self.gx[k]), (m[k][1].scale, self.gs[k])):
                if not leaf.requires_grad: continue
                if leaf.grad is None: leaf.grad = g
                elif leaf.grad is not g: leaf.grad.add_(g)
                # (leaf.grad is g: zero_grad(set_to_none=False) kept the buffer and nothing else was accumulated -- the kernel
                #  overwrote it, which is the gradient)
        self.dys = [None] * len(m)

    def release(self) -> None:
        for d, _, _ in self.members: d.group, d.slot = None, None


class _DeferredScaleLSQ(Function):
    """``CuLSQ_LT`` for an activation that belongs to an :class:`LSQActivationGroup`: forward is the same fake-quant launch,
    backward computes grad_x (the producer's backward needs it now) with ``ppqhip_fq_linear_t_bwd_main`` and leaves the
    partial sums of the SCALE gradient in the member's own buffer -- the scale is an autograd leaf, nothing further back waits
    for its gradient; :meth:`LSQActivationGroup.flush` finishes all of them in ONE launch after the sweep."""
    @ staticmethod
    def forward(ctx, tensor, scales, offsets, quant_min: int, quant_max: int, rounding, group, slot: int) -> torch.Tensor:
        r = rounding_value(rounding)
        quantized = CUDA.LinearQuantize_T(tensor=tensor, scales=scales, offsets=offsets, minimum=quant_min, maximum=quant_max, rounding=r)
        ctx.save_for_backward(tensor, scales, offsets)
        ctx._quant_params = [quant_min, quant_max, r]
        ctx.group, ctx.slot = group, slot
        return quantized

    @ staticmethod
    def backward(ctx, dy: torch.Tensor):
        tensor, scales, offsets = ctx.saved_tensors
        quant_min, quant_max, rounding = ctx._quant_params
        dx = ctx.group.backward_main(ctx.slot, tensor, scales, offsets, dy.contiguous(), quant_min, quant_max, rounding)
        return dx, None, None, None, None, None, None, None


class LSQActivationGroup:
    """The per-tensor ACTIVATION delegators of one block: each backward is one launch (grad_x + partial sums) instead of the
    per-tensor path's two (``ppqhip_fq_linear_t_bwd`` = main + finish), and ONE ``ppqhip_lsq_finish_multi`` launch per step
    produces every scale gradient -- bit-identical to ``CuLSQ_LT`` (same kernels, same summation order).  Buffers (partials,
    gradients) are allocated once per member and reused, so a captured HIP graph of the step finds them at fixed addresses."""
    def __init__(self, members):
        self.members = members                              # [(delegator, config)]
        self.partials = [None] * len(members)
        self.gs = [torch.empty_like(_as_1d(c.scale)) for _, c in members]
        self.live = {}                                      # slot -> (numel, qmin, qmax) of this step's backward
        self.launches = 0
        for k, (d, _) in enumerate(members): d.act_group, d.act_slot = self, k

    @ staticmethod
    def eligible(delegator, config) -> bool:
        pol = config.policy
        return (not delegator.is_parameter and pol.has_property(P.LINEAR) and pol.has_property(P.PER_TENSOR)
                and not pol.has_property(P.DYNAMIC) and isinstance(config.scale, torch.Tensor) and config.scale.is_cuda
                and config.scale.is_leaf and config.scale.numel() == 1 and isinstance(config.offset, torch.Tensor))

    @ classmethod
    def build(cls, delegators: dict):
        members = [(d, cfg) for cfg, d in delegators.items() if cls.eligible(d, cfg)]
        return cls(members) if len(members) >= 2 else None

    def backward_main(self, slot: int, tensor, scales, offsets, dy, quant_min: int, quant_max: int, rounding: int) -> torch.Tensor:
        need = CUDA.lsq_t_partials(tensor.numel())
        if slot in self.live or self.partials[slot] is None or self.partials[slot].numel() < need:
            if slot in self.live:                               # the same config quantises two tensors of the block: finish the first
                self.flush()
                leaf = self.members[slot][1].scale              # .. and keep its gradient apart from the buffer the second one overwrites
                if leaf.grad is not None and leaf.grad.data_ptr() == self.gs[slot].data_ptr(): leaf.grad = leaf.grad.clone()
            if self.partials[slot] is None or self.partials[slot].numel() < need:
                self.partials[slot] = torch.empty(need, dtype=torch.float32, device=tensor.device)
        dx = CUDA.LinearQuantize_T_B_Main(tensor, scales, offsets, dy, quant_min, quant_max, rounding, self.partials[slot])
        self.live[slot] = (tensor.numel(), quant_min, quant_max)
        return dx

    def flush(self) -> None:
        """End of the backward sweep: the scale gradient of every member whose backward ran (ONE launch), ADDED to ``.grad``."""
        if not self.live: return
        slots = sorted(self.live)
        CUDA.LSQ_Finish_Multi([self.partials[k] for k in slots], [self.live[k][0] for k in slots], [self.live[k][1] for k in slots],
                              [self.live[k][2] for k in slots], [self.gs[k] for k in slots])
        self.launches += 1
        for k in slots:
            leaf = self.members[k][1].scale
            if not leaf.requires_grad: continue
            g = self.gs[k].reshape(leaf.shape)              # a view of the member's own buffer (fixed address: graph replays)
            if leaf.grad is None: leaf.grad = g
            elif leaf.grad.data_ptr() != g.data_ptr(): leaf.grad.add_(g)
        self.live = {}

    def release(self) -> None:
        for d, _ in self.members: d.act_group, d.act_slot = None, None


class LSQDelegator:
    """training.py:318-421 (the TorchQuantizeDelegator protocol: ``__call__(tensor, config)``)."""
    def __init__(self, config, var, is_parameter_trainable: bool = True, is_scale_trainable: bool = True,
                 is_offset_trainable: bool = True) -> None:
        self.config = config
        self.is_parameter = var.is_parameter
        self.var = var
        self.policy = config.policy
        self.passive = state_value(config.state) == QuantizationStates.PASSIVE.value
        self.group, self.slot = None, None          # set by LSQWeightGroup: this weight rides the block's multi-tensor launches
        self.act_group, self.act_slot = None, None  # set by LSQActivationGroup: the scale gradient is finished after the sweep
        self.param_backup = None
        if self.is_parameter and is_parameter_trainable:
            # detached: a grad-tracked clone of a requires_grad leaf would create (and keep alive) the leaf's AccumulateGrad node
            # on whatever stream is current NOW -- the step then runs on another stream and a graph capture of it breaks
            self.param_backup = self.var.value.detach().clone()
        active = (state_value(config.state) == QuantizationStates.ACTIVATED.value and config.dominated_by == config)
        self.scale_backup, self.is_scale_trainable = None, False
        if is_scale_trainable:
            if (not config.policy.has_property(P.POWER_OF_2) and config.policy.has_property(P.LINEAR) and active
                    and isinstance(config.scale, torch.Tensor)):
                self.is_scale_trainable = True
                self.scale_backup = self.config.scale.detach().clone()
        self.offset_backup, self.is_offset_trainable = None, False
        if is_offset_trainable:
            if (not config.policy.has_property(P.SYMMETRICAL) and active and isinstance(config.offset, torch.Tensor)):
                self.is_offset_trainable = True
                self.offset_backup = self.config.offset.detach().clone()

    def trainable_tensors(self) -> List[torch.Tensor]:
        params = []
        if self.is_offset_trainable: params.append(self.config.offset)
        if self.is_scale_trainable: params.append(self.config.scale)
        if self.is_parameter: params.append(self.var.value)
        return params

    def withdraw(self) -> None:
        with torch.no_grad():
            if self.scale_backup is not None: self.config.scale.copy_(self.scale_backup)
            if self.offset_backup is not None: self.config.offset.copy_(self.offset_backup)
            if self.param_backup is not None: self.var.value.copy_(self.param_backup)

    def finalize(self) -> None:
        self.scale_backup = self.offset_backup = self.param_backup = None

    def __call__(self, tensor: torch.Tensor, config) -> torch.Tensor:
        if config.policy.has_property(P.LINEAR):
            if self.group is not None and self.group.outputs is not None and tensor is self.var.value:
                return _GroupedLSQ.apply(tensor, config.scale, config.offset, self.group, self.slot)
            if config.policy.has_property(P.PER_CHANNEL):
                return CuLSQ_LC.apply(tensor, config.scale, config.offset, config.channel_axis, config.quant_min,
                                      config.quant_max, config.rounding)
            elif config.policy.has_property(P.PER_TENSOR):
                if self.act_group is not None and torch.is_grad_enabled():
                    return _DeferredScaleLSQ.apply(tensor, _as_1d(config.scale), _as_1d(config.offset), config.quant_min,
                                                   config.quant_max, config.rounding, self.act_group, self.act_slot)
                return CuLSQ_LT.apply(tensor, _as_1d(config.scale), _as_1d(config.offset), config.quant_min,
                                      config.quant_max, config.rounding)
        elif config.policy.has_property(P.FLOATING):
            return PPQuantFunction(tensor=tensor, config=config)      # scale is not trainable for FP8
        raise ValueError('LSQDelegator: unsupported quantization policy')


class LearnedStepSizePass(QuantizationOptimizationPass):
    """ppq/quantization/optim/training.py:569-863: block-wise LSQ finetuning.

    The graph is cut into TrainableBlocks (ppq_amd/blocks.py; ``block_size`` = the reference's depth limit,
    default 5, training.py:711); per block: collect the FP32 block outputs (graph dequantised) and the
    quantised block inputs, put an LSQDelegator on every activated config of the block, run Adam over
    {weights, scales, offsets} through ``partial_graph_forward`` (forward fake-quant kernels, backward LSQ
    kernels), loss = MSE against the FP32 block output (+ gamma * weight quantisation error), withdraw when
    the block loss did not improve (training.py:728-826).  ``block_size=None`` treats the whole graph as ONE
    block; it is refused for graphs with more than one output (a block has ONE end point).

    Data-parallel finetuning (one process per GPU, each with its shard of the batches): the gradients of ALL
    trainable tensors of a block travel in ONE flat all-reduce per step (a few MB at most -- latency bound on
    xGMI, so never one collective per tensor); block losses are averaged so every rank takes the same keep /
    withdraw decision.  ``report`` = [(block, pre_loss, post_loss)]."""
    def __init__(self, name: str = 'PPQ LSQ Optimization', interested_layers: List[str] = None, steps: int = 500,
                 gamma: float = 0.0, is_scale_trainable: bool = True, lr: float = 5e-5, block_size: int = 5,
                 expire_device: str = 'cpu', collecting_device: str = 'cuda', loss_fn=None, optimizer=None, *,
                 process_group=None, group_weights: bool = True, use_hip_graph: bool = True, group_activations: bool = True,
                 fused_adam: bool = True):
        """The positional parameters are the reference's, in its order (training.py:700-713).  ``expire_device`` /
        ``collecting_device`` are accepted and unused: nothing is parked on the host, the block caches stay on the executor's
        device (288 GB of HBM).  ``loss_fn(y_pred, y_real)``: default = the reference's ``torch_mean_square_error``."""
        super().__init__(name=name)
        self.steps, self.lr, self.gamma, self.optimizer = steps, lr, gamma, optimizer
        if loss_fn is not None: self._loss = loss_fn
        self.process_group = process_group
        self.block_size = block_size
        self.interested_layers = interested_layers or []
        self.is_scale_trainable = is_scale_trainable
        # MI355X-side execution choices (no twin in the reference; the values trained are the per-tensor path's up to float
        # summation order): all weight delegators of a block in ONE forward + ONE backward launch (LSQWeightGroup), and the
        # optimizer step of a block captured ONCE as a HIP graph and replayed for its remaining steps (single process only)
        # use_hip_graph: on that path the default optimizer is Adam(capturable=True) -- its step count and bias corrections are
        # float32 DEVICE tensors instead of Python doubles, and it stays that optimizer for the eager steps of a block whose
        # capture failed -- so the trained values differ from the eager / reference `torch.optim.Adam` run by more than float
        # summation order: per step the bias-corrected step size differs in the last float32 bits (~1e-7 relative), and the
        # scale gradients here are ~1e-8 (the order of Adam's eps), so after a few steps individual scales may sit one
        # lr-sized step apart (test_hip_graph_replay_of_the_block_step_equals_eager_steps states the tolerance).  use_hip_graph=False is the
        # reference's optimizer bit for bit.
        self.group_weights = group_weights
        self.group_activations = group_activations     # per-tensor activation delegators: one backward launch each + ONE finish launch per step
        self.use_hip_graph = use_hip_graph
        self.fused_adam = fused_adam                   # on the HIP-graph path only (the eager path keeps the reference's optimizer)
        self._graph_broken = False              # a capture failed in THIS pass: its later blocks stay eager (reason: graph_error)
        self.graph_error = None
        self.capture_error_mode = 'global'
        self.incremental_inputs = True          # quantised block inputs from blocks.PrefixCache instead of a full forward per block
        self.max_blocks = None                  # a measuring aid: finetune only the first k blocks (bench.py's 500-step variant)
        self.profile_phases = False             # True: synchronise around the phases and fill phase_ms (a measuring aid)
        self.phase_ms = {}
        self.report = []
        self.stats = {'blocks': 0, 'graph_blocks': 0, 'graph_replays': 0, 'eager_steps': 0, 'grouped_weights': 0, 'grouped_activations': 0,
                      'graph_failures': 0}

    def _phase(self, name: str):
        import contextlib
        import time
        if not self.profile_phases: return contextlib.nullcontext()

        @ contextlib.contextmanager
        def timed():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            try: yield
            finally:
                torch.cuda.synchronize()
                self.phase_ms[name] = self.phase_ms.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return timed()

    def _world(self) -> int:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()): return 1
        return dist.get_world_size(self.process_group)

    def _average(self, tensors: List[torch.Tensor]) -> None:
        world = self._world()
        if world == 1 or not tensors: return
        import torch.distributed as dist
        flat = torch.cat([t.reshape(-1) for t in tensors])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.process_group)
        flat /= world
        pos = 0
        for t in tensors:
            n = t.numel(); t.copy_(flat[pos: pos + n].view_as(t)); pos += n

    @ staticmethod
    def _loss(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        return torch.mean(torch.square(a.flatten(1) - b.flatten(1)))          # torch_mean_square_error

    def _block_loss(self, block, qt_inputs, fp_outputs, executor) -> float:
        from .blocks import compute_block_loss
        loss = torch.tensor([compute_block_loss(block, qt_inputs, fp_outputs, executor, self._loss)],
                            dtype=torch.float32, device=next(iter(qt_inputs[0].values())).device)
        self._average([loss])
        return float(loss)

    @ staticmethod
    def enable_block_gradient(block) -> None:
        """training.py:150-168: every float parameter of the block (Clip bounds excepted) and every config scale."""
        for op in block.rps:
            for var in list(op.inputs) + list(op.outputs):
                if var.is_parameter and isinstance(var.value, torch.Tensor):
                    if op.type == 'Clip': continue
                    if var.value.dtype == torch.float32: var.value.requires_grad = True
            if hasattr(op, 'config'):
                for cfg, _ in op.config_with_variable:
                    if isinstance(cfg.scale, torch.Tensor) and cfg.scale.is_floating_point(): cfg.scale.requires_grad = True

    @ staticmethod
    def disable_block_gradient(block) -> None:
        """training.py:170-183 -- plus the offsets: nothing of the block keeps ``requires_grad`` or a ``.grad`` after
        the pass, whether it was handed to the optimizer or not (a later ``.numpy()`` / export must not raise)."""
        for op in block.rps:
            for var in list(op.inputs) + list(op.outputs):
                if var.is_parameter and isinstance(var.value, torch.Tensor) and var.value.is_leaf:
                    var.value.requires_grad = False; var.value.grad = None
            if hasattr(op, 'config'):
                for cfg, _ in op.config_with_variable:
                    for t in (cfg.scale, cfg.offset):
                        if isinstance(t, torch.Tensor) and t.is_leaf and t.is_floating_point():
                            t.requires_grad = False; t.grad = None

    def _graphable(self, qt_inputs, fp_outputs, tensors) -> bool:
        """A block's optimizer step is captured as a HIP graph when nothing in it needs the host between steps: one
        process (a gloo / RCCL gradient all-reduce is kept out of graphs), the default Adam (built ``capturable``), device
        tensors, every batch of the same shape (the graph reads two static staging buffers), and enough steps to pay for the
        capture (step 0 runs eagerly and warms MIOpen / the allocator, the capture itself executes nothing)."""
        if not self.use_hip_graph or self._graph_broken or self.optimizer is not None or self._world() != 1: return False
        if self.steps < 3 or not tensors or not all(t.is_cuda for t in tensors): return False
        for dicts in (qt_inputs, fp_outputs):
            first = {k: (tuple(v.shape), v.dtype) for k, v in dicts[0].items()}
            if any({k: (tuple(v.shape), v.dtype) for k, v in d.items()} != first for d in dicts[1:]): return False
        return True

    def _train_with_graph(self, train_step, qt_inputs, fp_outputs) -> int:
        """Step 0 eagerly on a side stream, ONE capture of the same step on that stream, replays for the rest; returns the
        number of steps done (0 or 1 when the capture failed: the caller finishes eagerly and later blocks do not retry)."""
        n = len(qt_inputs)
        static_in = {k: torch.empty_like(v) for k, v in qt_inputs[0].items()}
        static_fp = {k: torch.empty_like(v) for k, v in fp_outputs[0].items()}

        def load(i: int) -> None:
            for k, v in static_in.items(): v.copy_(qt_inputs[i][k])
            for k, v in static_fp.items(): v.copy_(fp_outputs[i][k])

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with self._phase('graph_step0_eager'):
            with torch.cuda.stream(side):               # the library's scratch / workspaces are per stream: warm THIS one
                load(0)
                train_step(static_in, static_fp)
            torch.cuda.current_stream().wait_stream(side)
        self.stats['eager_steps'] += 1
        # capture_begin / capture_end directly: the torch.cuda.graph() context manager empties the caching allocator on entry,
        # which with one capture per block (27 for the YOLOv6-s-like graph) means re-allocating every buffer 27 times.
        # (Each graph keeps its own private pool: it dies with the graph at the end of this function.)
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        error = None
        with self._phase('graph_capture'), torch.cuda.stream(side):
            try:
                graph.capture_begin(capture_error_mode=self.capture_error_mode)
            except Exception as e:
                error = e
            else:
                try:
                    train_step(static_in, static_fp)
                except Exception as e:                  # keep the FIRST error: ending a broken capture raises again
                    error = e
                try:
                    graph.capture_end()
                except Exception as e:
                    error = error or e
        if error is not None:                           # a call that cannot be captured: finish eagerly, stop trying
            import traceback
            self._graph_broken = True
            self.graph_error = self.stats['graph_error'] = \
                f'{type(error).__name__}: {str(error)[:600]} @ ' + ' <- '.join(
                    f'{f.name}:{f.lineno}' for f in reversed(traceback.extract_tb(error.__traceback__)[-6:]))
            self.stats['graph_failures'] += 1
            torch.cuda.synchronize()
            return 1
        self.stats['graph_blocks'] += 1
        with self._phase('graph_replays'):
            for step in range(1, self.steps):
                load(step % n)
                graph.replay()
                self.stats['graph_replays'] += 1
            torch.cuda.current_stream().synchronize()   # the graph's private pool dies with `graph`: no replay may be in flight
        return self.steps

    def finetune(self, block, executor, qt_inputs, fp_outputs):
        """training.py:728-826 for one block."""
        self.enable_block_gradient(block)
        with self._phase('pre_loss'):
            pre_loss = self._block_loss(block, qt_inputs, fp_outputs, executor)
        delegators, tensors = {}, []
        for op in block.rps:
            if not hasattr(op, 'config'): continue
            if op.type in COMPUTING_OP or op.type in {'Add', 'Mul'}:          # a lone Add / Mul trains its constant too
                for var in op.inputs:
                    if var.is_parameter and isinstance(var.value, torch.Tensor): tensors.append(var.value)
            for cfg, var in op.config_with_variable:
                if state_value(cfg.state) in (QuantizationStates.ACTIVATED.value, QuantizationStates.PASSIVE.value):
                    d = LSQDelegator(config=cfg, var=var, is_scale_trainable=self.is_scale_trainable)
                    tensors.extend(d.trainable_tensors())
                    executor.register_quantize_delegate(cfg, d)
                    delegators[cfg] = d
        uniq, seen = [], set()
        for t in tensors:                              # offsets never carry requires_grad: CuLSQ has no offset gradient
            if t.requires_grad and id(t) not in seen: seen.add(id(t)); uniq.append(t)
        if not uniq:
            for cfg in delegators: executor.remove_quantize_delegate(cfg)
            self.disable_block_gradient(block)
            return 0.0, 0.0
        # parameters whose config carries no delegator (an FP32 bias when the quantizer leaves biases unquantised -- with
        # quantize_graph(passive_bias=True) + PassiveParameterQuantizePass the bias config is PASSIVE and its delegator holds
        # the backup, as in the reference) are trained too -- keep their own backup
        covered = {id(d.var.value) for d in delegators.values() if d.is_parameter}
        loose = [(t, t.detach().clone()) for t in uniq if id(t) not in covered
                 and not any(t is c.scale or t is c.offset for c in delegators)]
        names = [v.name for v in block.ep.outputs]
        if len(qt_inputs) == 0: raise ValueError('Dataset is empty.')
        groups = LSQWeightGroup.build(delegators) if (self.group_weights and uniq[0].is_cuda) else []
        act_group = LSQActivationGroup.build(delegators) if (self.group_activations and uniq[0].is_cuda) else None
        self.stats['blocks'] += 1
        self.stats['grouped_weights'] += sum(len(g.members) for g in groups)
        self.stats['grouped_activations'] += len(act_group.members) if act_group is not None else 0
        graphable = self._graphable(qt_inputs, fp_outputs, uniq)
        if self.optimizer is not None: opt = self.optimizer(uniq, lr=self.lr)
        elif graphable:
            opt = None
            if self.fused_adam:                        # ONE multi-tensor kernel per step instead of the foreach form's 6-8 small ones
                try: opt = torch.optim.Adam(uniq, lr=self.lr, capturable=True, fused=True)
                except (RuntimeError, ValueError, TypeError) as e:
                    self.stats['fused_adam_error'] = f'{type(e).__name__}: {str(e)[:200]}'
                    if not getattr(LearnedStepSizePass, '_warned_fused_adam', False):
                        LearnedStepSizePass._warned_fused_adam = True
                        warnings.warn(f'LearnedStepSizePass: fused Adam unavailable ({self.stats["fused_adam_error"]}); using the foreach capturable form')
            if opt is None: opt = torch.optim.Adam(uniq, lr=self.lr, capturable=True)
            else: self.stats['fused_adam_blocks'] = self.stats.get('fused_adam_blocks', 0) + 1
        else: opt = torch.optim.Adam(uniq, lr=self.lr)

        def train_step(qt_input, fp_output) -> None:
            opt.zero_grad()
            # a step that aborted between a backward and its flush (a failed graph capture) must not leak half-built state into
            # this one: partial sums whose kernels never ran would be finished and ADDED to the scale gradients
            if act_group is not None: act_group.live.clear()
            for g in groups: g.prepare()
            with torch.enable_grad():
                outs = block_forward(executor, block.rps, qt_input, names, with_gradient=True)
                loss = sum(self._loss(y, fp_output[n]) for n, y in zip(names, outs))
                if self.gamma:
                    for op in block.rps:                                  # training.py:793-798 (the STE gradient passes)
                        if hasattr(op, 'config') and op.type in COMPUTING_OP:
                            w, wc = op.inputs[1].value, op.config.input_quantization_config[1]
                            # always the reference's own MSE over PPQLinearQuantFunction, whatever `loss_fn` is
                            loss = loss + torch_mean_square_error(w, PPQLinearQuantFunction(w, wc)) * self.gamma
            loss.backward()
            for g in groups: g.flush()
            if act_group is not None: act_group.flush()
            with torch.no_grad():
                self._average([t.grad for t in uniq if t.grad is not None])
            opt.step()

        done = self._train_with_graph(train_step, qt_inputs, fp_outputs) if graphable else 0
        with self._phase('eager_steps'):
            for step in range(done, self.steps):
                train_step(qt_inputs[step % len(qt_inputs)], fp_outputs[step % len(qt_inputs)])
                self.stats['eager_steps'] += 1
        for g in groups: g.outputs = None             # the arena is stale after the last optimizer step: per-tensor path from here
        with self._phase('post_loss'):
            post_loss = self._block_loss(block, qt_inputs, fp_outputs, executor)
        for g in groups: g.release()
        if act_group is not None: act_group.release()
        for cfg, d in delegators.items():
            if post_loss > pre_loss: d.withdraw()
            d.finalize()
            executor.remove_quantize_delegate(cfg)
        if post_loss > pre_loss:
            with torch.no_grad():
                for t, backup in loose: t.copy_(backup)
        self.disable_block_gradient(block)
        return pre_loss, post_loss

    def optimize(self, graph, dataloader, executor, collate_fn=None, **kwargs):
        from .blocks import TrainableBlock, collect, split_graph_into_blocks
        batches = [collate_fn(b) if collate_fn is not None else b for b in dataloader]
        if self.block_size is None:
            if len(graph.outputs) != 1:
                raise ValueError('LearnedStepSizePass(block_size=None) trains the whole graph against ONE output; '
                                 f'this graph has {len(graph.outputs)} -- pass a block_size')
            ops = graph.topological_sort()
            blocks = [TrainableBlock(sp=ops[0], ep=ops[-1], rps=ops)]
        else:
            blocks = split_graph_into_blocks(graph, graph.topological_sort(), self.block_size,
                                             interested_layers=self.interested_layers)
        if self.max_blocks is not None: blocks = blocks[: self.max_blocks]
        self.report = []
        # FP32 targets: the graph dequantised, i.e. (IR/quantize.py:124-141) computing with the parameters stored at
        # quantisation time -- what earlier blocks train does not move the targets of later ones, so the targets of EVERY
        # block come from ONE dequantised forward per batch (the reference runs that forward again for each block,
        # training.py:224-298: same values, 27 x the work on the YOLOv6-s-like graph; 288 GB of HBM hold them all)
        from .blocks import PrefixCache, collect_all_fp_outputs, supports_prefix_cache
        with self._phase('collect_fp_targets'):
            all_fp = collect_all_fp_outputs(graph, blocks, executor, batches) if self.incremental_inputs else None
        # quantised block inputs: incrementally (blocks.PrefixCache) -- the prefix of the graph runs once per batch overall
        # (on the reference's own executor, which has no forward_cached: a forward per block, as the reference does)
        prefix = PrefixCache(graph, executor, batches) if (self.incremental_inputs and supports_prefix_cache(executor)) else None
        for k, block in enumerate(blocks):
            with self._phase('collect_block_inputs'):
                targets = all_fp[k] if all_fp is not None else None       # None: too large to keep for all blocks -> per block
                if prefix is not None and targets is not None: qt_inputs, fp_outputs = prefix.inputs_of(block), targets
                else: qt_inputs, fp_outputs = collect(graph, block, executor, batches, fp_outputs=targets)
            if all_fp is not None: all_fp[k] = None
            pre_loss, post_loss = self.finetune(block, executor, qt_inputs, fp_outputs)
            if prefix is not None: prefix.invalidate(block)
            self.report.append((str(block), pre_loss, post_loss))
        if not self.report: return 0.0, 0.0
        return sum(r[1] for r in self.report), sum(min(r[1], r[2]) for r in self.report)

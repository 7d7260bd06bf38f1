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

import torch
from torch.autograd import Function

from .core import QuantizationProperty as P
from .core import QuantizationStates, rounding_value, state_value
from .ffi import CUDA
from .qfunction import PPQuantFunction, _as_1d


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


class LSQDelegator:
    """training.py:318-421 (the TorchQuantizeDelegator protocol: ``__call__(tensor, config)``)."""
    def __init__(self, config, var, is_parameter_trainable: bool = True, is_scale_trainable: bool = True,
                 is_offset_trainable: bool = True) -> None:
        self.config = config
        self.is_parameter = var.is_parameter
        self.var = var
        self.policy = config.policy
        self.passive = state_value(config.state) == QuantizationStates.PASSIVE.value
        self.param_backup = None
        if self.is_parameter and is_parameter_trainable:
            self.param_backup = self.var.value.clone()
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
            if config.policy.has_property(P.PER_CHANNEL):
                return CuLSQ_LC.apply(tensor, config.scale, config.offset, config.channel_axis, config.quant_min,
                                      config.quant_max, config.rounding)
            elif config.policy.has_property(P.PER_TENSOR):
                return CuLSQ_LT.apply(tensor, _as_1d(config.scale), _as_1d(config.offset), config.quant_min,
                                      config.quant_max, config.rounding)
        elif config.policy.has_property(P.FLOATING):
            return PPQuantFunction(tensor=tensor, config=config)      # scale is not trainable for FP8
        raise ValueError('LSQDelegator: unsupported quantization policy')


class LearnedStepSizePass:
    """The finetune loop of ppq/quantization/optim/training.py:728-826 (LearnedStepSizePass.finetune)
    for a graph treated as ONE trainable block: LSQDelegators on every activated config, Adam over
    {weights, scales, offsets}, MSE between the quantised and the FP32 block output (+ gamma * weight
    quantisation error), withdraw when the loss did not improve.  The reference's block splitting
    (BlockBuilder, training.py:191-315) is graph plumbing outside this package's scope."""
    def __init__(self, steps: int = 100, lr: float = 5e-5, gamma: float = 0.0, optimizer=None, process_group=None):
        self.steps, self.lr, self.gamma, self.optimizer = steps, lr, gamma, optimizer
        # data-parallel finetuning (one process per GPU, each with its shard of the batches): gradients of
        # ALL trainable tensors travel in ONE flat all-reduce per step (a few MB at most for a block --
        # latency bound on xGMI, so never one collective per tensor), block losses are averaged so every
        # rank takes the same keep / withdraw decision
        self.process_group = process_group

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

    def optimize(self, graph, dataloader, executor, **kwargs):
        ops = [op for op in graph.operations.values() if hasattr(op, 'config')]
        saved = {}
        # FP32 reference outputs: every config switched off (QuantableOperation.dequantize)
        for op in ops:
            for cfg, _ in op.config_with_variable:
                saved[cfg] = cfg.state
                if QuantizationStates.is_activated(cfg.state): cfg.state = QuantizationStates.FP32
        batches = list(dataloader)
        fp_outputs = [executor.forward(b)[0].detach() for b in batches]
        for cfg, st in saved.items(): cfg.state = st

        def block_loss():
            with torch.no_grad():
                loss = sum(self._loss(executor.forward(b)[0], f) for b, f in zip(batches, fp_outputs)) / len(batches)
                loss = loss.reshape(1).clone()
                self._average([loss])
                return float(loss)
        pre_loss = block_loss()
        delegators, tensors = {}, []
        for op in ops:
            if op.type in {'Conv', 'Gemm', 'ConvTranspose', 'MatMul', 'Add', 'Mul'}:
                for var in op.inputs:
                    if var.is_parameter:
                        var.value.requires_grad_(True); tensors.append(var.value)
            for cfg, var in op.config_with_variable:
                if state_value(cfg.state) in (QuantizationStates.ACTIVATED.value, QuantizationStates.PASSIVE.value):
                    for t in (cfg.scale, cfg.offset):
                        if isinstance(t, torch.Tensor) and t.is_floating_point(): t.requires_grad_(True)
                    d = LSQDelegator(config=cfg, var=var)
                    tensors.extend(d.trainable_tensors())
                    executor.register_quantize_delegate(cfg, d)
                    delegators[cfg] = d
        uniq, seen = [], set()
        for t in tensors:
            if t.requires_grad and id(t) not in seen: seen.add(id(t)); uniq.append(t)
        if not uniq:
            for cfg in delegators: executor.remove_quantize_delegate(cfg)
            return 0.0, 0.0
        opt = torch.optim.Adam(uniq, lr=self.lr) if self.optimizer is None else self.optimizer(uniq, lr=self.lr)
        for step in range(self.steps):
            b, f = batches[step % len(batches)], fp_outputs[step % len(batches)]
            opt.zero_grad()
            out = executor.forward_with_gradient(b)[0]
            loss = self._loss(out, f)
            if self.gamma:
                for op in ops:
                    if op.type in {'Conv', 'Gemm'}:
                        w, wc = op.inputs[1].value, op.config.input_quantization_config[1]
                        loss = loss + self._loss(w, PPQuantFunction(w, wc).detach()) * self.gamma
            loss.backward()
            with torch.no_grad():
                self._average([t.grad for t in uniq if t.grad is not None])
            opt.step()
        post_loss = block_loss()
        for cfg, d in delegators.items():
            if post_loss > pre_loss: d.withdraw()
            d.finalize()
            executor.remove_quantize_delegate(cfg)
        for t in uniq: t.requires_grad_(False)
        return pre_loss, post_loss

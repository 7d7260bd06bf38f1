"""Trainable blocks -- what the finetuning passes (LearnedStepSizePass, BiasCorrectionPass) iterate over.

Counterpart of ``TrainableBlock`` / ``BlockBuilder`` (ppq/quantization/algorithm/training.py:171-315) and of
``TrainingBasedPass.split_graph_into_blocks / collect / compute_block_loss``
(ppq/quantization/optim/training.py:177-335).  The block DEFINITION is the reference's (training.py:229-242):

    a block is a triple (S, E, M): S the input operation, E the output operation, M every operation on a
    path from S to E;  E lies on every path from S to a graph output, S on every path from a graph input to
    E;  the minimal block of an operation p is (p, p, {p});  depth(E) - depth(S) <= limit.

The SEARCH is this package's own: instead of the reference's recursive coherent-successor / blocking-node
walk over its SearchableGraph, one forward sweep in execution order grows the region reached from S and
records, after every operation it absorbs, whether the region is closed -- no value enters it except
through S, no value leaves it except through the operation just absorbed.  Each closed prefix is a block by
the definition above, and the last one within the depth limit is returned (the reference's walk stops at
the same operations: its coherent successor and blocking multi-input node are exactly the closure points).
"""
from typing import Dict, Iterable, List, Optional, Tuple

import torch

OPTIM_ADVOPT_GRAPH_MAXDEPTH = 4                                  # ppq/core/common.py
COMPUTING_OP = {'Conv', 'Gemm', 'ConvTranspose', 'MatMul', 'Attention', 'PPQBiasFusedMatMul'}      # ppq/core/common.py:55


class TrainableBlock:
    """training.py:171-188."""
    def __init__(self, sp, ep, rps: list) -> None:
        self.sp, self.ep, self.rps = sp, ep, rps

    def __str__(self) -> str:
        return f'[Graph Block from {self.sp.name} to {self.ep.name}]'


def upstream_operations(op) -> list:
    return [v.source_op for v in op.inputs if not v.is_parameter and v.source_op is not None]


def downstream_operations(op) -> list:
    out = []
    for v in op.outputs:
        for d in v.dest_ops:
            if d not in out: out.append(d)
    return out


class BlockBuilder:
    def __init__(self, graph, topo_order: Optional[list] = None) -> None:
        self.graph = graph
        self.op_orders = list(topo_order) if topo_order is not None else graph.topological_sort()
        self.index = {op.name: i for i, op in enumerate(self.op_orders)}
        self.depth: Dict[str, int] = {}
        for op in self.op_orders:                                  # training.py:300-315
            ups = upstream_operations(op)
            self.depth[op.name] = 0 if not ups else max(self.depth[u.name] for u in ups) + 1

    def build(self, op, limit: int) -> TrainableBlock:
        start = self.index[op.name]
        region = {op.name}
        members = [op]
        open_edges = {}                       # member name -> consumers not yet absorbed (graph outputs never close)
        graph_outputs = set(self.graph.outputs)

        def consumers(o):
            return {d.name for d in downstream_operations(o)}
        open_edges[op.name] = consumers(op)
        best = 1                              # members[:best] is the last closed prefix == the block
        for nxt in self.op_orders[start + 1:]:
            ups = [u.name for u in upstream_operations(nxt)]
            if not any(u in region for u in ups): continue        # a parallel branch that does not descend from S
            feeds_from_outside = any(u not in region for u in ups) or \
                any((not v.is_parameter) and v.source_op is None for v in nxt.inputs)
            if feeds_from_outside: break                          # S no longer dominates: nothing beyond is a block
            if self.depth[nxt.name] - self.depth[op.name] > limit: break
            region.add(nxt.name); members.append(nxt)
            open_edges[nxt.name] = consumers(nxt)
            closed = True
            for m in members[:-1]:
                if any(c not in region for c in open_edges[m.name]) or any(v.name in graph_outputs for v in m.outputs):
                    closed = False; break
            if closed: best = len(members)
        rps = members[:best]
        return TrainableBlock(sp=op, ep=rps[-1], rps=rps)


def split_graph_into_blocks(graph, executing_order: Optional[list] = None, blocksize: Optional[int] = None,
                            overlap: bool = False, interested_layers: Optional[List[str]] = None) -> List[TrainableBlock]:
    """optim/training.py:177-222: one block per (not yet visited) quantable computing op, in graph order."""
    if blocksize is None: blocksize = OPTIM_ADVOPT_GRAPH_MAXDEPTH
    builder = BlockBuilder(graph, executing_order)
    visited, blocks = set(), []
    for op in graph.operations.values():
        if op.name in visited and not overlap: continue
        if hasattr(op, 'config') and op.type in COMPUTING_OP:
            block = builder.build(op, blocksize)
            for o in block.rps: visited.add(o.name)
            blocks.append(block)
    if not interested_layers: return blocks
    return [b for b in blocks if any(o.name in interested_layers for o in b.rps)]


_TAKES_WITH_GRADIENT: Dict[type, bool] = {}


def block_forward(executor, operations, feed_dict, output_names, with_gradient: bool = False):
    """``executor.partial_graph_forward`` for this package's executor (which takes ``with_gradient``) AND for the reference's
    (executor/torch.py:654-682: no such parameter -- it records autograd history whenever grad mode is on), so that the
    block-wise passes run on either."""
    fn = executor.partial_graph_forward
    native = _TAKES_WITH_GRADIENT.get(type(executor))
    if native is None:
        import inspect
        native = _TAKES_WITH_GRADIENT[type(executor)] = 'with_gradient' in inspect.signature(fn).parameters
    if native: return fn(operations, feed_dict, output_names, with_gradient=with_gradient)
    device = getattr(executor, '_device', None)                  # the reference's own passes move the feeds themselves
    if device is not None: feed_dict = {k: v.to(device) for k, v in feed_dict.items()}        # (training.py:782)
    if with_gradient:
        with torch.enable_grad(): return fn(operations, feed_dict, output_names)
    with torch.no_grad(): return fn(operations, feed_dict, output_names)


def supports_prefix_cache(executor) -> bool:
    """PrefixCache needs ``forward_cached`` (this package's executor); on the reference's executor the passes collect the
    block inputs with a forward per block, as the reference does."""
    return hasattr(executor, 'forward_cached')


def torch_mean_square_error(y_pred: torch.Tensor, y_real: torch.Tensor) -> torch.Tensor:
    """ppq/quantization/measure/norm.py: mean over the batch of the per-sample mean squared error."""
    return torch.mean(torch.mean(torch.square(y_pred.flatten(1) - y_real.flatten(1)), dim=-1))


@ torch.no_grad()
def collect_fp_outputs(graph, blocks: List[TrainableBlock], executor, batches: Iterable) -> List[List[dict]]:
    """FP32 end-point outputs of EVERY block from one dequantised forward per batch (the weights as they are
    now): result[k][i] = {name: tensor} for block k, batch i."""
    quantable = [o for o in graph.operations.values() if hasattr(o, 'config')]
    names = [[v.name for v in b.ep.outputs] for b in blocks]
    flat = [n for ns in names for n in ns]
    for o in quantable: o.dequantize()
    out = [[] for _ in blocks]
    for b in batches:
        vals = dict(zip(flat, executor.forward(b, flat)))
        for k, ns in enumerate(names): out[k].append({n: vals[n].detach() for n in ns})
    for o in quantable: o.restore_quantize_state()
    return out


@ torch.no_grad()
def collect(graph, block: TrainableBlock, executor, batches: Iterable, fp_outputs: List[dict] = None) -> Tuple[List[dict], List[dict]]:
    """optim/training.py:224-298: FP32 outputs of the block's end point with the WHOLE graph dequantised,
    then the quantised inputs of its start point with the quantisation state restored (two forwards per
    batch; everything stays on the executor's device -- 288 GB of HBM hold the cache).  `fp_outputs`: use
    these targets instead of collecting them now."""
    quantable = [o for o in graph.operations.values() if hasattr(o, 'config')]
    if fp_outputs is None: fp_outputs = collect_fp_outputs(graph, [block], executor, batches)[0]
    feeds = [v for v in block.sp.inputs if not v.is_parameter]
    qt_inputs = []
    for b in batches:
        if all(v.name in graph.inputs for v in feeds):
            vals = [b if isinstance(b, torch.Tensor) else b[v.name] for v in feeds]
        else:
            vals = executor.forward(b, [v.name for v in feeds])
        qt_inputs.append({v.name: x.detach() for v, x in zip(feeds, vals)})
    return qt_inputs, fp_outputs


def collect_all_fp_outputs(graph, blocks: List['TrainableBlock'], executor, batches, max_resident_bytes: int = 64 << 30):
    """``collect_fp_outputs`` for every block at once when the targets fit ``max_resident_bytes`` of HBM (judged from the first
    batch), else None -- the caller then collects per block as the reference does.  The targets depend on the parameters stored
    at quantisation time only (IR/quantize.py:124-160), so one dequantised forward per batch serves all blocks."""
    batches = list(batches)
    if not blocks or not batches: return [[] for _ in blocks]
    first = collect_fp_outputs(graph, blocks, executor, batches[:1])
    per_batch = sum(t.numel() * t.element_size() for per_block in first for d in per_block for t in d.values())
    if per_batch * len(batches) > max_resident_bytes: return None
    rest = collect_fp_outputs(graph, blocks, executor, batches[1:]) if len(batches) > 1 else [[] for _ in blocks]
    return [a + b for a, b in zip(first, rest)]


class PrefixCache:
    """The quantised activations a block-wise pass needs as block inputs, computed INCREMENTALLY.

    The reference obtains the inputs of every block with a full forward from the graph inputs (training.py:224-298, per
    block and batch): O(blocks x graph) work.  The inputs of block k + 1 only differ from what was already computed for
    block k by the operations in between, so this cache keeps, per calibration batch, every quantised activation computed so
    far (``TorchExecutor.forward_cached``) and, when a block has been trained, drops what depends on it: the outputs of the
    block's operations and everything downstream.  Each operation of the prefix therefore runs once per batch in its final
    state -- same values as the reference's walk (a cached tensor never depends on a parameter / scale changed after it was
    computed), 4 forwards instead of 4 x 27 / 2 for the YOLOv6-s-like graph.  Everything stays on the device."""
    def __init__(self, graph, executor, batches):
        self.graph, self.executor, self.batches = graph, executor, list(batches)
        self.values: List[Dict[str, torch.Tensor]] = [dict() for _ in self.batches]

    @ torch.no_grad()
    def inputs_of(self, block: 'TrainableBlock') -> List[dict]:
        feeds = [v for v in block.sp.inputs if not v.is_parameter]
        names = [v.name for v in feeds]
        out = []
        for b, cache in zip(self.batches, self.values):
            vals = self.executor.forward_cached(b, names, cache)
            out.append({n: x.detach() for n, x in zip(names, vals)})
        return out

    def invalidate(self, block: 'TrainableBlock') -> None:
        """The block's parameters / scales changed: forget its outputs and everything computed from them -- and, with them
        gone, every tensor nothing will be computed from any more (see :meth:`prune`)."""
        dead, stack = set(), [v for op in block.rps for v in op.outputs]
        # a config OUTSIDE the block may read a scale the block just trained (TensorQuantizationConfig.dominated_by / master links,
        # core.py): the outputs of such an operation are stale too, wherever it sits in the graph (quantize_graph never
        # links across blocks today -- the guard costs one pass over the configs)
        owned = {id(cfg) for op in block.rps if hasattr(op, 'config') for cfg, _ in op.config_with_variable}
        inside = {id(op) for op in block.rps}
        for op in self.graph.operations.values():
            if id(op) in inside or not hasattr(op, 'config'): continue
            for cfg, _ in op.config_with_variable:
                root, hops = cfg, 0
                while getattr(root, 'dominated_by', root) is not root and hops < 64: root, hops = root.dominated_by, hops + 1
                if root is not cfg and id(root) in owned:
                    stack.extend(op.outputs)
                    break
        while stack:
            v = stack.pop()
            if v.name in dead: continue
            dead.add(v.name)
            for d in v.dest_ops: stack.extend(d.outputs)
        for cache in self.values:
            for n in dead: cache.pop(n, None)
        self.prune()

    def prune(self) -> None:
        """Drop the cached tensors that cannot be read again: a tensor is an operand of a future ``forward_cached`` only while
        one of its consumers still has an output missing from the cache.  Blocks are visited in execution order and a trained
        block only invalidates what lies downstream of it, so a consumer whose outputs are all present is never run again.
        What stays is the frontier the next blocks start from instead of every activation of the prefix -- per batch, the
        activations alive at one cut of the graph rather than all of them (ResNet-50 at batch 32: ~0.2 GB instead of 1.4 GB)."""
        variables = self.graph.variables
        for cache in self.values:
            for name in list(cache):
                v = variables.get(name)
                if v is None: continue
                if all(o.name in cache for d in v.dest_ops for o in d.outputs): del cache[name]

    def resident_bytes(self) -> int:
        seen, total = set(), 0
        for cache in self.values:
            for t in cache.values():
                key = t.untyped_storage().data_ptr()
                if key not in seen: seen.add(key); total += t.untyped_storage().nbytes()
        return total


@ torch.no_grad()
def compute_block_loss(block: TrainableBlock, qt_inputs, fp_outputs, executor, loss_fn=torch_mean_square_error) -> float:
    """optim/training.py:300-335."""
    names = [v.name for v in block.ep.outputs]
    terms = {n: [] for n in names}
    for qt_input, fp_output in zip(qt_inputs, fp_outputs):
        outs = block_forward(executor, block.rps, qt_input, names)
        for n, y in zip(names, outs): terms[n].append(loss_fn(y, fp_output[n]).reshape(1))
    # ONE device-to-host copy for all batches (the reference synchronises per batch, training.py:326-330); the sums are
    # formed on the host in the reference's order, in double like its Python floats
    host = torch.cat([t for n in names for t in terms[n]]).tolist()
    total, at = 0.0, 0
    for n in names:
        acc = 0.0
        for v in host[at: at + len(terms[n])]: acc += v
        at += len(terms[n])
        total += acc / len(qt_inputs)
    return total

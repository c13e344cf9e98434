"""Invertible-graph runtime, API-compatible with the reference's ``FrEIA.framework``
(reference: FrEIA/framework/graph_inn.py, reversible_graph_net.py, sequence_inn.py).

Differences in HOW (not in what is computed):
  * ``GraphINN.__init__`` lowers the node DAG once into a static *plan*.  A conditional wavelet-flow step whose
    blocks are all ConditionalAffineTransform (CWFA's default) becomes: run the blocks' sub-networks on the
    conditions (independent of the data), then ONE fused chain kernel for
    Haar1D / Split / permutations / affine couplings / log-det in either direction.
  * graphs that do not match fall back to a node-by-node walk with the same semantics as the reference's interpreter
    (every node still runs a HIP kernel).
"""
import warnings
from collections import deque
from typing import Iterable, List, Optional, Tuple, Union

import torch
import torch.nn as nn
from torch import Tensor

from ... import autograd as AG
from ... import ops
from ..modules.base import InvertibleModule

__all__ = ["SequenceINN", "ReversibleSequential", "GraphINN", "ReversibleGraphNet", "Node", "InputNode",
           "ConditionNode", "OutputNode", "topological_order"]


class Node:
    """One transformation in the graph.  ``inputs`` may be a Node (its output 0), a ``(node, idx)`` pair or a list of
    pairs; the module is instantiated here so that shapes propagate at build time.  graph_inn.py:13-116."""

    def __init__(self, inputs, module_type, module_args: dict, conditions=None, name=None):
        self.name = name if name else hex(id(self))[-6:]
        self.inputs = self.parse_inputs(inputs)
        if conditions is None:
            conditions = []
        self.conditions = conditions if isinstance(conditions, (list, tuple)) else [conditions, ]
        self.outputs: List[Optional[Tuple["Node", int]]] = []
        self.module_type = module_type
        self.module_args = module_args
        self.input_dims = [n.output_dims[i] for n, i in self.inputs]
        self.condition_dims = [cn.output_dims[0] for cn in self.conditions]
        self.module, self.output_dims = self.build_module(self.condition_dims, self.input_dims)
        for in_idx, (in_node, out_idx) in enumerate(self.inputs):
            in_node.outputs[out_idx] = (self, in_idx)
        for i in range(len(self.output_dims)):
            self.__dict__[f"out{i}"] = self, i
            self.outputs.append(None)

    def build_module(self, condition_shapes, input_shapes):
        if len(self.conditions) > 0:
            module = self.module_type(input_shapes, dims_c=condition_shapes, **self.module_args)
        else:
            module = self.module_type(input_shapes, **self.module_args)
        return module, module.output_dims(input_shapes)

    def parse_inputs(self, inputs):
        if isinstance(inputs, (list, tuple)):
            if len(inputs) == 0:
                return inputs
            if isinstance(inputs[0], (list, tuple)):
                return inputs
            if len(inputs) == 2:
                return [inputs, ]
            raise RuntimeError(f"Cannot parse inputs provided to node '{self.name}'.")
        if not isinstance(inputs, Node):
            raise ValueError(f"Received object of invalid type ({type(inputs)}) as input for node '{self.name}'.")
        return [(inputs, 0), ]

    def __str__(self):
        hint = self.module_type.__name__ if self.module_type is not None else ""
        return f"{self.__class__.__name__} {self.name!r}: {self.input_dims} -> {hint} -> {self.output_dims}"

    def __repr__(self):
        return f"{self.__class__.__name__} {self.name!r}"


class _TerminalNode(Node):
    def _no_conditions(self, condition_shapes):
        if len(condition_shapes) > 0:
            raise ValueError(f"{self.__class__.__name__} does not accept conditions")


class InputNode(_TerminalNode):
    """Graph input (output when run in reverse).  graph_inn.py:118-134."""

    def __init__(self, *dims: int, name=None):
        self.dims = dims
        super().__init__([], None, {}, name=name)

    def build_module(self, condition_shapes, input_shapes):
        self._no_conditions(condition_shapes)
        assert len(input_shapes) == 0, "Forbidden by constructor"
        return None, [self.dims]


class ConditionNode(_TerminalNode):
    """Conditional input to the sub-networks of coupling layers.  graph_inn.py:137-154."""

    def __init__(self, *dims: int, name=None):
        self.dims = dims
        super().__init__([], None, {}, name=name)
        self.outputs = []

    def build_module(self, condition_shapes, input_shapes):
        self._no_conditions(condition_shapes)
        assert len(input_shapes) == 0, "Forbidden by constructor"
        return None, [self.dims]


class OutputNode(_TerminalNode):
    """Graph output (input when run in reverse).  graph_inn.py:157-174."""

    def __init__(self, in_node, name=None):
        super().__init__(in_node, None, {}, name=name)

    def build_module(self, condition_shapes, input_shapes):
        self._no_conditions(condition_shapes)
        if len(input_shapes) != 1:
            raise ValueError(f"Output node received {len(input_shapes)} inputs,but only single input is allowed.")
        return None, []


def topological_order(all_nodes: List[Node], in_nodes: List[InputNode], out_nodes: List[OutputNode]) -> List[Node]:
    """Kahn's algorithm from the outputs backwards (condition nodes are not edges).  graph_inn.py:429-473.
    Edge containers are insertion-ordered, so the order (hence module_list / state_dict indices) is deterministic."""
    pred = {n: list(dict.fromkeys(a for a, _ in n.inputs)) for n in all_nodes + out_nodes}
    succ = {}
    for n, ps in pred.items():
        for a in ps:
            succ.setdefault(a, [])
            if n not in succ[a]:
                succ[a].append(n)
    order, ready = [], deque(out_nodes)
    while ready:
        n = ready.popleft()
        order.append(n)
        for a in list(pred[n]):
            pred[n].remove(a)
            succ[a].remove(n)
            if not succ[a]:
                ready.append(a)
    for n in in_nodes:
        if n not in order:
            raise ValueError(f"Error in graph: {n} is not connected to any output.")
    if any(len(v) for v in succ.values()):
        raise ValueError("Graph is cyclic.")
    return order[::-1]


class GraphINN(InvertibleModule):
    """The invertible network: ``forward(x_or_z, c=None, rev=False, jac=True) -> (out | tuple, logdet[B])``.
    graph_inn.py:177-326.  Attributes read by callers are kept: ``dims_c``, ``global_out_shapes``, ``module_list``,
    ``in_nodes`` / ``out_nodes`` / ``condition_nodes`` / ``node_list``."""

    def __init__(self, node_list, force_tuple_output=False, verbose=False):
        in_nodes = [n for n in node_list if isinstance(n, InputNode)]
        out_nodes = [n for n in node_list if isinstance(n, OutputNode)]
        condition_nodes = [n for n in node_list if isinstance(n, ConditionNode)]
        for node in node_list:
            for in_node, idx in node.inputs:
                if in_node not in node_list:
                    raise ValueError(f"{node} gets input from {in_node}, but the latter is not in the node_list "
                                     f"passed to GraphINN.")
            for out_node, idx in node.outputs:
                if out_node not in node_list:
                    raise ValueError(f"{out_node} gets input from {node}, but the it's not in the node_list passed "
                                     f"to GraphINN.")
        node_list = topological_order(node_list, in_nodes, out_nodes)
        global_in_shapes = [n.output_dims[0] for n in in_nodes]
        global_out_shapes = [n.input_dims[0] for n in out_nodes]
        global_cond_shapes = [n.output_dims[0] for n in condition_nodes]
        super().__init__(global_in_shapes, global_cond_shapes)
        self.node_list = node_list
        self.in_nodes, self.condition_nodes, self.out_nodes = in_nodes, condition_nodes, out_nodes
        self.global_out_shapes = global_out_shapes
        self.force_tuple_output = force_tuple_output
        self.module_list = nn.ModuleList([n.module for n in node_list if n.module is not None])
        self._plan = _lower_cat_step(self)
        if verbose:
            print(self)

    def output_dims(self, input_dims):
        if len(self.global_out_shapes) == 1 and not self.force_tuple_output:
            raise ValueError("You can only call output_dims on a GraphINN with more than one output or when setting "
                             "force_tuple_output=True.")
        return self.global_out_shapes

    # ------------------------------------------------------------------ execution
    def forward(self, x_or_z: Union[Tensor, Iterable[Tensor]], c: Iterable[Tensor] = None, rev: bool = False,
                jac: bool = True, intermediate_outputs: bool = False, x: None = None, sumsq: Tensor = None):
        """``sumsq`` (extension, optional float64[1] device tensor): the fused forward plan adds ||Z||^2 into it, which
        is the prior term of the NLL (CWFA.py:970) -- saves re-reading Z."""
        if x is not None:
            x_or_z = x
            warnings.warn("You called GraphINN(x=...). x is now called x_or_z, please pass input as positional argument.")
        if torch.is_tensor(x_or_z):
            x_or_z = x_or_z,
        if torch.is_tensor(c):
            c = c,
        x_or_z = tuple(x_or_z)
        start_nodes = self.out_nodes if rev else self.in_nodes
        if len(x_or_z) != len(start_nodes):
            raise ValueError(f"Got {len(x_or_z)} inputs, but expected {len(start_nodes)}.")
        c = [] if c is None else list(c)
        if len(c) != len(self.condition_nodes):
            raise ValueError(f"Got {len(c)} conditions, but expected {len(self.condition_nodes)}.")
        if AG.tracking(list(x_or_z), c, self):
            # torch would record a graph (training): the all-CAT step keeps its fused chain as ONE autograd node per direction;
            # every other graph goes node by node, each module an autograd node of its own (cwfa_amd.autograd)
            plan = self._plan
            if (type(plan) is _CatStepPlan and not intermediate_outputs and not any(k == "act" for k, _ in plan.chain)
                    and all(hasattr(n.module.subnet, "affine_parts") for k, n in plan.chain if k == "cat")):
                res = plan.run_tracked(x_or_z, c, rev)
                if not jac:
                    res = (res[0], None)                # as the inference plan: no log-det asked for, none returned
                if sumsq is not None and not rev:      # the ||Z||^2 extension: one extra pass here (the inference plan fuses it)
                    zt = res[0][plan.flow_out_idx].detach()
                    sumsq += ops.sample_stats(zt.reshape(1, -1, 1, 1))[1]
                return res
        elif self._plan is not None and not intermediate_outputs and not self._plan.needs_walk():
            return self._plan.run(x_or_z, c, rev, sumsq, jac)
        if any(t is None for t in x_or_z):
            # None = an all-zero latent (the default T = 0 inverse pass, CWFA.py:54-55).  The fused plans never read it; the node
            # walk (any other graph, or a plan that first has to initialise an ActNorm from this batch) needs the tensor
            ref = next((t for t in list(x_or_z) + c if t is not None), None)
            if not rev or ref is None:
                raise ValueError("None (an all-zero latent) is only understood for rev=True with at least one tensor input")
            x_or_z = tuple(t if t is not None else torch.zeros((ref.shape[0],) + tuple(shp), dtype=ref.dtype, device=ref.device)
                           for t, shp in zip(x_or_z, self.global_out_shapes))
        res = self._walk(x_or_z, c, rev, jac, intermediate_outputs)
        if sumsq is not None and not rev and not intermediate_outputs:       # the ||Z||^2 extension outside the fused plans: one extra pass
            outs = res[0]
            zt = (outs[0] if isinstance(outs, (tuple, list)) else outs).detach()
            sumsq += ops.sample_stats(zt.contiguous().reshape(1, -1, 1, 1))[1]
        return res

    def _walk(self, x_or_z, c, rev, jac, intermediate_outputs):
        """Node-by-node execution (any graph).  Same data flow as graph_inn.py:259-326."""
        first = x_or_z[0]
        jacobian = torch.zeros(first.shape[0], dtype=first.dtype, device=first.device)
        vals = {}
        jac_by_node = {} if jac else None
        for t, n in zip(x_or_z, self.out_nodes if rev else self.in_nodes):
            vals[n, 0] = t
        for t, n in zip(c, self.condition_nodes):
            vals[n, 0] = t
        special = set(self.in_nodes + self.out_nodes + self.condition_nodes)
        for node in (reversed(self.node_list) if rev else self.node_list):
            if node in special:
                continue
            mod_in = tuple(vals[p, ch] for p, ch in (node.outputs if rev else node.inputs))
            mod_c = tuple(vals[cn, 0] for cn in node.conditions)
            if len(node.conditions) > 0:
                mod_out = node.module(mod_in, c=mod_c, rev=rev, jac=jac)
            else:
                mod_out = node.module(mod_in, rev=rev, jac=jac)
            out, mod_jac = self._check_output(node, mod_out, jac, rev)
            for i, v in enumerate(out):
                vals[node, i] = v
            if jac:
                jacobian = jacobian + mod_jac
                jac_by_node[node] = mod_jac
        ends = self.in_nodes if rev else self.out_nodes
        for n in ends:
            vals[n, 0] = vals[(n.outputs if rev else n.inputs)[0]]
        if intermediate_outputs:
            return vals, jac_by_node
        outs = [vals[n, 0] for n in ends]
        if len(outs) == 1 and not self.force_tuple_output:
            return outs[0], jacobian
        return tuple(outs), jacobian

    def _check_output(self, node, mod_out, jac, rev):
        if torch.is_tensor(mod_out):
            raise ValueError(f"The node {node}'s module returned a tensor only. This is deprecated without fallback. "
                             f"Please follow the signature of InvertibleOperator#forward in your module if you want to "
                             f"use it in a GraphINN.")
        if len(mod_out) != 2:
            raise ValueError(f"The node {node}'s module returned a tuple of length {len(mod_out)}, but should return a "
                             f"tuple `z_or_x, jac`.")
        out, mod_jac = mod_out
        if torch.is_tensor(out):
            raise ValueError(f"The node {node}'s module returns a tensor. This is deprecated.")
        want = len(node.inputs if rev else node.outputs)
        if len(out) != want:
            raise ValueError(f"The node {node}'s module returned {len(out)} output variables, but should return {want}.")
        if not torch.is_tensor(mod_jac):
            if isinstance(mod_jac, (float, int)):
                mod_jac = torch.zeros(out[0].shape[0], dtype=out[0].dtype, device=out[0].device) + mod_jac
            elif jac:
                raise ValueError(f"The node {node}'s module returned a non-tensor as Jacobian: {mod_jac}")
            elif mod_jac is not None:
                raise ValueError(f"The node {node}'s module returned neither None nor a Jacobian: {mod_jac}")
        return out, mod_jac

    def log_jacobian_numerical(self, x, c=None, rev=False, h=1e-04):
        """Finite-difference log-det (tiny graphs only).  graph_inn.py:369-407."""
        single = not isinstance(x, (list, tuple))
        xs = [x] if single else list(x)
        B = xs[0].shape[0]
        sizes = [int(v[0].numel()) for v in xs]
        n = sum(sizes)
        flat = torch.cat([v.reshape(B, -1) for v in xs], dim=1)

        def run(fl):
            parts = [p.reshape(v.shape) for p, v in zip(torch.split(fl, sizes, dim=1), xs)]
            y, _ = self.forward(parts[0] if single else parts, c=c, rev=rev, jac=False)
            ys = [y] if torch.is_tensor(y) else list(y)
            return torch.cat([v.reshape(B, -1) for v in ys], dim=1)

        J = torch.zeros(B, n, n, device=flat.device)
        for i in range(n):
            off = torch.zeros_like(flat)
            off[:, i] = h
            J[:, :, i] = (run(flat + off) - run(flat - off)) / (2 * h)
        return torch.stack([torch.slogdet(J[i].cpu())[1] for i in range(B)]).to(flat.device)

    def get_node_by_name(self, name) -> Optional[Node]:
        for node in self.node_list:
            if node.name == name:
                return node
        return None

    def get_module_by_name(self, name) -> Optional[nn.Module]:
        node = self.get_node_by_name(name)
        return getattr(node, "module", None)


# ---------------------------------------------------------------------------------------------------------------------
# plan lowering: conditional wavelet-flow step with ConditionalAffineTransform blocks  (networks.py:305-366)
# ---------------------------------------------------------------------------------------------------------------------
class _CatStepPlan:
    """Haar1D -> Split -> [permutation | ConditionalAffineTransform]* on the detail half; outputs (flow, low).

    ``chain`` lists the flow-branch nodes in forward order as ('perm', module) / ('cat', node).  Executing a direction
    = evaluating every block's sub-network on its conditions (they do not depend on the data) and launching ONE fused
    kernel (ops.chain_fwd / ops.chain_inv)."""

    def __init__(self, graph, chain, flow_out_idx, low_out_idx):
        self.graph, self.chain = graph, chain
        self.flow_out_idx, self.low_out_idx = flow_out_idx, low_out_idx
        self._tables = {}

    def needs_walk(self):
        """An ActNorm that still has to initialise itself from its first batch (invertible_resnet.py:45-66) needs its input
        tensor: that call goes node by node."""
        return any(k == "act" and obj.init_on_next_batch for k, obj in self.chain)

    def _composed(self, rev, perms, final_perm, shape, device):
        """Per-axis composition of the direction's gathers (ops.chain_tables), cached while the permutation parameters
        are unchanged."""
        tabs = [t for t, _ in perms if t is not None] + ([final_perm] if final_perm is not None else [])
        key = (tuple(shape), str(device), tuple((t.data_ptr(), t._version) for t in tabs))
        hit = self._tables.get(rev)
        if hit is None or hit[0] != key:
            hit = self._tables[rev] = (key, ops.chain_tables(perms, final_perm, *shape, device))
        return hit[1]

    def _stages(self, cond_of, rev, coefficients=None):
        """Stage list in execution order; every stage = (input gather, affine).  ``coefficients(module, conditions)``
        -> (s_raw, t, t_neg_div_sqrt2) replaces the block's own sub-network call (the training path keeps a tape)."""
        stages, pending = [], None
        seq = list(reversed(self.chain)) if rev else self.chain
        for kind, obj in seq:
            if kind == "perm":
                if pending is not None:                                 # two permutations in a row: identity affine
                    stages.append(ops.stage(None, None, perm=pending[0], axis=pending[1]))
                pending = (obj.table(rev), obj.axis)
            elif kind == "act":                                         # ActNorm: per-channel s, t (invertible_resnet.py:68-81)
                perm, axis = pending if pending is not None else (None, 1)
                stages.append(obj.chain_stage(perm=perm, axis=axis))
                pending = None
            else:
                c = [cond_of[cn] for cn in obj.conditions]
                perm, axis = pending if pending is not None else (None, 1)
                if coefficients is None:
                    stages.append(obj.module.stage(c, perm=perm, axis=axis))
                else:
                    s_raw, t, tneg = coefficients(obj.module, c)
                    stages.append(obj.module._stage(s_raw, t, t_neg_div_sqrt2=tneg, perm=perm, axis=axis))
                pending = None
        return stages, pending

    @staticmethod
    def _perms_of(stages):
        """(table, axis) of every stage: the int64 tensor a stage keeps alive for its gather, or None."""
        out = []
        for st, keep in stages:
            table = next((t for t in keep if t.dtype == torch.int64), None) if st.perm else None
            out.append((table, int(st.perm_axis)))
        return out

    def run_tracked(self, x_or_z, c, rev):
        """``run`` with torch recording a graph: the sub-networks are autograd nodes (cwfa_amd.autograd.subnet), the chain launch
        is one node whose backward (ops.chain_bwd / chain_inv_bwd) recomputes every stage input by inverting the stage."""
        g = self.graph
        cond_of = dict(zip(g.condition_nodes, c))
        cache = {}

        def coefficients(module, parts):
            hit = cache.get(id(module))
            if hit is None:
                hit = cache[id(module)] = module.coefficients(parts)
            return hit

        def coef_of(stages):            # per coefficient stage, in stage order, the tensors its struct was built from
            out = []
            for kind, obj in (list(reversed(self.chain)) if stages is rstages else self.chain):
                if kind == "cat":
                    s_raw, t, _ = cache[id(obj.module)]
                    out.append((s_raw, t))
            return out

        first = next(t for t in x_or_z if t is not None)
        fstages, pending = self._stages(cond_of, False, coefficients=coefficients)
        final_perm = None
        if pending is not None:
            if pending[1] == 1:
                final_perm = pending[0]
            else:
                fstages.append(ops.stage(None, None, perm=pending[0], axis=pending[1]))
        rstages = None
        if not rev:
            shp = (first.shape[1] // 2, first.shape[2], first.shape[3])
            tabs = self._composed(False, self._perms_of(fstages), final_perm, shp, first.device)
            z, low, ld = AG.chain_fwd(first, fstages, final_perm, tabs, coef_of(fstages))
            outs = [None, None]
            outs[self.flow_out_idx], outs[self.low_out_idx] = z, low
            return tuple(outs), ld
        rstages, rpending = self._stages(cond_of, True, coefficients=coefficients)
        if rpending is not None:
            rstages.append(ops.stage(None, None, perm=rpending[0], axis=rpending[1]))
        z, low = x_or_z[self.flow_out_idx], x_or_z[self.low_out_idx]
        rtabs = self._composed(True, self._perms_of(rstages), None, tuple(low.shape[1:]), low.device)
        xhat, ld = AG.chain_inv(z, low, rstages, rtabs, fstages, final_perm, coef_of(fstages))
        return (xhat if not g.force_tuple_output else (xhat,)), ld

    def run(self, x_or_z, c, rev, sumsq=None, jac=True):
        """``jac=False`` skips the log-det reduction (the reconstruction loop discards it, CWFA.py:912) and returns
        ``None`` in its place."""
        g = self.graph
        cond_of = dict(zip(g.condition_nodes, c))
        from ...networks import merged_first_maps
        jobs = [(n.module.subnet, [cond_of[cn] for cn in n.conditions]) for k, n in self.chain if k == "cat"]
        # the blocks' sub-networks share their condition: its ones channel is built once, their first 1x1 maps in one launch
        with ops.ones_channel_scope(), ops.first_map_scope(merged_first_maps(jobs)):
            stages, pending = self._stages(cond_of, rev)
        first = next(t for t in x_or_z if t is not None)
        acc = torch.zeros(first.shape[0], dtype=torch.float64, device=first.device) if jac else None
        if rev:
            z, low = x_or_z[self.flow_out_idx], x_or_z[self.low_out_idx]
            if pending is not None:
                stages.append(ops.stage(None, None, perm=pending[0], axis=pending[1]))
            tabs = self._composed(True, self._perms_of(stages), None, tuple(low.shape[1:]), low.device)
            out = ops.chain_inv(z, low, stages, logdet=acc, tables=tabs)
            res = out if not g.force_tuple_output else (out,)
            return res, (acc.to(torch.float32) if jac else None)
        final_perm = None
        if pending is not None:
            if pending[1] == 1:
                final_perm = pending[0]
            else:
                stages.append(ops.stage(None, None, perm=pending[0], axis=pending[1]))
        shp = (first.shape[1] // 2, first.shape[2], first.shape[3])
        tabs = self._composed(False, self._perms_of(stages), final_perm, shp, first.device)
        z, low = ops.chain_fwd(first, stages, final_perm, logdet=acc, sumsq=sumsq, tables=tabs)
        outs = [None, None]
        outs[self.flow_out_idx], outs[self.low_out_idx] = z, low
        return tuple(outs), (acc.to(torch.float32) if jac else None)


class _MixedStepPlan(_CatStepPlan):
    """The same step graph with data-dependent coupling blocks (GLOW / RNVP / GIN / NICE / one-sided / AllInOne,
    networks.py:305-366 with ``block_type`` != 'CAT') between the permutations.

    ``chain`` additionally holds ('blk', node) entries.  The run of permutations / ConditionalAffineTransforms next to the
    Haar transform -- forward: [first CAT, permutation]; the reference builds every step with this prefix, networks.py:
    321-333 -- goes through ONE fused chain launch together with Haar1D and Split (ops.chain_fwd / chain_inv, exactly as in
    the CAT plan); the blocks after it run as their own modules: each sub-network applies its coupling in the epilogue of
    its last convolution (s, t never reach memory, halves written in place into the block's output), and the permutation
    in front of a block is one gather launch."""

    def __init__(self, graph, chain, flow_out_idx, low_out_idx):
        super().__init__(graph, chain, flow_out_idx, low_out_idx)
        first_blk = next(i for i, (k, _) in enumerate(chain) if k == "blk")
        self.fused, self.rest = chain[:first_blk], chain[first_blk:]
        self._sub = _CatStepPlan(graph, self.fused, flow_out_idx, low_out_idx)      # stage lists of the fused prefix

    def _walk_rest(self, v, cond_of, rev, acc):
        """The block part of the chain, module by module (forward order, or reversed)."""
        for kind, obj in (reversed(self.rest) if rev else self.rest):
            if kind == "perm":
                (v,), _ = obj((v,), rev=rev)
                continue
            if kind == "act":
                (v,), j = obj((v,), rev=rev)
                if acc is not None:
                    acc += j.to(torch.float64)
                continue
            c = tuple(cond_of[cn] for cn in obj.conditions)
            (v,), j = obj.module((v,), c=c, rev=rev, jac=acc is not None) if len(c) else obj.module((v,), rev=rev, jac=acc is not None)
            if acc is not None and torch.is_tensor(j):
                acc += j.to(torch.float64)
            elif acc is not None and j:
                acc += float(j)
        return v

    def run(self, x_or_z, c, rev, sumsq=None, jac=True):
        with ops.ones_channel_scope():       # (condition | 1) of the composed first layers: built once per condition tensor and step
            return self._run(x_or_z, c, rev, sumsq, jac)

    def _run(self, x_or_z, c, rev, sumsq, jac):
        g = self.graph
        cond_of = dict(zip(g.condition_nodes, c))
        first = next(t for t in x_or_z if t is not None)
        acc = torch.zeros(first.shape[0], dtype=torch.float64, device=first.device) if jac else None
        stages, pending = self._sub._stages(cond_of, rev)
        if rev:
            z, low = x_or_z[self.flow_out_idx], x_or_z[self.low_out_idx]
            if z is None:
                z = torch.zeros_like(low)
            v = self._walk_rest(z, cond_of, True, acc)
            if pending is not None:
                stages.append(ops.stage(None, None, perm=pending[0], axis=pending[1]))
            tabs = self._composed(True, self._perms_of(stages), None, tuple(low.shape[1:]), low.device)
            out = ops.chain_inv(v, low, stages, logdet=acc, tables=tabs)
            res = out if not g.force_tuple_output else (out,)
            return res, (acc.to(torch.float32) if jac else None)
        final_perm = None
        if pending is not None:
            if pending[1] == 1:
                final_perm = pending[0]
            else:
                stages.append(ops.stage(None, None, perm=pending[0], axis=pending[1]))
        shp = (first.shape[1] // 2, first.shape[2], first.shape[3])
        tabs = self._composed(False, self._perms_of(stages), final_perm, shp, first.device)
        v, low = ops.chain_fwd(first, stages, final_perm, logdet=acc, tables=tabs)
        z = self._walk_rest(v, cond_of, False, acc)
        if sumsq is not None:
            ops.affine(z, ops.stage(None, None), False, sumsq=sumsq, out=z)      # identity stage: one pass that adds ||Z||^2
        outs = [None, None]
        outs[self.flow_out_idx], outs[self.low_out_idx] = z, low
        return tuple(outs), (acc.to(torch.float32) if jac else None)


def _lower_cat_step(g: "GraphINN"):
    """Recognise the CWFA step graph; return a plan or None (-> generic walk)."""
    from ..modules.coupling import ConditionalAffineTransform, _TwoSided, AffineCouplingOneSided, AllInOneBlock
    if len(g.in_nodes) != 1 or len(g.out_nodes) != 2:
        return None
    inner = [n for n in g.node_list if n.module is not None]
    if len(inner) < 3:
        return None
    haar, split = inner[0], inner[1]
    if type(haar.module).__name__ != "HaarTransform1D" or type(split.module).__name__ != "Split":
        return None
    if getattr(haar.module, "jac_fwd", 0.0) != 0.0 or getattr(haar.module, "jac_rev", 0.0) != 0.0:
        return None                 # rebalance != 1: the Haar node contributes ndims*log(rebalance) to the log-det
        #                             (INN_utils.py:133-157) which the fused chain does not carry -> generic node walk
    if haar.inputs[0][0] is not g.in_nodes[0] or split.inputs[0][0] is not haar or len(split.output_dims) != 2:
        return None
    if getattr(split.module, "dim", None) != 0 or split.output_dims[0] != split.output_dims[1]:
        return None
    # walk the flow branch from split.out1
    chain, cur, cur_idx = [], split, 1
    low_out = split.outputs[0][0] if split.outputs[0] is not None else None
    if not isinstance(low_out, OutputNode):
        return None
    while True:
        nxt = cur.outputs[cur_idx]
        if nxt is None:
            return None
        node, in_idx = nxt
        if isinstance(node, OutputNode):
            flow_out = node
            break
        if len(node.inputs) != 1 or len(node.output_dims) != 1:
            return None
        m = node.module
        if isinstance(m, ConditionalAffineTransform):
            if m.clamp_kind is None:
                return None
            chain.append(("cat", node))
        elif hasattr(m, "table") and hasattr(m, "axis"):
            chain.append(("perm", m))
        elif type(m).__name__ == "ActNorm" and hasattr(m, "chain_stage") and len(m.dims_in) == 3:
            chain.append(("act", m))
        elif isinstance(m, (_TwoSided, AffineCouplingOneSided, AllInOneBlock)):
            chain.append(("blk", node))
        else:
            return None
        cur, cur_idx = node, 0
    def n_stages(seq):          # a permutation merges into the block that follows it in execution order
        n, pending = 0, False
        for k, _ in seq:
            if k == "perm":
                n += int(pending)
                pending = True
            else:
                n, pending = n + 1, False
        return n + int(pending)

    if len(inner) != 2 + len(chain):
        return None
    if any(k == "blk" for k, _ in chain):
        head = chain[:next(i for i, (k, _) in enumerate(chain) if k == "blk")]
        if max(n_stages(head), n_stages(head[::-1])) > ops._lib.CHAIN_MAX:
            return None
        return _MixedStepPlan(g, chain, g.out_nodes.index(flow_out), g.out_nodes.index(low_out))
    if max(n_stages(chain), n_stages(chain[::-1])) > ops._lib.CHAIN_MAX or not any(k == "cat" for k, _ in chain):
        return None
    return _CatStepPlan(g, chain, g.out_nodes.index(flow_out), g.out_nodes.index(low_out))


class ReversibleGraphNet(GraphINN):
    """Deprecated alias kept for API compatibility.  reversible_graph_net.py:9-36."""

    def __init__(self, node_list, ind_in=None, ind_out=None, verbose=True, force_tuple_output=False):
        warnings.warn("ReversibleGraphNet is deprecated in favour of GraphINN. It will be removed in the next version "
                      "of FrEIA.", DeprecationWarning)
        if ind_in is not None:
            raise ValueError("ReversibleGraphNet's ind_in was removed in FrEIA v0.3.0. Please use InputNodes and switch "
                             "to GraphINN.")
        if ind_out is not None:
            raise ValueError("ReversibleGraphNet's ind_out was removed in FrEIA v0.3.0. Please use OutputNodes and "
                             "switch to GraphINN.")
        super().__init__(node_list, verbose=verbose, force_tuple_output=force_tuple_output)

    def forward(self, x_or_z, c=None, rev=False, jac=True, intermediate_outputs=False):
        warnings.warn("ReversibleGraphNet's forward() now returns a tuple (output, jacobian). It will be removed in the "
                      "next version of FrEIA.", DeprecationWarning)
        return super().forward(x_or_z, c, rev, jac, intermediate_outputs)


class SequenceINN(InvertibleModule):
    """Linear chain of modules with ``append``.  sequence_inn.py:10-101."""

    def __init__(self, *dims: int, force_tuple_output=False):
        super().__init__([dims])
        self.shapes = [tuple(dims)]
        self.conditions = []
        self.module_list = nn.ModuleList()
        self.force_tuple_output = force_tuple_output

    def append(self, module_class, cond=None, cond_shape=None, **kwargs):
        dims_in = [self.shapes[-1]]
        self.conditions.append(cond)
        if cond is not None:
            kwargs['dims_c'] = [cond_shape]
        module = module_class(dims_in, **kwargs)
        self.module_list.append(module)
        out_dims = module.output_dims(dims_in)
        assert len(out_dims) == 1, "Module has more than one output"
        self.shapes.append(out_dims[0])

    def output_dims(self, input_dims):
        if not self.force_tuple_output:
            raise ValueError("You can only call output_dims on a SequentialINN when setting force_tuple_output=True.")
        return input_dims

    def forward(self, x_or_z, c=None, rev=False, jac=True):
        order = range(len(self.module_list))
        total = 0
        if torch.is_tensor(x_or_z):
            x_or_z = (x_or_z,)
        for i in (reversed(order) if rev else order):
            if self.conditions[i] is None:
                x_or_z, j = self.module_list[i](x_or_z, jac=jac, rev=rev)
            else:
                x_or_z, j = self.module_list[i](x_or_z, c=[c[self.conditions[i]]], jac=jac, rev=rev)
            total = j + total
        return x_or_z if self.force_tuple_output else x_or_z[0], total


class ReversibleSequential(SequenceINN):
    """Deprecated alias of SequenceINN (reversible_sequential_net.py)."""

    def __init__(self, *dims: int):
        warnings.warn("ReversibleSequential is deprecated in favour of SequenceINN.", DeprecationWarning)
        super().__init__(*dims)

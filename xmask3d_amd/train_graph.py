"""HIP-graph replay of the static-shape stages of a TRAINING iteration (SURVEY.md §8d metric (ii): /root/reference/run/train.py:504-540
runs every stage eagerly; at one view per GPU the iteration is bound by the host's ~13 000 kernel launches, not by the device).

* ``GraphedRegion``: ``fn(*inputs) -> pytree of tensors`` with TRAINABLE parameters inside, replayed as one forward and one backward
  graph.  The parameters are arguments of the autograd node, so their gradients come back through the ordinary AccumulateGrad nodes
  of the real parameters: DistributedDataParallel's bucket hooks, gradient accumulation and the optimizer see nothing unusual.
* ``GraphedNoGrad``: a frozen, gradient-free stage (the VAE encoder / decoder taps) as one graph.

Same discipline as image_branch._GraphedTaps: every tensor autograd sees during warm-up and capture is created on the private capture
stream, the pool and the graphs live as long as the object, and what is handed out are copies (the static tensors are overwritten by
the next replay)."""
from __future__ import annotations

import torch
from torch.utils import _pytree as pytree


def _flat_tensors(tree):
    leaves, spec = pytree.tree_flatten(tree)
    idx = [i for i, l in enumerate(leaves) if torch.is_tensor(l)]
    return leaves, spec, idx


class GraphedNoGrad:
    def __init__(self, fn, inputs, warmup=2):
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream), torch.no_grad():
            self.sin = [x.detach().clone() for x in inputs]
            for _ in range(warmup):  # kernel selection and allocator warm-up outside the capture
                fn(*self.sin)
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, stream=self.stream):
            self.leaves, self.spec, self.tidx = _flat_tensors(fn(*self.sin))
        torch.cuda.synchronize()

    @torch.no_grad()
    def __call__(self, *inputs):
        for dst, src in zip(self.sin, inputs):
            dst.copy_(src)
        self.graph.replay()
        leaves = list(self.leaves)
        for i in self.tidx:
            leaves[i] = leaves[i].clone()
        return pytree.tree_unflatten(leaves, self.spec)


class GraphedRegion:
    def __init__(self, fn, inputs, params, warmup=3):
        self.params = tuple(p for p in params if p.requires_grad)
        # The parameters' AccumulateGrad nodes come into being inside the warm-up, on the private stream, and stay alive with the captured
        # autograd graph.  That is on purpose: made on the caller's (legacy) stream instead, the engine orders that stream against the
        # capturing one INSIDE the backward capture, and the replay of such a graph ends in a segmentation fault in hipGraphLaunch.  The
        # price is that the eager backward hands the parameter gradients across streams (the engine synchronises them; torch warns once).
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.sin = [x.detach().clone().requires_grad_(x.requires_grad) for x in inputs]
            self.in_grad = [i for i, x in enumerate(self.sin) if x.requires_grad]
            self.wrt = [self.sin[i] for i in self.in_grad] + list(self.params)
            for _ in range(warmup):
                leaves, _, tidx = _flat_tensors(fn(*self.sin))
                diff = [leaves[i] for i in tidx if leaves[i].requires_grad]
                torch.autograd.grad(diff, self.wrt, [torch.ones_like(o) for o in diff], allow_unused=True)
            del leaves, diff
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        self.pool = torch.cuda.graph_pool_handle()
        self.fwd, self.bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd, pool=self.pool, stream=self.stream):
            self.leaves, self.spec, self.tidx = _flat_tensors(fn(*self.sin))
        self.outs = [self.leaves[i] for i in self.tidx]
        self.diff = [j for j, o in enumerate(self.outs) if o.requires_grad]
        with torch.cuda.stream(self.stream):
            self.gouts = [torch.zeros_like(self.outs[j]) for j in self.diff]
        self.stream.synchronize()
        with torch.cuda.graph(self.bwd, pool=self.pool, stream=self.stream):
            self.gins = torch.autograd.grad([self.outs[j] for j in self.diff], self.wrt, self.gouts, only_inputs=True, allow_unused=True)
        torch.cuda.synchronize()

    def __call__(self, *inputs):
        outs = _GraphedRegionFn.apply(self, len(inputs), *inputs, *self.params)
        leaves = list(self.leaves)
        for i, o in zip(self.tidx, outs):
            leaves[i] = o
        return pytree.tree_unflatten(leaves, self.spec)


class _GraphedRegionFn(torch.autograd.Function):
    @staticmethod
    def forward(c, g, n_in, *args):
        with torch.no_grad():
            for dst, src in zip(g.sin, args[:n_in]):
                dst.copy_(src)
        g.fwd.replay()
        c.g, c.n_in = g, n_in
        g.generation = getattr(g, "generation", 0) + 1
        c.generation = g.generation
        outs = tuple(o.detach().clone() for o in g.outs)
        c.mark_non_differentiable(*[o for j, o in enumerate(outs) if j not in set(g.diff)])
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(c, *grads):
        g = c.g
        if c.generation != g.generation:
            raise RuntimeError("graphed region: backward of a forward whose activations a later forward of the same shapes has overwritten")
        for dst, j in zip(g.gouts, g.diff):
            if grads[j] is None:
                dst.zero_()
            else:
                dst.copy_(grads[j])
        g.bwd.replay()
        gin = [None] * c.n_in
        for k, i in enumerate(g.in_grad):
            gin[i] = None if g.gins[k] is None else g.gins[k].detach().clone()
        gp = [None if x is None else x.detach().clone() for x in g.gins[len(g.in_grad):]]
        return (None, None, *gin, *gp)

"""Single-node data parallelism for the fusion trainer: one process per GPU, parameters
replicated, the flat fp32 gradient buffer all-reduced once per optimizer step -- in one piece
(``GradAllReducer``) or in two buckets, the first overlapped with the node-level backward
(``BucketedGradAllReducer``) -- over RCCL/xGMI when the process group's backend is ``nccl``;
``gloo`` on CPU in the tests.

Semantics (the reference itself is single-process): the reference's gradients are the
SUM over the samples of a minibatch, clipped once (train_multimodal.py:238-279).  A global
minibatch sharded over ranks therefore needs ``all_reduce(SUM)`` -- not the mean -- BEFORE the
clip, so that N ranks x B samples equals the reference run with ``batch_size = N*B``.  The clip
coefficient and the AdamW update are then computed redundantly (and identically) on every rank
from the reduced buffer, so parameters stay bit-identical across ranks without a broadcast.

Samples are independent (no BatchNorm, per-row LayerNorm), so there is no other exchange step.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradAllReducer:
    """Callable handed to ``FusedClipAdamW.step(allreduce=...)``."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._timing = None                     # list of per-call event tuples while timing is on (bench.py's ddp diagnostics)

    def enable_timing(self, on=True):
        """Record device events around every all-reduce from now on (``timing_summary`` reads them).  Costs a few event records
        per step; off in product runs."""
        self._timing = [] if on else None

    @staticmethod
    def _ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def timing_summary(self):
        """Mean microseconds per all-reduce over the timed calls (synchronises).  Buckets: ``a`` = the overlapped tail bucket
        (BucketedGradAllReducer), ``b`` = the rest / the single all-reduce.  ``overlap_fraction``: share of bucket a's duration that
        had elapsed when the backward kernels of the step were done (1.0 = fully hidden behind them)."""
        if not self._timing:
            return None
        torch.cuda.synchronize()
        out = {"calls": len(self._timing)}
        a = [t for t in self._timing if t[0] is not None]
        if a:
            dur = [t[0].elapsed_time(t[1]) * 1e3 for t in a]
            hid = [min(max(t[0].elapsed_time(t[2]) * 1e3, 0.0), d) / max(d, 1e-9) for t, d in zip(a, dur)]
            out["bucket_a_us"] = round(sum(dur) / len(dur), 2); out["overlap_fraction"] = round(sum(hid) / len(hid), 4)
        b = [t[3].elapsed_time(t[4]) * 1e3 for t in self._timing]
        out["bucket_b_us"] = round(sum(b) / len(b), 2)
        return out

    def __call__(self, flat_grads: torch.Tensor):
        if self.world > 1:
            t0 = self._ev() if self._timing is not None and flat_grads.is_cuda else None
            dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=self.group)
            if t0 is not None:
                self._timing.append((None, None, None, t0, self._ev()))
        return flat_grads


class BucketedGradAllReducer(GradAllReducer):
    """Two buckets instead of one, the first overlapped with the node-level backward.

    The gradients of the per-sample tail (pooled KG FFN layer, fusion layer, heads: ``engine.tail_grad_offset()`` .. end of
    the flat buffer, ~1/3 of it) are final before the four node-level backward launches start.  The native training call
    records an event at that point (``camo_forward_loss_backward(tail_event=...)``); bucket A's all-reduce is issued on a
    side stream behind that event and runs beside the backward kernels, bucket B (everything in front of the offset) is
    issued behind the whole call.  Both land before the clip: ``wait()`` orders the launch stream behind them.  The sum
    is the same as the single all-reduce's (a different partition of the same element-wise SUM)."""
    overlapped = True

    def __init__(self, group=None):
        super().__init__(group)
        self._event = None
        self._side = None

    def tail_event(self, device):
        """Raw hipEvent_t handle for the training call, or None without a GPU / with a single rank."""
        if self.world <= 1 or torch.device(device).type != "cuda":
            return None
        if self._event is None:
            with torch.cuda.device(device):
                self._event = torch.cuda.Event()
                self._event.record()                         # (the handle exists once the event has been recorded)
                self._side = torch.cuda.Stream(device)
        h = self._event.cuda_event
        return int(getattr(h, "value", h) or 0) or None

    def __call__(self, flat_grads: torch.Tensor, split=None):
        if self.world <= 1:
            return flat_grads
        if not split or split >= flat_grads.numel():
            return super().__call__(flat_grads)
        head, tail = flat_grads[:split], flat_grads[split:]
        if flat_grads.is_cuda and self._event is not None:
            timed = self._timing is not None
            self._side.wait_event(self._event)
            with torch.cuda.stream(self._side):
                a0 = self._ev() if timed else None
                wa = dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if timed:
                    wa.wait()                                # (orders the side stream behind the collective: its end event)
                    a1 = self._ev()
            bwd_done = self._ev() if timed else None         # launch stream: every backward kernel of the step is in front of this
            wb = dist.all_reduce(head, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            wa.wait(); wb.wait()
            if timed:
                self._timing.append((a0, a1, bwd_done, bwd_done, self._ev()))
        else:
            dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(head, op=dist.ReduceOp.SUM, group=self.group)
        return flat_grads


class OneShotGradAllReducer(GradAllReducer):
    """The all-reduce as ONE exchange step per direction instead of a ring: reduce-scatter by ``all_to_all`` (every rank sends
    shard j of its gradients straight to rank j -- on an xGMI node each of the seven peers has its own link, so the seven
    724 KB shards travel concurrently), a local sum of the ``world`` received shards in rank order (deterministic, identical
    on every rank after the gather), then ``all_gather``.  SURVEY.md 8e proposes this shape for the 5.8 MB buffer, where a
    ring's 2 (N - 1) dependent steps are latency-bound.  UNMEASURED (no multi-GPU node was available): kept as an option
    (``bench.py --allreduce oneshot``), not the default; the CPU gloo test holds it to the plain all-reduce."""

    def __init__(self, group=None):
        super().__init__(group)
        self._send = self._recv = None

    def __call__(self, flat_grads: torch.Tensor, split=None):
        w = self.world
        if w <= 1:
            return flat_grads
        n = flat_grads.numel()
        shard = ((n + w - 1) // w + 3) // 4 * 4
        if self._send is None or self._send.numel() != w * shard or self._send.device != flat_grads.device:
            self._send = torch.zeros(w * shard, dtype=flat_grads.dtype, device=flat_grads.device)
            self._recv = torch.empty_like(self._send)
        self._send[:n].copy_(flat_grads)
        dist.all_to_all_single(self._recv, self._send, group=self.group)          # recv[j] = rank j's copy of MY shard
        mine = self._recv.view(w, shard).sum(dim=0)                                # ranks 0..w-1 in order: every rank sums alike
        dist.all_gather_into_tensor(self._send, mine, group=self.group)
        flat_grads.copy_(self._send[:n])
        return flat_grads


def broadcast_parameters(flat_params, src=0, group=None):
    """Make every rank start from rank ``src``'s parameters (one flat broadcast).  ``flat_params``: the flat buffer, or the
    ``FusionEngine`` that owns it -- pass the engine whenever weight shadows may exist: a collective that fills the buffer in place
    is not guaranteed to bump torch's version counter on the receiving ranks, so the shadows are invalidated explicitly."""
    eng = None
    if not isinstance(flat_params, torch.Tensor):
        eng, flat_params = flat_params, flat_params.flat_params
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)
    if eng is not None:
        eng.invalidate_shadows()
    return flat_params


def shard_by_rows(nrs, world, rank):
    """Split a global minibatch over ranks balancing the total number of RG rows (Nr varies
    303..530 in the real data) rather than the sample count: longest-first greedy bin packing.
    Deterministic, identical on every rank.  Returns the sorted sample indices of ``rank``."""
    order = sorted(range(len(nrs)), key=lambda i: (-int(nrs[i]), i))
    load = [0] * world
    bins = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        bins[r].append(i)
        load[r] += int(nrs[i])
    return sorted(bins[rank])


def sharded_weighted_sampler(weights, num_samples, epoch, world, rank, seed=0):
    """The reference draws ``num_samples`` indices with replacement from ``weights``
    (WeightedRandomSampler, train_multimodal.py:385-387).  Every rank draws the SAME sequence from
    a common seed and keeps its strided share, so the union over ranks is one reference epoch."""
    g = torch.Generator()
    g.manual_seed(seed * 1000003 + epoch)
    # every rank must cut the SAME number of minibatches (each optimizer step is a collective): the draw is rounded up to a
    # multiple of the world size, so the strided shares have equal length
    total = -(-int(num_samples) // world) * world
    idx = torch.multinomial(torch.as_tensor(weights, dtype=torch.double), total, replacement=True, generator=g)
    return idx[rank::world].tolist()


def reduce_metrics(values: torch.Tensor, group=None):
    """SUM a small tensor of running metrics (loss sum, TP/FP/FN/TN counts) over ranks -- replaces
    the reference's per-sample ``.item()`` bookkeeping (train_multimodal.py:269-275)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM, group=group)
    return values

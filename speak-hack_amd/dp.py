"""Data-parallel gradient exchange for one-process-per-GPU training (BASELINE config 4).

The reference wraps the model with HF ``accelerate`` (train.py:333-338,399-401), i.e. stock
``DistributedDataParallel`` over NCCL -- and in fact only runs at world size 1 (it reaches through the
wrapper for ``model.D`` / ``model.Gd``).  This is the MI355X-side replacement: the
mini-batch is sharded by rank (no data-path collective in forward), and the ONE exchange step of a
training iteration -- summing parameter gradients over ranks -- is done here:

* parameters are grouped, in reverse registration order (~ the order autograd finishes them: toRGB
  first, mapping last), into flat fp32 buckets;
* ``zero_grad()`` drops the gradients (``p.grad = None``), so backward's first -- here only -- gradient of a
  parameter is simply adopted by autograd: no zero-fill of 386 MB and no ``grad += g`` kernel per parameter
  (630 launches, 2 ms of the generator step when the gradients lived inside pre-zeroed buckets);
* a post-accumulate hook per parameter counts arrivals; when a bucket is complete its gradients are gathered into
  the flat buffer by ONE multi-tensor copy, ``param.grad`` is re-pointed at its slice of the buffer, and the
  all-reduce is launched asynchronously -- on GPU that is RCCL over xGMI on its own stream, overlapping the rest
  of backward (the big low-resolution 512x512x3x3 weight gradients, 9.4 MB each, arrive last and are the only
  exposed part).  With one rank there is nothing to exchange and nothing is copied;
* ``finish()`` waits for the outstanding collectives (stream-side on GPU) and leaves ``1/world``-scaled
  sums in place, exactly what DDP's gradient averaging leaves.

Bucket size: xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce of S bytes costs
about 2*(7/8)*S/153 GB/s; 32 MiB buckets keep each collective ~0.4 ms (far above the ~20 us launch
latency) and give the decoder's 104 MB of gradients 4 chances to overlap.  The same class runs on CPU
tensors with the ``gloo`` backend (tests/test_dp_gloo.py, world size 2).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class GradBucketReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 32 << 20,
                 process_group: Optional[dist.ProcessGroup] = None, average: bool = True):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.average = average
        plist = [p for p in params if p.requires_grad]
        if not plist:
            raise ValueError("GradBucketReducer: no trainable parameters")
        self.params = list(reversed(plist))
        dev, dtype = self.params[0].device, self.params[0].dtype
        # ---- assign parameters to buckets ----
        self.buckets: List[dict] = []
        cur, cur_bytes = [], 0
        for p in self.params:
            if p.device != dev or p.dtype != dtype:
                raise ValueError("GradBucketReducer: parameters must share device and dtype")
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                self._seal(cur, dev, dtype)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._seal(cur, dev, dtype)
        self._handles = []
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(bi))
                       for bi, b in enumerate(self.buckets) for p in b["params"]]
        self.zero_grad()

    def _seal(self, plist, dev, dtype):
        total = sum(p.numel() for p in plist)
        flat = views = None
        if self.world > 1:                     # the communication buffer and each parameter's slice of it
            flat = torch.zeros(total, device=dev, dtype=dtype)
            views, off = [], 0
            for p in plist:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
        self.buckets.append({"params": plist, "flat": flat, "views": views, "numel": total, "pending": len(plist),
                             "n": len(plist)})

    def _make_hook(self, bi):
        def hook(param):
            b = self.buckets[bi]
            b["pending"] -= 1
            if b["pending"] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        if self.world == 1:
            return
        have_v, have_g, missing = [], [], []
        for p, v in zip(b["params"], b["views"]):
            if p.grad is None:
                missing.append(v)              # no gradient this step: contributes zeros
            elif p.grad.data_ptr() != v.data_ptr():
                have_v.append(v)
                have_g.append(p.grad)
        with torch.no_grad():
            if missing:
                torch._foreach_zero_(missing)
            if have_v:
                torch._foreach_copy_(have_v, have_g)
            for p, v in zip(b["params"], b["views"]):
                p.grad = v                     # the optimizer reads the reduced values in place
            if self.average:
                b["flat"].div_(self.world)
        self._handles.append(dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    # ---- step protocol -------------------------------------------------------------------------
    def zero_grad(self):
        """Drop the gradients (``set_to_none``) and re-arm the buckets."""
        for b in self.buckets:
            for p in b["params"]:
                p.grad = None
            b["pending"] = b["n"]
        self._handles = []

    def finish(self):
        """Call after ``loss.backward()``: launches buckets whose parameters got no gradient this step
        (unused parameters contribute zeros), then waits for every collective."""
        for b in self.buckets:
            if b["pending"] > 0:           # incomplete or untouched: still takes part, in bucket order on every rank
                self._launch(b)
            b["pending"] = b["n"]
        for h in self._handles:
            h.wait()
        self._handles = []

    def _grads(self):
        return [p.grad for p in self.params if p.grad is not None]

    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm of the (already synchronised, hence rank-identical) gradients -- what
        ``clip_grad_norm_`` (train.py:207-208) needs; no further collective."""
        if self.world > 1:
            return torch.sqrt(sum((b["flat"].double() ** 2).sum() for b in self.buckets)).float()
        grads = self._grads()
        if not grads:
            return torch.zeros((), device=self.params[0].device)
        return torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads)).double()).float()

    def clip_(self, max_norm: float) -> torch.Tensor:
        total = self.grad_norm()
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        if self.world > 1:
            for b in self.buckets:
                b["flat"].mul_(coef)
        else:
            grads = self._grads()
            if grads:
                torch._foreach_mul_(grads, coef)
        return total

    def bytes_per_step(self) -> int:
        return sum(b["numel"] * p.element_size() for b in self.buckets for p in b["params"][:1])

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def shard_batch(global_batch: int, rank: int, world: int):
    """Rank r takes samples [r*B/world, (r+1)*B/world) -- SURVEY.md 8e."""
    per = global_batch // world
    return rank * per, (rank + 1) * per

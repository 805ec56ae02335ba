"""Data-parallel gradient exchange for one-process-per-GPU training (BASELINE config 4).

The reference wraps the model with HF ``accelerate`` (train.py:333-338,399-401), i.e. stock
``DistributedDataParallel`` over NCCL -- and in fact only runs at world size 1 (it reaches through the
wrapper for ``model.D`` / ``model.Gd``).  This is the MI355X-side replacement: the
mini-batch is sharded by rank (no data-path collective in forward), and the ONE exchange step of a
training iteration -- summing parameter gradients over ranks -- is done here:

* construction broadcasts every parameter from rank 0 (replicas start identical whatever the callers seeded);
* parameters are grouped into flat fp32 buckets.  The first backward runs on buckets in reverse registration
  order; the order in which the per-parameter hooks actually fired is recorded, rank 0's order is broadcast, and
  the buckets are REBUILT in that order (``rebuilt``): the first bucket then holds what autograd finishes first
  (toRGB, the 256^2 layers) and parameters that never receive a gradient (``Cm`` in a reconstruction-only step,
  SURVEY.md 3.1) sit in trailing buckets of their own instead of stalling a hot one;
* ``zero_grad()`` drops the gradients (``p.grad = None``), so backward's first -- here only -- gradient of a
  parameter is simply adopted by autograd: no zero-fill of 386 MB and no ``grad += g`` kernel per parameter;
* a post-accumulate hook per parameter counts arrivals; when a bucket is complete AND every lower-index bucket has
  been launched (collectives must pair up in the same order on every rank) its gradients are gathered into the flat
  buffer by ONE multi-tensor copy, ``param.grad`` is re-pointed at its slice of the buffer, and the all-reduce is
  launched asynchronously -- on GPU that is RCCL over xGMI on its own stream, overlapping the rest of backward
  (the big low-resolution 512x512x3x3 weight gradients, 9.4 MB each, arrive last and are the only exposed part).
  With one rank there is nothing to exchange and nothing is copied;
* ``finish()`` launches, in index order, what the hooks could not (buckets with parameters that got no gradient
  this step: they contribute zeros), waits for the outstanding collectives (stream-side on GPU) and leaves
  ``1/world``-scaled sums in place, exactly what DDP's gradient averaging leaves;
* ``algo="rs_ag"`` exchanges a bucket as ``reduce_scatter_tensor`` + ``all_gather_into_tensor`` on the (world-padded) flat
  buffer instead of one ``all_reduce``: on xGMI every GPU is one hop from every other, so a reduce-scatter in which rank r
  receives its 1/world slice from all 7 peers at once drives all 7 links (S/8 per link and phase) where a single ring is
  bound by one link.  Same sums, same bucket order, same hooks; on a stream-ordered backend (RCCL) both
  phases are queued from the hook, on a host-ordered one (gloo) the all-gather is issued when the reduce-scatter is
  waited for.  ``all_reduce`` stays the default: RCCL's own all-reduce already picks multi-ring / direct algorithms.
* transport.  RCCL ("nccl") reduces device buffers in place on its own stream.  Any other backend on device tensors (gloo: the
  one-GPU rehearsal of the multi-rank path, tests/test_dp_gpu.py) goes through pinned host buffers that THIS class stages
  (``transport="host"``): the bucket is copied device -> host on a normal-priority copy stream right from the hook, the
  collective runs on the host copies when the bucket is waited for, the result goes back on the same stream.  torch's gloo
  backend can take device tensors itself, but it parks a HIGH-priority stream on an event of the launching stream; on this
  part a waiting high-priority queue starves the process's own normal-priority queues, and a step whose kernels take 45 ms
  ran for 2.5-10 s (profiles/r04_a_*: the per-bucket timeline shows the launching stream reaching a collective's launch point
  seconds after the host queued it, and the collective itself done 10-25 ms later).
* ``stats["timeline"]``: per launched bucket its bytes, who launched it (hook / finish) and host times of launch, wait
  entry and wait return relative to the step's re-arm; with ``profile = True`` on device tensors also stream events at
  the same three points (``timeline()`` resolves them), so an exchange that costs more than its bytes can be read off.
* ``no_sync()`` / ``GradAccumulator``: the gradient-accumulation schedule of ``accelerator.accumulate``
  (train.py:152,335; config.yaml gradient_accumulation_steps): micro-steps inside ``no_sync`` only accumulate
  locally, the last micro-step exchanges the sums.
* A backward that is NOT this reducer's step but reaches its parameters -- the generator step backpropagates through
  ``model.D`` (train.py:196-203), so a reducer over D's parameters sees its hooks fire during ``loss_G.backward()`` --
  must run inside ``no_sync()`` (spelled ``foreign_backward()`` for that use): the gradients pile up locally exactly as
  they do in the reference, and no collective goes out.  Collectives that WERE launched and never finished (a
  synchronising backward without ``finish()``) are waited for by ``zero_grad()`` / the next re-arm before their flat
  buffers are written again, never dropped.

Bucket size: xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce of S bytes costs
about 2*(7/8)*S/153 GB/s; 32 MiB buckets keep each collective ~0.4 ms (far above the ~20 us launch
latency) and give the decoder's 104 MB of gradients 4 chances to overlap.  The same class runs on CPU
tensors with the ``gloo`` backend (tests/test_dp_gloo.py, world size 2) and on HIP tensors with ``gloo``
(two ranks on one GPU: tests/test_dp_gpu.py) or RCCL.
"""
from __future__ import annotations

import contextlib
import time
from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


class GradBucketReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 32 << 20,
                 process_group: Optional[dist.ProcessGroup] = None, average: bool = True,
                 broadcast_parameters: bool = True, rebuild_after_first_step: bool = True, algo: str = "all_reduce",
                 transport: str = "auto"):
        if algo not in ("all_reduce", "rs_ag"):
            raise ValueError(f"GradBucketReducer: algo must be 'all_reduce' or 'rs_ag', not {algo!r}")
        if transport not in ("auto", "device", "host"):
            raise ValueError(f"GradBucketReducer: transport must be 'auto', 'device' or 'host', not {transport!r}")
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.algo = algo
        # RCCL queues collectives on its own stream in issue order: the all-gather may be issued right behind the
        # reduce-scatter.  gloo runs each collective on a worker thread: the second phase waits for the first on the host.
        self._stream_ordered = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.profile = False
        self.average = average
        self._transport_arg, self._copy_stream = transport, None
        self.bucket_bytes = int(bucket_bytes)
        plist = [p for p in params if p.requires_grad]
        if not plist:
            raise ValueError("GradBucketReducer: no trainable parameters")
        dev, dtype = plist[0].device, plist[0].dtype
        for p in plist:
            if p.device != dev or p.dtype != dtype:
                raise ValueError("GradBucketReducer: parameters must share device and dtype")
        self.staged = self.world > 1 and dev.type == "cuda" and (transport == "host" or (transport == "auto" and not self._stream_ordered))
        if self.staged:
            self._copy_stream = torch.cuda.Stream(dev)     # normal priority, ours
        self._registration = plist                         # index space of the recorded / broadcast order
        self._index = {id(p): i for i, p in enumerate(plist)}
        self.params = list(reversed(plist))                # ~ the order autograd finishes them (refined after step 1)
        if broadcast_parameters and self.world > 1:
            self._broadcast_parameters()
        self.buckets: List[dict] = []
        self._build(self.params, cold=())
        self.rebuilt = not rebuild_after_first_step or self.world == 1
        self._fired: List[int] = []                        # registration indices in hook order (first synced backward)
        self._sync = True
        self._in_flat = False                              # gradients live in the flat buffers (after a finish(), world > 1)
        self._handles = []                                 # (work, bucket index, second phase to issue after the wait | None)
        self._next = 0                                     # lowest bucket index not launched yet
        self._t0, self._ev0 = time.perf_counter(), None
        self.stats = self._fresh_stats()                   # bucket indices + timeline, since the last zero_grad()
        self._hooks = [p.register_post_accumulate_grad_hook(self._hook) for p in plist]
        self.zero_grad()

    # ---- construction --------------------------------------------------------------------------
    def _broadcast_parameters(self):
        """Replicas must start identical: rank 0's values win (DDP does the same at construction)."""
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1) for p in self._registration])
            dist.broadcast(flat, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            off = 0
            for p in self._registration:
                p.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()

    def _build(self, ordered, cold):
        """Bucket ``ordered`` (hot parameters, in completion order) then ``cold`` (never seen a gradient) separately."""
        dev, dtype = ordered[0].device if ordered else cold[0].device, self._registration[0].dtype
        self.buckets = []
        for group in (ordered, cold):
            cur, cur_bytes = [], 0
            for p in group:
                nbytes = p.numel() * p.element_size()
                if cur and cur_bytes + nbytes > self.bucket_bytes:
                    self._seal(cur, dev, dtype, cold=group is cold)
                    cur, cur_bytes = [], 0
                cur.append(p)
                cur_bytes += nbytes
            if cur:
                self._seal(cur, dev, dtype, cold=group is cold)
        self._bucket_of = {id(p): bi for bi, b in enumerate(self.buckets) for p in b["params"]}
        self.params = [p for b in self.buckets for p in b["params"]]

    def _seal(self, plist, dev, dtype, cold=False):
        total = sum(p.numel() for p in plist)
        flat = views = shard = None
        if self.world > 1:                     # the communication buffer and each parameter's slice of it
            padded = total + (-total) % self.world if self.algo == "rs_ag" else total
            flat = torch.zeros(padded, device=dev, dtype=dtype)
            if self.algo == "rs_ag":           # this rank's 1/world slice of the sums, between the two phases
                shard = torch.zeros(padded // self.world, device=dev if not self.staged else "cpu", dtype=dtype,
                                    pin_memory=self.staged)
            views, off = [], 0
            for p in plist:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
        host = torch.empty(flat.numel(), dtype=dtype, pin_memory=True) if (self.staged and flat is not None) else None
        self.buckets.append({"params": plist, "flat": flat, "views": views, "shard": shard, "host": host, "numel": total, "pending": len(plist),
                             "n": len(plist), "ready": False, "launched": False, "cold": cold})

    # ---- backward-time protocol ----------------------------------------------------------------
    def _hook(self, param):
        if not self._sync:                     # accumulation micro-step: gradients only pile up locally
            return
        if not self.rebuilt:
            self._fired.append(self._index[id(param)])
        b = self.buckets[self._bucket_of[id(param)]]
        b["pending"] -= 1
        if b["pending"] == 0:
            b["ready"] = True
            self._drain(by_hook=True)

    def _drain(self, by_hook):
        """Launch ready buckets strictly in index order: every rank issues the same sequence of collectives."""
        while self._next < len(self.buckets) and self.buckets[self._next]["ready"]:
            self._launch(self._next, by_hook)
            self.stats["launched_by_hook" if by_hook else "launched_by_finish"].append(self._next)
            self._next += 1

    def _fresh_stats(self):
        return {"launched_by_hook": [], "launched_by_finish": [], "timeline": []}

    def _ms(self):
        return round((time.perf_counter() - self._t0) * 1e3, 3)

    def _event(self, b):
        if self.profile and b["flat"] is not None and b["flat"].is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev
        return None

    def _launch(self, bi, by_hook=False):
        b = self.buckets[bi]
        b["launched"] = True
        if self.world == 1:
            return
        rec = {"bucket": bi, "bytes": b["numel"] * b["params"][0].element_size(), "by": "hook" if by_hook else "finish",
               "launch_ms": self._ms(), "wait_begin_ms": None, "wait_end_ms": None, "_ev_launch": self._event(b), "_ev_done": None}
        self.stats["timeline"].append(rec)
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
        if self.staged:                        # device -> pinned host now, behind the gradients; the collective at wait time
            cs = self._copy_stream
            cs.wait_stream(torch.cuda.current_stream(b["flat"].device))
            with torch.cuda.stream(cs):
                b["host"].copy_(b["flat"], non_blocking=True)
                copied = torch.cuda.Event()
                copied.record(cs)
            self._handles.append((_StagedWork(self, b, copied), rec, None))
            return
        buf = b["flat"]
        if self.algo == "rs_ag":
            h = dist.reduce_scatter_tensor(b["shard"], buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            gather = lambda: dist.all_gather_into_tensor(buf, b["shard"], group=self.group, async_op=True)
            if self._stream_ordered:
                self._handles.append((h, rec, None))
                self._handles.append((gather(), rec, None))
            else:
                self._handles.append((h, rec, gather))
        else:
            self._handles.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), rec, None))

    def _host_exchange(self, b):
        """The bucket's collective on its pinned host copy (blocking: a host-ordered backend)."""
        if self.algo == "rs_ag":
            dist.reduce_scatter_tensor(b["shard"], b["host"], op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(b["host"], b["shard"], group=self.group)
        else:
            dist.all_reduce(b["host"], op=dist.ReduceOp.SUM, group=self.group)

    # ---- step protocol -------------------------------------------------------------------------
    def _wait(self):
        """Outstanding all-reduces complete (stream-side on RCCL) before anything else touches their flat buffers."""
        for h, rec, then in self._handles:
            if rec["wait_begin_ms"] is None:
                rec["wait_begin_ms"] = self._ms()
            h.wait()
            if then is not None:               # host-ordered backend: the second phase goes out once the first is done
                then().wait()
            rec["wait_end_ms"] = self._ms()
            if rec["_ev_launch"] is not None:
                rec["_ev_done"] = self._event(self.buckets[rec["bucket"]])
        self._handles = []

    def timeline(self):
        """``stats["timeline"]`` as plain numbers.  With ``profile`` on device tensors: ``gpu_launch_ms`` / ``gpu_done_ms`` =
        when the launching stream reached the collective's launch point / had the reduced bucket available, relative to the
        step's re-arm (synchronises the device)."""
        out = []
        for rec in self.stats["timeline"]:
            r = {k: v for k, v in rec.items() if not k.startswith("_")}
            if rec["_ev_launch"] is not None and rec["_ev_done"] is not None and self._ev0 is not None:
                rec["_ev_done"].synchronize()
                r["gpu_launch_ms"] = round(self._ev0.elapsed_time(rec["_ev_launch"]), 3)
                r["gpu_done_ms"] = round(self._ev0.elapsed_time(rec["_ev_done"]), 3)
            out.append(r)
        return out

    def _rearm(self):
        self._wait()
        for b in self.buckets:
            b["pending"], b["ready"], b["launched"] = b["n"], False, False
        self._next = 0

    def zero_grad(self, params: Optional[Iterable[torch.nn.Parameter]] = None):
        """Drop the gradients (``set_to_none``) and re-arm the buckets.  ``params``: drop only these (the reference's
        ``optimizer_G.zero_grad()`` clears ``Gd`` alone, train.py:187 -- the encoders' gradients keep accumulating from
        one generator step to the next); the others stay where they are, in the flat buffers
        after a ``finish()``, and the next backward adds to them in place."""
        self._wait()                           # a backward that launched collectives but was never finish()ed
        if params is None:
            for b in self.buckets:
                for p in b["params"]:
                    p.grad = None
            self._in_flat = False
        else:
            for p in params:
                if id(p) not in self._bucket_of:
                    raise ValueError("GradBucketReducer.zero_grad: parameter is not managed by this reducer")
                p.grad = None
        self._rearm()
        self.stats = self._fresh_stats()
        self._t0 = time.perf_counter()
        if self.profile and self._registration[0].is_cuda:
            self._ev0 = torch.cuda.Event(enable_timing=True)
            self._ev0.record()

    @contextlib.contextmanager
    def no_sync(self):
        """Backward passes inside this context accumulate into ``param.grad`` without any exchange (the first k-1
        micro-steps of ``accelerator.accumulate``, train.py:152); the next backward outside it exchanges the sums."""
        prev, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = prev

    foreign_backward = no_sync                 # a backward of ANOTHER network's loss that reaches these parameters

    def finish(self):
        """Call after a synchronising ``loss.backward()``: launches, in index order, the buckets the hooks could not
        (a parameter without a gradient this step contributes zeros), waits for every collective, and -- once, after
        the first step -- re-buckets the parameters in the order autograd completed them."""
        for b in self.buckets:
            b["ready"] = True
        self._drain(by_hook=False)
        self._wait()
        self._in_flat = self.world > 1         # every gradient now lives in its bucket's flat buffer
        if not self.rebuilt:
            self._rebuild()
        self._rearm()                          # a caller that never zeroes (the reference's encoder gradients) may go on

    def _rebuild(self):
        """Parameters in rank 0's hook order first, never-fired ones ("cold") in trailing buckets of their own.  The
        gradients just reduced stay valid: they are moved into the new flat buffers."""
        n = len(self._registration)
        order = torch.full((n,), -1, dtype=torch.int64)
        seen = list(dict.fromkeys(self._fired))
        order[:len(seen)] = torch.tensor(seen, dtype=torch.int64) if seen else order[:0]
        if self.world > 1:                                  # every rank adopts rank 0's order
            dev = self._registration[0].device
            t = order.to(dev)
            dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            order = t.cpu()
        hot_idx = [int(i) for i in order.tolist() if i >= 0]
        hot_set = set(hot_idx)
        hot = [self._registration[i] for i in hot_idx]
        cold = [p for i, p in reversed(list(enumerate(self._registration))) if i not in hot_set]
        grads = {id(p): p.grad for p in self._registration}
        self._build(hot, cold)
        if self.world > 1:
            with torch.no_grad():
                for b in self.buckets:
                    src = [grads[id(p)] for p in b["params"]]
                    dst = [v for v, g in zip(b["views"], src) if g is not None]
                    src = [g for g in src if g is not None]
                    if dst:
                        torch._foreach_copy_(dst, src)
                    for p, v in zip(b["params"], b["views"]):
                        if grads[id(p)] is not None:
                            p.grad = v
        self.rebuilt = True
        self._fired = []
        self._rearm()

    # ---- gradient norm / clipping ----------------------------------------------------------------
    def _grads(self):
        return [p.grad for p in self.params if p.grad is not None]

    def _tensors(self):
        return [b["flat"] for b in self.buckets] if self._in_flat else self._grads()

    def _partial_norms(self) -> Optional[torch.Tensor]:
        """fp64 vector of per-tensor L2 norms (one multi-tensor launch), or None when there is no gradient at all."""
        tensors = self._tensors()
        return torch.stack(torch._foreach_norm(tensors)).double() if tensors else None

    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm of the (already synchronised, hence rank-identical) gradients -- what
        ``clip_grad_norm_`` (train.py:207-208) needs; no further collective and no fp64 copy of the buckets:
        one multi-tensor norm launch, the handful of partial norms combined in fp64."""
        parts = self._partial_norms()
        if parts is None:
            return torch.zeros((), device=self.params[0].device)
        return torch.linalg.vector_norm(parts).float()

    def scale_(self, coef: torch.Tensor):
        tensors = self._tensors()
        if tensors:
            torch._foreach_mul_(tensors, coef)

    def clip_(self, max_norm: float) -> torch.Tensor:
        total = self.grad_norm()
        self.scale_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
        return total

    def bytes_per_step(self) -> int:
        return sum(b["numel"] * b["params"][0].element_size() for b in self.buckets)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


class _StagedWork:
    """A bucket on its way through the host (``GradBucketReducer.staged``): ``wait()`` = the copy has landed, the collective
    runs on the host buffer, the result is queued back to the device and the current stream waits for it."""

    def __init__(self, reducer, bucket, copied):
        self.reducer, self.bucket, self.copied = reducer, bucket, copied

    def wait(self):
        r, b = self.reducer, self.bucket
        self.copied.synchronize()
        r._host_exchange(b)
        with torch.cuda.stream(r._copy_stream):
            b["flat"].copy_(b["host"], non_blocking=True)
        torch.cuda.current_stream(b["flat"].device).wait_stream(r._copy_stream)


class GradAccumulator:
    """``accelerator.accumulate(model)`` + ``accelerator.backward(loss)`` (train.py:152,182,205) on the reducer:
    ``backward(loss)`` scales the loss by 1/k, runs the first k-1 micro-steps of every group of k without exchange and
    the k-th with it; ``sync_gradients`` tells the caller when to clip and step (train.py:207)."""

    def __init__(self, reducer: GradBucketReducer, steps: int = 1):
        if steps < 1:
            raise ValueError("GradAccumulator: steps must be >= 1")
        self.reducer, self.steps, self._micro = reducer, int(steps), 0
        self.sync_gradients = False

    def backward(self, loss):
        self._micro += 1
        self.sync_gradients = self._micro % self.steps == 0
        loss = loss / self.steps if self.steps > 1 else loss
        if self.sync_gradients:
            loss.backward()
            self.reducer.finish()
        else:
            with self.reducer.no_sync():
                loss.backward()
        return self.sync_gradients

    def zero_grad(self):
        """Only after a synchronised step (gradients of unfinished groups must survive)."""
        if self._micro % self.steps == 0:
            self.reducer.zero_grad()


def clip_grad_norm_(reducers: Sequence[GradBucketReducer], max_norm: float) -> torch.Tensor:
    """``accelerator.clip_grad_norm_(model.parameters(), v)`` (train.py:207-208) when the model's parameters are spread over
    several reducers (one per network: ``D`` has its own step and its own exchange): ONE global norm over all of them, one
    coefficient, every reducer's gradients scaled in place.  Every reducer must have been ``finish()``ed since its last
    backward -- the norm is only rank-identical over exchanged gradients."""
    parts = [q for q in (r._partial_norms() for r in reducers) if q is not None]
    if not parts:
        return torch.zeros((), device=reducers[0].params[0].device)
    total = torch.linalg.vector_norm(torch.cat(parts)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for r in reducers:
        r.scale_(coef)
    return total


def shard_batch(global_batch: int, rank: int, world: int):
    """Rank r takes samples [r*B/world, (r+1)*B/world) -- SURVEY.md 8e."""
    per = global_batch // world
    return rank * per, (rank + 1) * per

"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The forward / sampling path shards by independent clips: no data-path collective.
The only exchanges are the timing reduction of the benchmark and (training) the gradient all-reduce.
"""
from __future__ import annotations

import os
import time

import torch


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str, device=None):
    import torch.distributed as td
    rank, _, world = env_rank()
    if world > 1 and not td.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        td.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_clips(n_clips: int, rank: int, world: int):
    """Contiguous, balanced partition of clip indices across ranks (independent units, no exchange)."""
    base, rem = divmod(n_clips, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def timed_steps(step_fn, steps: int, warmup: int, sync=None, device=None):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + device sync on both sides;
    returns the MAX elapsed seconds over ranks (the bench contract)."""
    import torch.distributed as td
    dist = td.is_available() and td.is_initialized() and td.get_world_size() > 1
    sync = sync or (lambda: None)
    for _ in range(warmup):
        step_fn()
    sync()
    if dist:
        td.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync()
    if dist:
        td.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed


# ----------------------------------------------------------------------------- gradient exchange (training)
ALIGN = 64  # elements: every tensor starts on a 256-byte boundary of its arena (kernels need 16-byte operands)


# Groups of parameters that must sit NEXT TO EACH OTHER in the arenas, in the given order, so that their concatenation is a
# view (the fused Q/K/V projection of an encoder layer: autograd.FusedAlias).  Set by the Trainer before it builds the arenas;
# a group is honoured only if all members are trainable and whole multiples of ALIGN elements.
ADJACENT = []


def flat_layout(params):
    """Arena layout shared by the parameter arena, the gradient arena and the Adam moments: trainable parameters
    in REVERSE registration order (ADJACENT groups pulled together at their first member's place), each offset rounded
    up to ALIGN elements.  Returns (ordered params, offsets, total)."""
    ps = [p for p in params if p.requires_grad][::-1]
    have = {id(p) for p in ps}
    for grp in ADJACENT:
        if not all(id(p) in have and p.numel() % ALIGN == 0 for p in grp):
            continue
        ids = {id(p) for p in grp}
        first = min(i for i, p in enumerate(ps) if id(p) in ids)
        rest = [p for p in ps if id(p) not in ids]
        n_before = sum(1 for p in ps[:first] if id(p) not in ids)
        ps = rest[:n_before] + list(grp) + rest[n_before:]
    offs, off = [], 0
    for p in ps:
        offs.append(off)
        off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
    return ps, offs, off


class GradBucketReducer:
    """Bucketed gradient all-reduce overlapped with backward.

    * every trainable parameter's ``.grad`` is a VIEW into one flat fp32 arena, laid out in REVERSE registration
      order (heads -> decoder layers 7..0 -> audio_feature_map -> encoder layers 11..0 -> ...), i.e. the order in
      which backward produces gradients, cut into ~bucket_mb buckets;
    * a post-accumulate-grad hook per parameter counts arrivals; when a bucket is complete its all-reduce (sum)
      is launched on a dedicated side stream behind an event recorded on the compute stream, so the collective
      (RCCL over xGMI with backend "nccl") runs under the rest of backward;
    * ``finish()`` launches buckets that did not complete (parameters that received no gradient this step stay
      zero in the arena: LayerDrop-skipped layers, unused null tokens), makes the compute stream wait for the
      side stream and returns the flat arena; the 1/world_size factor is folded into the optimizer step
      (``msmd_adam_step(grad_scale=1/world)``).
    Frozen parameters (requires_grad=False) are excluded.  With world_size 1 nothing is communicated.
    The KL term of the reference is a batch SUM (utils/common.py:454): scale its weight by world_size to keep
    parity with a single-process global batch (`kl_weight_scale`).
    """

    def __init__(self, params, bucket_mb: float = 32.0, process_group=None):
        import torch.distributed as td
        self.td = td
        self.group = process_group
        self.world = td.get_world_size(process_group) if td.is_initialized() else 1
        self.params, offs, total = flat_layout(params)
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        self.arena = torch.zeros(total, device=dev, dtype=torch.float32)
        self.buckets = []  # (start, end, [param indices])
        cap = int(bucket_mb * (1 << 20) / 4)
        start = 0
        members = []
        self.slot = {}
        for i, (p, off) in enumerate(zip(self.params, offs)):
            n = p.numel()
            p.grad = self.arena[off:off + n].view_as(p)
            self.slot[id(p)] = (len(self.buckets), off, n)
            members.append(i)
            end = offs[i + 1] if i + 1 < len(offs) else total
            if end - start >= cap:
                self.buckets.append((start, end, members))
                start, members = end, []
        if members:
            self.buckets.append((start, total, members))
        self.pending = [len(m) for _, _, m in self.buckets]
        self.launched = [False] * len(self.buckets)
        # Hook-driven launches go out strictly in bucket-index order (`next_launch`): ranks may complete buckets in different
        # orders -- LayerDrop skips a different layer on every rank in eager mode, so its bucket never completes there -- and
        # a collective sequence that differs between ranks mismatches payloads or hangs.  A bucket that is complete but
        # sits behind an incomplete one waits for finish(), which launches the rest in the same index order everywhere.
        self.ready = [False] * len(self.buckets)
        self.next_launch = 0
        self.cuda = dev.type == "cuda"
        self.side = torch.cuda.Stream(device=dev) if self.cuda else None
        self.works = []
        self.enabled = True  # set False on gradient-accumulation micro-steps (reference training_script.py:199)
        self.mute = False    # measurement switch (bench.py --mode train): run the step with NO exchange at all
        # One-GPU rehearsal of the overlapped exchange (tests/test_train_gpu.py, tools/dp_sidestream_check.py): with
        # world_size 1 there is nothing to reduce, so `standin(view)` -- any in-place, value-preserving device work on the
        # bucket -- is run on the side stream exactly where the all-reduce would be: same events, same second queue
        # active under the rest of backward.  None (default) = world 1 launches nothing.
        self.standin = None
        # Completion tracking that does not depend on autograd hooks alone: every gradient WRITE of a parameter -- an
        # autograd accumulation (hook below) or a kernel that adds straight into the arena view (autograd.GRAD_WRITTEN)
        # -- is noted in order.  `trace_begin()` / `trace_end()` record one backward's write sequence; afterwards
        # `on_write` fires `on_bucket_final(b)` at the write after which bucket b receives nothing more this backward
        # (parameters are written once per window, so "first arrival" is not "final").
        self.trace = None
        self.final_pos = None
        self.write_count = 0
        self.on_bucket_final = None
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._hook)

    @property
    def kl_weight_scale(self):
        return float(self.world)

    def zero_grad(self):
        """Zero the arena (grads are views) and re-arm the arrival counters."""
        self.arena.zero_()
        self.begin_backward()

    def begin_backward(self):
        """Re-arm the arrival counters.  Called before EVERY backward (Trainer.step), not only after an optimizer step:
        with gradient accumulation a bucket that received only part of its gradients in an earlier micro-step (a
        LayerDrop-skipped layer, an unused null token) must not reach zero early in the stepping micro-step and be
        reduced while backward is still adding into it."""
        self.pending = [len(m) for _, _, m in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.ready = [False] * len(self.buckets)
        self.next_launch = 0
        self.works = []
        self.write_count = 0

    # ---- write-sequence tracking (hipGraph segments; see __init__)
    def trace_begin(self):
        self.trace, self.final_pos, self.write_count = [], None, 0

    def trace_end(self):
        """-> {write position: [buckets final after that write]} of the traced backward."""
        last = {}
        for i, b in enumerate(self.trace):
            last[b] = i
        self.trace = None
        self.final_pos = {}
        for b, i in last.items():
            self.final_pos.setdefault(i, []).append(b)
        self.write_count = 0
        return self.final_pos

    def on_write(self, p):
        """A gradient of parameter p was just accumulated into / written to the arena (stream order = call order)."""
        ent = self.slot.get(id(p))
        if ent is None:
            return
        if self.trace is not None:
            self.trace.append(ent[0])
            return
        if self.final_pos is not None and self.on_bucket_final is not None:
            done = self.final_pos.get(self.write_count)
            self.write_count += 1
            if done:
                for b in done:
                    self.on_bucket_final(b)

    def _hook(self, p):
        b, off, n = self.slot[id(p)]
        if p.grad.data_ptr() != self.arena[off:off + n].data_ptr():  # autograd replaced the view: copy back
            self.arena[off:off + n].copy_(p.grad.reshape(-1))
            p.grad = self.arena[off:off + n].view_as(p)
        self.on_write(p)
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self.ready[b] = True
        if self.enabled:
            while self.next_launch < len(self.buckets) and self.ready[self.next_launch]:
                self._launch(self.next_launch)
                self.next_launch += 1

    def exchange_only(self):
        """All buckets' all-reduce with nothing to overlap with (bench.py: the exchange's own duration).  The arena's
        contents are scaled back afterwards so repeated calls do not overflow."""
        self.begin_backward()
        for b in range(len(self.buckets)):
            self._launch(b)
        self._join()
        self.arena.mul_(1.0 / self.world)
        self.begin_backward()

    def _join(self):
        for w in self.works:
            if w is not None:
                w.wait()   # NCCL: the CURRENT STREAM waits for the collective (no host block); gloo: host wait
        self.works = []
        if self.cuda and (self.world > 1 or self.standin is not None):
            torch.cuda.current_stream().wait_stream(self.side)

    def _launch(self, b):
        if self.launched[b] or self.mute or (self.world == 1 and self.standin is None):
            self.launched[b] = True
            return
        self.launched[b] = True
        start, end, _ = self.buckets[b]
        view = self.arena[start:end]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.side.wait_event(ev)
            with torch.cuda.stream(self.side):
                if self.world == 1:
                    self.standin(view)
                else:
                    self.works.append(self.td.all_reduce(view, op=self.td.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.works.append(self.td.all_reduce(view, op=self.td.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Call after backward on a stepping iteration: returns (flat_grad_arena, grad_scale)."""
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)
        self._join()
        return self.arena, 1.0 / self.world


def flatten_parameters(params):
    """Re-home trainable parameters into one flat fp32 arena (views, `flat_layout` order / alignment) so the fused
    Adam kernel (msmd_adam_step) updates the whole model in one launch.  Returns the flat tensor."""
    ps, offs, total = flat_layout(params)
    flat = torch.zeros(total, device=ps[0].device, dtype=torch.float32)
    for p, off in zip(ps, offs):
        n = p.numel()
        flat[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat[off:off + n].view_as(p)
    return flat

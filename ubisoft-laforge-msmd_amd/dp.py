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
class RcclComm:
    """One RCCL communicator per process through the C ABI (include/msmd_hip.h: msmd_comm_unique_id / msmd_comm_init /
    msmd_allreduce_bucket / msmd_comm_destroy; csrc/comm.hip resolves librccl with dlopen at first use).  Rank 0's 128-byte
    id travels over the torch.distributed group the job already has for its rendezvous (any backend: it is only a side
    channel here); a one-rank job needs no side channel at all.  The collectives themselves go straight to RCCL on the HIP
    stream the caller names -- no torch.distributed in the data path."""

    _DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}     # MSMD_F32 / MSMD_BF16 / MSMD_F16

    def __init__(self, device=None, group=None):
        import ctypes
        import torch.distributed as td
        from . import _lib
        self._lib = _lib
        self.lib = _lib.load()
        dist = td.is_available() and td.is_initialized()
        self.world = td.get_world_size(group) if dist else 1
        self.rank = td.get_rank(group) if dist else 0
        ident = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _lib.check(self.lib.msmd_comm_unique_id(ident), "msmd_comm_unique_id")
        if self.world > 1:
            box = [bytes(ident.raw) if self.rank == 0 else None]
            td.broadcast_object_list(box, src=td.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = ctypes.create_string_buffer(box[0], 128)
        self.comm = ctypes.c_void_p()
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        with torch.cuda.device(dev):
            _lib.check(self.lib.msmd_comm_init(ctypes.byref(self.comm), self.world, self.rank, ident), "msmd_comm_init")
        self.device = dev

    def all_reduce(self, t, stream=None):
        """t <- sum over ranks, in place, enqueued on `stream` (default: torch's current stream)."""
        if not t.is_cuda or not t.is_contiguous():
            raise ValueError("RcclComm.all_reduce takes a contiguous device tensor")
        st = torch.cuda.current_stream(t.device) if stream is None else stream
        self._lib.check(self.lib.msmd_allreduce_bucket(self.comm, t.data_ptr(), t.numel(), self._DT[t.dtype], st.cuda_stream),
                        "msmd_allreduce_bucket")

    @property
    def version(self):
        """ncclGetVersion of the librccl behind the C ABI (a copy torch already mapped is adopted, csrc/comm.hip)."""
        return int(self.lib.msmd_comm_version())

    def destroy(self):
        if getattr(self, "comm", None) is not None and self.comm.value:
            torch.cuda.synchronize(self.device)
            self._lib.check(self.lib.msmd_comm_destroy(self.comm), "msmd_comm_destroy")
        self.comm = None

    def __del__(self):
        # ncclCommDestroy is a COLLECTIVE-ordered call: run from the garbage collector or at interpreter teardown it can hang
        # when the peer ranks are not destroying their communicators at the same point.  Only an explicit destroy() tears the
        # communicator down; a forgotten one is dropped with the process.
        self.comm = None


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

    def __init__(self, params, bucket_mb: float = 32.0, process_group=None, comm=None, bucket_dtype=None,
                 exchange_at_world_1=False):
        """comm (RcclComm): the buckets' all-reduce goes through the C ABI (msmd_allreduce_bucket) on the side stream instead of
        torch.distributed.  bucket_dtype (torch.bfloat16 / torch.float16): every bucket is cast into a 16-bit staging arena,
        summed there and cast back -- half the bytes over xGMI (260 MB instead of 519 MB per step) for one
        16-bit rounding of each rank's contribution plus the ring's 16-bit partial sums.  exchange_at_world_1: run the
        collective also in a one-rank job (RCCL executes, the sum over one rank is the identity): the one-GPU rehearsal of
        the overlapped exchange with the real library in place of the stand-in."""
        import torch.distributed as td
        self.td = td
        self.group = process_group
        self.comm = comm
        self.bucket_dtype = bucket_dtype
        self.exchange_at_world_1 = bool(exchange_at_world_1)
        self.world = td.get_world_size(process_group) if td.is_initialized() else 1
        if comm is not None and comm.world != self.world:
            raise ValueError(f"communicator of {comm.world} ranks in a job of {self.world}")
        self.params, offs, total = flat_layout(params)
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        self.arena = torch.zeros(total, device=dev, dtype=torch.float32)
        self.stage = torch.zeros(total, device=dev, dtype=bucket_dtype) if bucket_dtype is not None else None
        if bucket_dtype == torch.float16 and self.world > 1:
            import warnings
            # contributions are staged / world to keep the sum inside fp16's range, with no loss scaling on this path:
            # gradient elements below ~6e-5 * world go subnormal, below ~6e-8 * world they are lost.  bf16 has fp32's range.
            warnings.warn("GradBucketReducer: fp16 buckets flush gradient elements below ~6e-8 x world_size; bf16 is the "
                          "supported 16-bit bucket type", stacklevel=2)
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
        if self.cuda and (self.world > 1 or self.standin is not None or self.exchange_at_world_1):
            torch.cuda.current_stream().wait_stream(self.side)

    def _reduce(self, view, start, end):
        """The collective on one bucket (caller is on the side stream on CUDA): through the C ABI when a communicator was
        given, torch.distributed otherwise; through the 16-bit staging arena when bucket_dtype is set."""
        buf = view
        if self.stage is not None:
            buf = self.stage[start:end]
            if buf.dtype == torch.float16 and self.world > 1:
                # fp16 has no headroom for a sum over ranks (max 65 504, no loss scaling on this path): stage contribution / world,
                # so the SUM is the mean and cannot exceed the largest single contribution; scaled back after the exchange.
                # bf16 staging (fp32's exponent range) needs no such guard and is the recommended 16-bit bucket type.
                torch.mul(view, 1.0 / self.world, out=buf)   # scaled and cast straight into the staging arena (no bucket-sized temporary)
            else:
                buf.copy_(view)                  # fp32 -> 16 bit (one rounding of this rank's contribution)
        if self.comm is not None:
            self.comm.all_reduce(buf, self.side if self.cuda else None)
        elif self.world > 1:
            w = self.td.all_reduce(buf, op=self.td.ReduceOp.SUM, group=self.group, async_op=True)
            if self.stage is None:
                self.works.append(w)
            else:
                w.wait()     # NCCL: this (side) stream waits for the collective, no host block; gloo: host wait -- then the copy back
        if self.stage is not None:
            view.copy_(buf)
            if buf.dtype == torch.float16 and self.world > 1:
                view.mul_(float(self.world))

    def _launch(self, b):
        exchanging = self.world > 1 or (self.exchange_at_world_1 and self.comm is not None)
        if self.launched[b] or self.mute or (not exchanging and self.standin is None):
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
                if exchanging:
                    self._reduce(view, start, end)
                else:
                    self.standin(view)
        else:
            self._reduce(view, start, end)

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

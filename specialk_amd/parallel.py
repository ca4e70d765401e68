"""Data-parallel gradient exchange: one process per GPU, bucketed RCCL all-reduce (SUM)
overlapped with backward.

Replaces the reference's single-process nn.DataParallel wrapping of encoder and decoder
(train.py:385-388).  The loss is a token SUM (lib/TransformerModel.py:370,373), so summing the
per-rank gradients reproduces the gradient of the concatenated global batch; no averaging.

Layout: parameters are packed, in reverse registration order (the order backward produces
gradients), into flat fp32 buckets of ~`bucket_mb`; each parameter's .grad is a view into its
bucket, so a bucket is reduced in place with a single collective as soon as its last gradient
has been accumulated, on RCCL's own stream while backward continues.  Bucket size: xGMI is
point-to-point (7 links x ~153 GB/s per GPU) and RCCL's ring moves 2(N-1)/N of the bucket over
one link pair per step, so buckets must be large enough (tens of MB) to reach link bandwidth
yet small enough that the last bucket's exposed tail stays short.

While gradients are being exchanged, no parameter carries a .grad at the start of a step and the buckets are
NOT zeroed (392 MB per step for the base model): the first producer of a gradient overwrites its slice.
  * The grouped weight-gradient launches and the embedding / generator products ask `claim(p)` for the slice
    and write it in place (first producer of the step: overwrite; later producers of a shared matrix — tied
    embeddings and generator — accumulate into it).
  * Everything else reaches .grad through autograd, which adopts the freshly computed tensor without a kernel
    when .grad is empty (with a bucket view attached it would ADD into it with a kernel per tensor: 156
    one-dimensional parameters in a 6+6 Transformer, ~5 us each).  When a bucket is complete those tensors are
    copied into it by one multi-tensor copy, after which .grad is the bucket view again.
  * A parameter that got no gradient in a step has its slice zeroed in finish(), before the bucket is reduced.
  * A producer that writes a gradient in place announces the parameter with direct(p) when its backward runs and
    reports it with mark_ready([p]) once the gradient is written; such a parameter must not reach .grad through
    autograd in the same step as well.
  * Readiness is counted per parameter: a matrix that several layers share is reported by mark_ready once per
    grouped launch that wrote a part of its gradient, and its bucket must wait for the last of them.  How many
    reports a parameter makes per step is learned (the largest count seen so far, 1 to begin with); in the step
    that first shows a further report the bucket may already be on its way, and the late part is then written to a
    tensor of its own (late_target), reduced by itself in finish() and added to the bucket's sum.
"""
import os

import torch
import torch.distributed as dist


class GradBucketReducer:
    def __init__(self, params, bucket_mb=32, process_group=None, always_reduce=False):
        """always_reduce: run the collectives even in a group of one rank (a rehearsal of the RCCL path on a
        single card: the sum over one rank leaves the gradients as they are)."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (always_reduce and dist.is_initialized())
        self.params = [p for p in params if p.requires_grad]
        seen, uniq = set(), []
        for p in self.params:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.params = uniq
        cap = int(bucket_mb * (1 << 20) / 4)
        self.buckets = []      # list of (flat tensor, [params])
        cur, cur_n = [], 0
        for p in reversed(self.params):
            if cur and cur_n + p.numel() > cap:
                self._close(cur, cur_n)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self._close(cur, cur_n)
        self._pending = {}
        self._handles = []
        self._hooks = []
        # A transport that cannot read device memory (gloo) gets the buckets through a host copy at
        # finish(); in-backward overlap is only for RCCL, which reduces the device buffers in place.
        self._staged = (self.active and self.buckets and self.buckets[0][0].is_cuda
                        and dist.get_backend(process_group) != "nccl")
        self._bucket_of = {p.data_ptr(): bi for bi, (_, plist) in enumerate(self.buckets) for p in plist}
        self._views, self._claimed = {}, set()
        self._expected, self._reports = {}, {}   # id(p) -> mark_ready reports per step (learned) / so far this step
        self._launched, self._late = set(), []   # buckets whose all-reduce has started / [(param, late gradient part)]
        self._direct = set()                     # id(p) of parameters announced with direct() this step
        self.gathered_matrix_bytes = 0  # diagnostics: matrix gradients that took the copy instead of a direct write
        self.late_parts = 0             # diagnostics: gradient parts produced after their bucket had left
        # SK_DP_RESERVE_CUS=n: from a step's first all-reduce until finish() the large-tile GEMMs plan their rounds
        # for n CUs fewer (skhip_set_reserved_cus): RCCL's kernels hold CUs meanwhile, and a launch planned for all
        # of them then needs an extra, almost empty round.  0 (default): off — how many CUs RCCL takes has to be
        # measured on a multi-GPU node.
        self.reserve_cus = int(os.environ.get("SK_DP_RESERVE_CUS", "0"))
        self._reserved = False
        for flat, plist in self.buckets:
            off = 0
            for p in plist:
                self._views[id(p)] = flat[off:off + p.numel()].view(p.shape)
                off += p.numel()
        if self.active and not self._staged:
            for bi, (_, plist) in enumerate(self.buckets):
                for p in plist:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        self.zero_grad()

    def _close(self, plist, n):
        flat = torch.zeros(n, dtype=torch.float32, device=plist[0].device)
        self.buckets.append((flat, list(plist)))
        self._reattach(flat, plist)

    def _gather_late(self, bi):
        """Copy the gradients that autograd adopted as tensors of their own into the bucket."""
        flat, plist = self.buckets[bi]
        views, grads, off = [], [], 0
        for p in plist:
            n = p.numel()
            if p.grad is not None and p.grad.data_ptr() != flat.data_ptr() + 4 * off:
                views.append(flat[off:off + n].view(p.shape))
                grads.append(p.grad)
                self.gathered_matrix_bytes += 4 * n * (p.dim() > 1)
            off += n
        if views:
            torch._foreach_copy_(views, grads)
            off = 0
            for p in plist:
                n = p.numel()
                if p.grad is not None:
                    p.grad = flat[off:off + n].view(p.shape)
                off += n

    def _count(self, bi):
        """One more parameter of bucket bi has its gradient: the last one starts the bucket's all-reduce."""
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            flat = self.buckets[bi][0]
            self._gather_late(bi)
            self._reserve(True)
            self._launched.add(bi)
            self._handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _make_hook(self, bi):
        def hook(p):
            # (autograd runs a parameter's accumulation node, and with it this hook, even when the producer handed it
            # no gradient: a parameter announced with direct() counts when mark_ready reports it, not here)
            if id(p) not in self._direct:
                self._count(bi)
        return hook

    def direct(self, p):
        """Announce, before the backward pass reaches the accumulation node of p, that this step's gradient of p is
        written by its producer in place and reported with mark_ready (ops.batched_weight_grads does so when it
        queues a product)."""
        self._direct.add(id(p))

    def _reserve(self, on):
        if self.reserve_cus > 0 and on != self._reserved:
            from . import _lib
            _lib.set_reserved_cus(self.reserve_cus if on else 0)
            self._reserved = on

    def bucket_view(self, p):
        """The slice of its bucket that holds the gradient of p, for producers that write a gradient in place
        (ops.batched_weight_grads writes gradients there and then calls mark_ready); None when gradients
        are staged through host copies or nothing is reduced (the caller then allocates a tensor of its own)."""
        if not self.active or self._staged:
            return None
        return self._views.get(id(p))

    def claim(self, p):
        """(slice of the bucket that holds the gradient of p, accumulate?) for a producer that writes in place:
        the first claim of a step must overwrite the slice (it still holds the previous step's sum), later ones
        — further uses of a shared matrix — add to it.  (None, False) when there is nothing to write into."""
        view = self.bucket_view(p)
        if view is None or self._bucket_of[p.data_ptr()] in self._launched:
            return None, False
        first = id(p) not in self._claimed
        self._claimed.add(id(p))
        return view, not first

    def late_target(self, p):
        """A tensor of its own for a part of the gradient of p that is produced after the bucket of p has started
        its all-reduce (a shared matrix's further use, in the first step that shows it), or None when the bucket
        can still be written.  finish() reduces the tensor and adds it to the bucket's sum."""
        if not self.active or self._staged or self._bucket_of.get(p.data_ptr()) not in self._launched:
            return None
        t = torch.empty(p.shape, dtype=torch.float32, device=p.device)
        self.late_parts += 1
        self._late.append((p, t))
        return t

    def mark_ready(self, params):
        """Cue for gradients that were written into the bucket views directly instead of being
        accumulated by autograd (ops.batched_weight_grads).  A parameter counts towards its bucket when the last
        of its expected reports of the step has come in (see the module docstring)."""
        if self.active and not self._staged:
            for p in params:
                k = self._reports[id(p)] = self._reports.get(id(p), 0) + 1
                want = self._expected.get(id(p), 1)
                if k == want:
                    self._count(self._bucket_of[p.data_ptr()])
                elif k > want:   # a further use, seen for the first time: expected from the next step on
                    self._expected[id(p)] = k

    def zero_grad(self):
        """Start a step.  Exchanging gradients: every .grad is dropped and nothing is zeroed (see the module
        docstring).  Otherwise (one rank, nothing to reduce): the buckets are zeroed in place and stay attached as
        the parameters' .grad."""
        for bi, (flat, plist) in enumerate(self.buckets):
            if not self.active:
                flat.zero_()
            self._pending[bi] = len(plist)
            self._reattach(flat, plist)  # also repairs views a foreign zero_grad(set_to_none=True) detached
        self._claimed = set()
        self._reports, self._launched, self._late, self._direct = {}, set(), [], set()
        self._handles = []

    def _reattach(self, flat, plist):
        off = 0
        for p in plist:
            if self.active:
                p.grad = None
            elif p.grad is None or p.grad.data_ptr() != flat.data_ptr() + 4 * off:
                p.grad = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()

    def _zero_missing(self, bi):
        """Parameters without a gradient this step (unused ones) contribute zeros to the sum."""
        flat, plist = self.buckets[bi]
        off = 0
        for p in plist:
            if p.grad is None and id(p) not in self._claimed:
                flat[off:off + p.numel()].zero_()
            off += p.numel()

    def finish(self):
        """Wait for every bucket's all-reduce; buckets whose parameters got no gradient this
        step (unused parameters) are reduced here so that all ranks stay in lock-step."""
        if self._staged:
            for bi, (flat, _) in enumerate(self.buckets):
                self._zero_missing(bi)
                self._gather_late(bi)
                host = flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                flat.copy_(host)
                self._pending[bi] = 0
        elif self.active:
            for bi, (flat, _) in enumerate(self.buckets):
                if self._pending[bi] > 0:
                    self._zero_missing(bi)
                    self._gather_late(bi)
                    self._handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                    self._pending[bi] = 0
            for h in self._handles:
                h.wait()
            for p, t in self._late:   # (same order on every rank: the order of the backward pass)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                self._views[id(p)].add_(t)
            self._late = []
            self._reserve(False)
        if self.active:  # every .grad is the bucket view again (parameters without a gradient read as zero)
            for flat, plist in self.buckets:
                off = 0
                for p in plist:
                    if p.grad is None:
                        p.grad = flat[off:off + p.numel()].view(p.shape)
                    off += p.numel()
        self._handles = []


def all_reduce_scalars(values, device):
    """Sum a few python/torch scalars over ranks in one collective (loss, n_word, n_correct)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":  # only RCCL reads device memory
            t = torch.as_tensor(values, dtype=torch.float64, device=device)
        else:
            t = torch.as_tensor([float(v) for v in values], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.to(device)
    return torch.as_tensor(values, dtype=torch.float64, device=device)

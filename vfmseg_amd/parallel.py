"""Data parallelism for the hot path: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm),
gloo on CPU for tests.  Replaces mmengine's MMDistributedDataParallel + SyncBatchNorm plumbing
(configs/_base_/default_runtime.py:5, tools/dist_train.sh:9-17).

Only ~16.6 M fp32 gradients (66.5 MB) exist per step, produced in the order VFMHead -> LinearHead -> LoRA layers
L-1..0, and they already sit in ONE flat buffer in that order (optim.FusedAdamW).  The all-reduce is therefore a
handful of large in-place collectives on contiguous slices, issued on a side HIP stream as soon as a slice is
final, overlapping the remaining backward; xGMI is point-to-point so few large buckets beat many small ones.
"""
import os

import torch
import torch.distributed as dist


def single_rank_rehearsal():
    """VFMSEG_DIST_SINGLE=1: run the whole DP plumbing (process group, parameter broadcast, bucketed all-reduce on the side
    stream, SyncBN exchange) with a world of ONE rank.  A box with one GPU cannot host two RCCL ranks, so this is how the `nccl`
    code path itself gets exercised there (tests/test_dp_equivalence_gpu.py); results must equal the non-distributed step."""
    return os.environ.get("VFMSEG_DIST_SINGLE", "0") == "1"


def init_from_env(backend=None):
    """torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT). Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or single_rank_rehearsal()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("VFMSEG_DIST_BACKEND", backend)
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_total, rank, world, seed=0):
    """mmengine InfiniteSampler: one global shuffled index stream, rank r takes every world-th element."""
    g = torch.Generator().manual_seed(seed)
    perm = torch.randperm(n_total, generator=g).tolist()
    return perm[rank::world]


def broadcast_params(model, src=0):
    """DDP-constructor semantics: rank 0's parameters and buffers win (C3)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not single_rank_rehearsal()):
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src)


def make_buckets(names, offsets, max_lora_buckets=2):
    """Contiguous [start, end) slices of the flat gradient buffer in production order.
    names/offsets as in FusedAdamW (offsets has len(names)+1 entries)."""
    groups = []
    cur = None
    for i, nm in enumerate(names):
        g = "lora" if "lora_" in nm else ("aux_decoder" if nm.startswith("aux_decoder") else "decode_head")
        if g != cur:
            groups.append([g, offsets[i], offsets[i + 1]])
            cur = g
        else:
            groups[-1][2] = offsets[i + 1]
    out = []
    for g, a, b in groups:
        if g == "lora" and max_lora_buckets > 1:
            # split the LoRA run at parameter boundaries into roughly equal halves (layers L-1.. then ..0)
            idx = [i for i, nm in enumerate(names) if "lora_" in nm]
            tgt = (b - a) / max_lora_buckets
            start = a
            k = 1
            for i in idx:
                if offsets[i + 1] - a >= tgt * k and k < max_lora_buckets:
                    out.append((f"lora{k - 1}", start, offsets[i + 1]))
                    start = offsets[i + 1]
                    k += 1
            if start < b:
                out.append((f"lora{k - 1}", start, b))
        else:
            out.append((g, a, b))
    return out


class GradSync:
    """Averaging all-reduce of the flat gradient buffer in production-order buckets.

    `ready(i)` may be called from backward hooks as soon as bucket i is final (overlap on a side stream);
    `finish()` (or calling the object) reduces whatever is still pending and joins the side stream."""

    def __init__(self, gflat, buckets, group=None):
        self.gflat, self.buckets, self.group = gflat, buckets, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (dist.is_initialized() and single_rank_rehearsal())
        self.cuda = gflat.is_cuda
        self.stream = torch.cuda.Stream() if self.cuda else None
        self.done = [False] * len(buckets)
        self.works = []
        self.post_scale = (1.0 / self.world) if self.cuda else 1.0  # consumed by OptimWrapper -> vfm_adamw grad_scale
        self.measure = False      # bench.py --gpus N: HIP events around the join in finish() (the all-reduce time backward did not hide)
        self._pairs = []

    def ready(self, i):
        if not self.active or self.done[i]:
            return
        _, a, b = self.buckets[i]
        sl = self.gflat[a:b]
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group)  # the 1/world average is folded into AdamW
        else:
            self.works.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.done[i] = True

    def finish(self):
        for i in range(len(self.buckets)):
            self.ready(i)
        if self.active:
            if self.cuda:
                cur = torch.cuda.current_stream()
                if self.measure:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(cur)
                    cur.wait_stream(self.stream)
                    e1.record(cur)
                    self._pairs.append((e0, e1))
                else:
                    cur.wait_stream(self.stream)
            else:
                for w in self.works:
                    w.wait()
                self.works = []
                self.gflat.mul_(1.0 / self.world)
        self.done = [False] * len(self.buckets)

    __call__ = finish

    def exposed_ms(self):
        """Sum over the measured steps of the time the compute stream stood at the join (call after a device synchronize)."""
        ms = sum(a.elapsed_time(b) for a, b in self._pairs)
        self._pairs = []
        return ms


def bn_sync_fn(group=None):
    """In-place SUM all-reduce used by SyncBatchNorm's moment / gradient exchange (C2)."""
    def f(t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t
    return f


def attach(model, optim_wrapper, group=None):
    """Wire DP into a built model + OptimWrapper: parameter broadcast, gradient buckets, SyncBN exchange, and the
    backward-time bucket launches (heads' buckets when the backbone backward starts, the first LoRA bucket once the
    blocks it covers are done; GradSync.finish() sends the rest and joins the side stream)."""
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if world == 1 and not (dist.is_available() and dist.is_initialized() and single_rank_rehearsal()):
        return None
    broadcast_params(model)
    opt = optim_wrapper.optimizer
    buckets = make_buckets(opt.names, opt.offsets)
    gs = GradSync(opt.gflat, buckets, group)
    optim_wrapper.grad_sync = gs
    from . import backbones
    head_ids = [i for i, b in enumerate(buckets) if not b[0].startswith("lora")]
    lora_ids = [i for i, b in enumerate(buckets) if b[0].startswith("lora")]
    # block index whose completion finalises LoRA bucket k: the smallest block number among the parameters it holds
    last_block = {}
    for i in lora_ids:
        _, a, b = buckets[i]
        blks = [int(nm.split("blocks.")[1].split(".")[0]) for nm, o in zip(opt.names, opt.offsets[:-1]) if a <= o < b and "lora_" in nm]
        last_block[min(blks)] = i

    def heads_done():
        for i in head_ids:
            gs.ready(i)

    def block_done(li):
        if li in last_block:
            gs.ready(last_block[li])

    backbones.BACKWARD_EVENTS["heads_done"], backbones.BACKWARD_EVENTS["block_done"] = heads_done, block_done
    head = getattr(model, "decode_head", None)
    if head is not None and hasattr(head, "bn_sync"):
        head.bn_sync, head.bn_world = bn_sync_fn(group), world
    return gs

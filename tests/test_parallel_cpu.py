"""Data-parallel plumbing on CPU (gloo, world_size 2): bucket layout, averaging all-reduce, index sharding, the
SyncBN exchange callable, parameter broadcast.  The GPU path uses the same code with backend 'nccl' (= RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vfmseg_amd import parallel
from vfmseg_amd.optim import production_order


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _names(depth=3):
    n = []
    for i in range(depth):
        n += [f"backbone.model.base_model.model.blocks.{i}.attn.qkv.lora_A.default.weight",
              f"backbone.model.base_model.model.blocks.{i}.attn.qkv.lora_B.default.weight"]
    n += ["decode_head.conv_seg.weight", "decode_head.conv_seg.bias", "aux_decoder.conv_seg.weight",
          "aux_decoder.transformer_decoder.mask_token"]
    return n


def test_production_order_and_buckets():
    names = production_order(_names())
    assert names[0].startswith("aux_decoder") and names[2].startswith("decode_head")
    lora = [n for n in names if "lora_" in n]
    assert [int(n.split("blocks.")[1][0]) for n in lora] == [2, 2, 1, 1, 0, 0]  # last block's grads are ready first
    sizes = [7, 3, 5, 2, 11, 13, 17, 19, 23, 29]
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + s)
    b = parallel.make_buckets(names, offs, max_lora_buckets=2)
    assert b[0][0] == "aux_decoder" and b[1][0] == "decode_head" and b[2][0].startswith("lora")
    # contiguous, complete, non-overlapping
    assert b[0][1] == 0 and b[-1][2] == offs[-1]
    for x, y in zip(b[:-1], b[1:]):
        assert x[2] == y[1]


def test_shard_indices_partition():
    a, b = parallel.shard_indices(10, 0, 2, seed=3), parallel.shard_indices(10, 1, 2, seed=3)
    assert sorted(a + b) == list(range(10)) and len(a) == len(b) == 5


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init_from_env("gloo")
    torch.manual_seed(rank)
    n = 1000
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    names = production_order(_names())
    sizes = [100] * len(names)
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + s)
    gs = parallel.GradSync(g, parallel.make_buckets(names, offs), None)
    gs.ready(0)          # early bucket (overlap path)
    gs.finish()          # the rest + join; CPU path averages in place
    expect = torch.arange(n, dtype=torch.float32) * 1.5
    ok1 = torch.allclose(g, expect)
    # second step re-arms the buckets
    g.copy_(torch.ones(n) * (rank + 1))
    gs()
    ok2 = torch.allclose(g, torch.full((n,), 1.5))
    # SyncBN-style SUM exchange and parameter broadcast
    t = torch.tensor([1.0 + rank, 10.0 * (rank + 1)])
    parallel.bn_sync_fn()(t)
    ok3 = torch.allclose(t, torch.tensor([3.0, 30.0]))
    m = torch.nn.Linear(4, 4)
    parallel.broadcast_params(m)
    w = m.weight.detach().clone()
    ws = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    ok4 = torch.equal(ws[0], ws[1])
    q.put((rank, ok1, ok2, ok3, ok4))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gradsync_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=100) for _ in range(2)]
    for p in ps:
        p.join(30)
    for r in res:
        assert all(r[1:]), r

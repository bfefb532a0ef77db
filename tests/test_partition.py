"""Node-partitioned spatial encoder: partition / halo bookkeeping on the host, and the
world_size-2 exchange logic over gloo on CPU (device ops replaced by a plain-torch stand-in
that exists only here; the product path uses the HIP ops)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sgp_oracle as O
from sgp_amd import graph, partition, synthetic
from sgp_amd.sgp_preprocessing import spatial_operators


class TorchOps:
    """CPU stand-in for partition.HipOps (test infrastructure)."""

    @staticmethod
    def gather_nodes(x, index, out):
        out.copy_(x[:, index.long()])
        return out

    @staticmethod
    def propagate(op, x, y, halo):
        full = x if halo is None else torch.cat([x, halo], dim=1)
        y.copy_(torch.einsum("ij,tjf->tif", op.to_dense(), full))
        return y

    @staticmethod
    def node_sums(x):
        return x.sum(1)

    @staticmethod
    def bcast_rows(src, scale, y):
        y.copy_((src * scale)[:, None, :].expand_as(y))
        return y


def test_partition_bounds():
    assert partition.partition_bounds(10, 3) == [0, 3, 6, 10]
    rp = np.array([0, 10, 10, 10, 20, 30, 40])
    b = partition.partition_bounds(6, 2, rp)
    assert b[0] == 0 and b[-1] == 6 and rp[b[1]] >= 20 - 10


@pytest.mark.parametrize("world", [2, 3, 4])
def test_split_operator_reassembles(world):
    ei, ew, _ = synthetic.knn_graph(400, 9, seed=2)
    op = graph.ShiftOperator.from_edges(ei, ew, 400)
    bounds = partition.partition_bounds(400, world)
    dense = op.to_dense()
    blocks = [partition.split_operator(op, bounds, r) for r in range(world)]
    for r, b in enumerate(blocks):
        cols = torch.cat([torch.arange(b.lo, b.hi), b.halo_global])
        full = torch.zeros(b.n_own, 400)
        full[:, cols] = b.op.to_dense()
        assert torch.equal(full, dense[b.lo:b.hi])
        assert sum(b.recv_counts) == b.n_halo and b.recv_counts[r] == 0
        # what I receive from p is exactly what p sends to me, in the same order
        off = 0
        for p in range(world):
            n = b.recv_counts[p]
            want = b.halo_global[off:off + n]
            off += n
            so = sum(blocks[p].send_counts[:r])
            sent = blocks[p].send_index[so:so + blocks[p].send_counts[r]].long() + blocks[p].lo
            assert torch.equal(want, sent)


@pytest.mark.parametrize("world", [2, 3, 5])
def test_split_operator_gather_form_reassembles(world):
    """A graph without locality: the ranks together reference more than half of all remote rows, so the
    partitioner picks the all_gather exchange; the local operator's halo columns then address the
    gathered buffer (rank p's shard, padded to the largest, at p * gather_rows)."""
    ei, ew = synthetic.random_graph(300, 30, seed=1)
    op = graph.ShiftOperator.from_edges(ei, ew, 300)
    bounds = partition.partition_bounds(300, world, op.rowptr.numpy())
    dense = op.to_dense()
    for r in range(world):
        b = partition.split_operator(op, bounds, r, exchange="auto")
        g = b.gather_rows
        assert g == max(bounds[p + 1] - bounds[p] for p in range(world)) and b.n_halo == world * g
        d = b.op.to_dense()
        full = torch.zeros(b.n_own, 300)
        full[:, b.lo:b.hi] = d[:, :b.n_own]
        for p in range(world):
            blk = d[:, b.n_own + p * g:b.n_own + p * g + bounds[p + 1] - bounds[p]]
            if p == r:
                assert blk.abs().sum() == 0                  # own rows are read from x, not from the buffer
            else:
                full[:, bounds[p]:bounds[p + 1]] = blk
        assert torch.equal(full, dense[b.lo:b.hi])
        assert partition.split_operator(op, bounds, r, exchange="packed").gather_rows == 0
    # a graph with locality keeps the packed exchange
    ei, ew, _ = synthetic.knn_graph(400, 9, seed=2)
    knn = graph.ShiftOperator.from_edges(ei, ew, 400)
    assert partition.split_operator(knn, partition.partition_bounds(400, world), 0, exchange="auto").gather_rows == 0


def _worker(rank, world, port, cfg, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        n, t, d, k = cfg["n"], 5, 8, cfg["k"]
        ei, ew, _ = synthetic.knn_graph(n, 7, seed=4)
        if cfg.get("scramble"):                       # node numbering without locality
            ei = torch.randperm(n, generator=torch.Generator().manual_seed(3))[ei]
        if cfg.get("random"):                         # no locality at all: every rank needs nearly every row
            ei, ew = synthetic.random_graph(n, cfg["random"], seed=6)
        ops = spatial_operators(ei, ew, n, bidirectional=cfg["bidir"])
        x = torch.randn(t, n, d)
        enc, bounds = partition.make_partitioned_spatial(ops, k, cfg["glob"], ops=TorchOps,
                                                         balance=cfg.get("balance", "nnz"),
                                                         locality=cfg.get("locality", "auto"),
                                                         exchange=cfg.get("exchange", "auto"))
        if "gather" in cfg:                           # which exchange the partitioner chose
            assert all((b.gather_rows > 0) == cfg["gather"] for b in enc.blocks)
        lo, hi = bounds[rank], bounds[rank + 1]
        if "random" not in cfg:
            assert (enc.node_order is not None) == bool(cfg.get("scramble"))
        rows = torch.arange(lo, hi) if enc.node_order is None else enc.node_order[lo:hi]
        out = torch.zeros(t, hi - lo, enc.num_blocks() * d)
        out[:, :, :d] = x[:, rows]
        enc.encode_into(out, d)
        ref = O.spatial_encoder_forward(x, ei, ew, k, cfg["bidir"], False, cfg["glob"])
        ok = torch.allclose(out, ref[:, rows], rtol=1e-5, atol=1e-5)
        if cfg.get("scramble"):                       # the renumbering keeps the exchange compact
            ok = ok and enc.blocks[0].n_halo < (hi - lo)
        ret[rank] = (bool(ok), float((out - ref[:, rows]).abs().max()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cfg", [dict(n=150, k=3, bidir=True, glob=True),
                                 dict(n=97, k=2, bidir=False, glob=False, balance="rows"),
                                 dict(n=900, k=2, bidir=True, glob=True, scramble=True),
                                 # SURVEY 8e's general case: halo ~ everything -> all_gather of full shards
                                 dict(n=301, k=2, bidir=True, glob=True, random=40, locality="never", gather=True),
                                 dict(n=150, k=2, bidir=False, glob=False, exchange="gather", gather=True),
                                 dict(n=301, k=2, bidir=False, glob=True, random=40, locality="never", gather=True,
                                      world=3)])
def test_two_rank_gloo_matches_single_process(cfg):
    world = cfg.get("world", 2)
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, cfg, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        assert ret[r][0], f"rank {r}: max err {ret[r][1]}"


def test_permute_operator_is_a_relabelling():
    ei, ew, _ = synthetic.knn_graph(120, 6, seed=8)
    op = graph.ShiftOperator.from_edges(ei, ew, 120)
    order = torch.randperm(120, generator=torch.Generator().manual_seed(1))
    p = partition.permute_operator(op, order)
    assert torch.equal(p.to_dense(), op.to_dense()[order][:, order])
    rp = p.rowptr.long()
    for r in range(120):                               # columns sorted inside every row
        c = p.col[rp[r]:rp[r + 1]]
        assert torch.all(c[1:] > c[:-1])


def test_locality_renumbering_shrinks_the_halo_of_a_scrambled_graph():
    n, world = 4000, 8
    ei, ew, _ = synthetic.knn_graph(n, 12, seed=6)
    ei = torch.randperm(n, generator=torch.Generator().manual_seed(2))[ei]
    ops = spatial_operators(ei, ew, n)
    plain = partition.halo_rows(ops[0], partition.partition_bounds(n, world), world // 2)
    enc, bounds = partition.make_partitioned_spatial(ops, 2, False, rank=world // 2, world_size=world,
                                                     ops=TorchOps)
    assert enc.node_order is not None and sorted(enc.node_order.tolist()) == list(range(n))
    own = bounds[world // 2 + 1] - bounds[world // 2]
    assert plain > 4 * own                              # near-full exchange without it
    assert enc.blocks[0].n_halo < own                   # boundary strip with it
    # a graph whose numbering already has locality is left alone
    ei2, ew2, _ = synthetic.knn_graph(n, 12, seed=6)
    enc2, _ = partition.make_partitioned_spatial(spatial_operators(ei2, ew2, n), 2, False,
                                                 rank=1, world_size=world, ops=TorchOps)
    assert enc2.node_order is None

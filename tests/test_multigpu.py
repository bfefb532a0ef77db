"""Drop-in multi-GPU entry (``sgp_amd/multigpu.py``): ``SGPEncoder.forward(..., gpus=N)`` and
``encode_dataset(..., gpus=N)`` -- the reference's single-process call (lib/utils.py:27-31) with the graph
node-partitioned over N ranks behind it.  On the one-GPU test box the ranks share the device over gloo:
the results must equal the single-GPU call."""
import os

import pytest
import torch

import sgp_amd
from sgp_amd import multigpu, synthetic


# ------------------------------------------------------------------ host logic (CPU)
def test_resolve_gpus(monkeypatch):
    monkeypatch.delenv("SGP_AMD_GPUS", raising=False)
    assert multigpu.resolve_gpus(None) == 1
    assert multigpu.resolve_gpus(4) == 4 and multigpu.resolve_gpus("2") == 2
    monkeypatch.setenv("SGP_AMD_GPUS", "8")
    assert multigpu.resolve_gpus(None) == 8
    assert multigpu.resolve_gpus(1) == 1                        # the argument wins over the environment
    assert multigpu.resolve_gpus("all") == max(1, torch.cuda.device_count())
    with pytest.raises(ValueError):
        multigpu.resolve_gpus(-1)


def test_rank_rows_and_chunk_steps():
    rows, n = multigpu.rank_rows([0, 5, 9], None, 1)
    assert rows == slice(5, 9) and n == 4
    order = torch.tensor([3, 1, 4, 0, 2, 8, 7, 6, 5])
    rows, n = multigpu.rank_rows([0, 5, 9], order, 1)
    assert n == 4 and rows.tolist() == [8, 7, 6, 5]
    assert multigpu.chunk_steps(1000, 100, 3, 320, 10 ** 9) == 1000
    assert multigpu.chunk_steps(1000, 100000, 64, 320, 2 * 10 ** 9) == 8          # the floor
    tc = multigpu.chunk_steps(1000, 10000, 64, 320, 10 ** 9)
    assert 8 <= tc < 1000 and 4 * tc * 10000 * 384 * 4 <= 10 ** 9       # two input + two embedding slots


def test_shm_space_is_checked_before_anything_is_mapped(tmp_path, monkeypatch):
    """First-contact hardening (round-5 review): input + result must fit the shared filesystem BEFORE the mapping is
    made -- tmpfs over-commit is a SIGBUS in the middle of a run otherwise; the error names the bytes."""
    free = multigpu.require_shm_space(str(tmp_path), 1024, "a probe")
    assert free >= 1024
    with pytest.raises(RuntimeError, match=r"needs .* GiB \(\d+ bytes\).*shard_dir="):
        multigpu.require_shm_space(str(tmp_path), free + (1 << 40), "the shared result")
    # shared_result refuses a shape the directory cannot hold, and leaves no file behind
    huge = (1 << 20, 1 << 12, 1 << 10)                                         # 16 PiB of float32
    with pytest.raises(RuntimeError, match="shared result"):
        multigpu.shared_result(huge, str(tmp_path))
    assert list(tmp_path.iterdir()) == []
    t, path = multigpu.shared_result((3, 4, 5), str(tmp_path))
    assert tuple(t.shape) == (3, 4, 5) and os.path.exists(path)
    os.unlink(path)
    t.fill_(2.0)                                                               # the mapping outlives the name
    assert float(t.sum()) == 120.0
    monkeypatch.setenv("SGP_AMD_DIST_TIMEOUT", "77")
    assert multigpu.dist_timeout().total_seconds() == 77
    monkeypatch.delenv("SGP_AMD_DIST_TIMEOUT")
    assert multigpu.dist_timeout().total_seconds() == 1800


def test_partition_plan_cut_once_equals_every_ranks_own_cut():
    """``plan_partition`` (one process cuts all ranks' blocks: what ``encode_multi_gpu`` does in the parent) gives
    every rank exactly the blocks ``make_partitioned_spatial`` computes for itself -- also under a locality
    renumbering -- and the blocks survive the trip through ``torch.save``."""
    import io
    import warnings
    from sgp_amd import partition
    from sgp_amd.sgp_preprocessing import spatial_operators
    for scramble in (False, True):
        n, world = 900, 3
        ei, ew, _ = synthetic.knn_graph(n, 12, seed=4)
        if scramble:
            ei = torch.randperm(n, generator=torch.Generator().manual_seed(0))[ei]
        ops = spatial_operators(ei, ew, n, bidirectional=True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            plan = partition.plan_partition(ops, world)
            assert (plan.node_order is not None) == scramble
            buf = io.BytesIO()
            torch.save(plan.rank_blocks[1], buf)
            buf.seek(0)
            loaded = torch.load(buf, weights_only=False)
            for r in range(world):
                sp, bounds = partition.make_partitioned_spatial(ops, 2, False, rank=r, world_size=world,
                                                                ops=partition.HipOps)
                assert [int(b) for b in bounds] == plan.bounds
                for mine, theirs in zip(plan.rank_blocks[r] if r != 1 else loaded, sp.blocks):
                    assert (mine.lo, mine.hi, mine.gather_rows) == (theirs.lo, theirs.hi, theirs.gather_rows)
                    assert torch.equal(mine.op.rowptr, theirs.op.rowptr) and torch.equal(mine.op.col, theirs.op.col)
                    assert torch.equal(mine.op.val, theirs.op.val) and mine.op.num_cols == theirs.op.num_cols
                    assert torch.equal(mine.halo_global, theirs.halo_global)
                    assert torch.equal(mine.send_index, theirs.send_index)
                    assert mine.send_counts == theirs.send_counts and mine.recv_counts == theirs.recv_counts
                assert sp.norm_inf == plan.norm_inf


@pytest.mark.gpu
def test_rank_pipeline_overlaps_transfers_with_the_next_chunk():
    """The per-rank time-chunk pipeline: the D2H of chunk i runs while the device already encodes chunk i + 1 (event
    times), the host gather / scatter never sits between two chunks' compute, and the data arrive intact."""
    import time
    dev = torch.device("cuda", 0)
    T, tc, n_own, f_in, d_out = 48, 8, 4000, 16, 1024
    x = torch.randn(T, 2 * n_own, f_in)
    rows = torch.arange(0, 2 * n_own, 2)
    got = torch.zeros(T, n_own, d_out)

    def encode(xs, oc):                                            # ~15 ms of device work per chunk, only enqueued
        torch.cuda._sleep(30_000_000)
        oc.copy_(xs.repeat(1, 1, d_out // f_in))

    def sink(t0, n, emb):
        got[t0:t0 + n] = emb

    pipe = multigpu.RankPipeline(dev, tc, n_own, f_in, d_out)
    events = []
    t0 = time.perf_counter()
    pipe.run(x, rows, T, encode, sink, events)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    assert torch.equal(got, x[:, rows].repeat(1, 1, d_out // f_in))
    comp = [a.elapsed_time(b) for a, b, _ in events]
    d2h_after_next_start = [events[i + 1][0].elapsed_time(events[i][2]) for i in range(len(events) - 1)]
    gaps = [events[i][1].elapsed_time(events[i + 1][0]) for i in range(len(events) - 1)]
    assert min(d2h_after_next_start) > 0, d2h_after_next_start      # chunk i leaves while chunk i + 1 is being encoded
    assert max(gaps[1:]) < 0.5 * min(comp), (gaps, comp)            # nothing (host copies, D2H) between two chunks' compute
    assert wall * 1e3 < sum(comp) + 3 * max(comp), (wall, comp)


def test_gpus_argument_is_rejected_where_it_cannot_be_served():
    enc = sgp_amd.SGPEncoder(input_size=3, reservoir_size=16, reservoir_layers=1, leaking_rate=.9,
                             spectral_radius=.9, density=.7, input_scaling=1., receptive_field=1,
                             bidirectional=False, alpha_decay=False, global_attr=False)
    ei = torch.tensor([[0, 1], [1, 0]])
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):                       # no GPU: the product path fails loudly
            enc(torch.randn(4, 2, 3), ei, None, gpus=2)
    with pytest.raises(ValueError):
        enc(torch.randn(4, 2, 3), ei, None, gpus=2, return_device=True)


def test_sharded_embedding_reads_time_and_node_shards_back(tmp_path):
    """ShardedEmbedding over shard files as the one-GPU path (time shards of all nodes) and the ranks of the
    partitioned path (time x node-block shards, rows in any order) write them: every time range comes back in the
    original node order."""
    from sgp_amd.datasets import ShardedEmbedding
    t, n, d = 23, 17, 5
    full = torch.randn(t, n, d)
    paths = []
    for t0 in range(0, t, 8):                                   # time shards
        p = str(tmp_path / f"a_{t0}.pt")
        torch.save(dict(t0=t0, steps=min(8, t - t0), rows=None, embedding=full[t0:t0 + 8].clone()), p)
        paths.append(p)
    ShardedEmbedding.write_index(str(tmp_path), paths, (t, n, d))
    emb = ShardedEmbedding.from_dir(str(tmp_path))
    assert emb.shape == (t, n, d) and len(emb) == t
    assert torch.equal(emb.load_steps(0, t), full) and torch.equal(emb.load_steps(5, 19), full[5:19])
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(0))
    paths = []
    for r, rows in enumerate((perm[:9], perm[9:])):             # two ranks, scattered node blocks, two time chunks
        for t0 in (0, 12):
            p = str(tmp_path / f"b_{r}_{t0}.pt")
            steps = 12 if t0 == 0 else t - 12
            torch.save(dict(t0=t0, steps=steps, rank=r, rows=rows, embedding=full[t0:t0 + steps][:, rows].clone()), p)
            paths.append(p)
    emb = ShardedEmbedding(paths, t, n, d)
    assert torch.equal(emb.load_steps(0, t), full) and torch.equal(emb.load_steps(11, 13), full[11:13])


def test_shard_steps_needs_a_directory_and_a_host_embedding():
    from test_host_logic import FakeDataset, StubEncoder
    ds = FakeDataset(torch.randn(6, 4, 1), torch.randn(6, 2), torch.tensor([[0, 1], [1, 2]]), None)
    with pytest.raises(ValueError):
        sgp_amd.encode_dataset(ds, StubEncoder, dict(input_size=3), shard_steps=4)                 # no save_path
    with pytest.raises(ValueError):
        sgp_amd.encode_dataset(ds, StubEncoder, dict(input_size=3), shard_steps=4, save_path="x", return_device=True)


# ------------------------------------------------------------------ GPU
def _encoder(seed=3, **kw):
    torch.manual_seed(seed)
    args = dict(input_size=3, reservoir_size=32, reservoir_layers=2, leaking_rate=.9, spectral_radius=.9,
                density=.7, input_scaling=1., receptive_field=3, bidirectional=True, alpha_decay=True,
                global_attr=True)
    args.update(kw)
    return sgp_amd.SGPEncoder(**args)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_forward_gpus_equals_single_gpu(world):
    """2 and 4 ranks (sharing the GPU over gloo on a one-GPU box) == the single-GPU forward, in the
    ORIGINAL node order, through the same call."""
    n, t = 3000, 40
    ei, ew, _ = synthetic.knn_graph(n, 30, seed=5)
    enc = _encoder()
    x = torch.randn(t, n, 3, generator=torch.Generator().manual_seed(1))
    ref = enc(x, ei, ew)
    info = {}
    y = multigpu.encode_multi_gpu(enc, x, ei, ew, world, info=info)
    assert y.shape == ref.shape and not y.is_cuda
    assert info["world"] == world and len(info["bounds"]) == world + 1 and info["halo_rows"] > 0
    assert torch.allclose(y, ref, rtol=1e-6, atol=1e-6), float((y - ref).abs().max())
    y2 = enc(x, ei, ew, gpus=world)                              # the drop-in spelling
    assert torch.equal(y2, y)


@pytest.mark.gpu
def test_forward_gpus_renumbers_graphs_without_locality_and_chunks_time():
    """Scrambled node labels: the ranks own a locality renumbering, the caller still gets its own node order;
    a small device budget cuts the time axis (state carried on the devices)."""
    n, t = 2400, 50
    ei, ew, _ = synthetic.knn_graph(n, 12, seed=2)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1))
    ei = perm[ei]
    enc = _encoder(seed=8, reservoir_layers=1, bidirectional=False, alpha_decay=False)
    x = torch.randn(t, n, 3, generator=torch.Generator().manual_seed(2))
    ref = enc(x, ei, ew)
    info = {}
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y = multigpu.encode_multi_gpu(enc, x, ei, ew, 3, info=info, device_budget_bytes=6 * 2 ** 20)
    assert info["reordered"] and info["t_chunk"] < t
    assert torch.allclose(y, ref, rtol=1e-6, atol=1e-6), float((y - ref).abs().max())


@pytest.mark.gpu
def test_encode_dataset_gpus_on_the_reference_harness_fixture():
    """``encode_dataset(..., gpus=2)`` on a g5 fixture (the reference's own harness output)."""
    import numpy as np
    from conftest import GOLDEN, golden_files
    from test_host_logic import FakeDataset
    name = golden_files("g5_")[0]
    z = np.load(os.path.join(GOLDEN, name))
    ds = FakeDataset(torch.from_numpy(z["data"]), torch.from_numpy(z["u"]),
                     torch.from_numpy(z["edge_index"]), torch.from_numpy(z["edge_weight"]))
    enc_exo = bool(z["encode_exogenous"])
    kw = dict(input_size=3 if enc_exo else 1, reservoir_size=16, reservoir_layers=1, leaking_rate=.9,
              spectral_radius=.9, density=.7, input_scaling=1., receptive_field=2, bidirectional=False,
              alpha_decay=False, global_attr=False, add_self_loops=False, undirected=False)
    torch.manual_seed(int(z["seed"]))
    sgp_amd.encode_dataset(ds, sgp_amd.SGPEncoder, kw, encode_exogenous=enc_exo, keep_raw=bool(z["keep_raw"]),
                           gpus=2)
    got, want = ds._t["encoded_x"], torch.from_numpy(z["encoded_x"])
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5), float((got - want).abs().max())


@pytest.mark.gpu
def test_shard_dir_writes_rank_shards_instead_of_a_host_tensor(tmp_path):
    n, t = 2200, 30
    ei, ew, _ = synthetic.knn_graph(n, 20, seed=4)
    enc = _encoder(seed=5, reservoir_layers=1)
    x = torch.randn(t, n, 3, generator=torch.Generator().manual_seed(3))
    ref = enc(x, ei, ew)
    paths = multigpu.encode_multi_gpu(enc, x, ei, ew, 2, shard_dir=str(tmp_path), device_budget_bytes=8 * 2 ** 20)
    assert len(paths) >= 4
    got = torch.full_like(ref, float("nan"))
    for p in paths:
        s = torch.load(p)
        got[s["t0"]:s["t0"] + s["steps"]].index_copy_(1, s["rows"], s["embedding"])
    assert torch.allclose(got, ref, rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------ shards on disk, node-sharded sampling
@pytest.mark.gpu
def test_encode_dataset_streams_time_shards_to_disk(tmp_path):
    """``encode_dataset(save_path=dir, shard_steps=S)``: time shards on disk + encoder description, equal to
    slices of the one-tensor embedding; ``encoded_x`` is a ShardedEmbedding that reads any range back."""
    from sgp_amd.datasets import ShardedEmbedding
    from test_host_logic import FakeDataset
    n, t = 700, 50
    ei, ew, _ = synthetic.knn_graph(n, 10, seed=1)
    ds = FakeDataset(torch.randn(t, n, 1), torch.randn(t, 2), ei, ew)
    kw = dict(input_size=3, reservoir_size=16, reservoir_layers=1, leaking_rate=.9, spectral_radius=.9,
              density=.7, input_scaling=1., receptive_field=2, bidirectional=False, alpha_decay=False,
              global_attr=True)
    torch.manual_seed(5)
    sgp_amd.encode_dataset(ds, sgp_amd.SGPEncoder, kw, save_path=str(tmp_path / "emb"), shard_steps=16)
    sharded = ds._t["encoded_x"]
    assert isinstance(sharded, ShardedEmbedding) and sharded.shape == (t, n, 16 * 4) and len(sharded.paths) == 4
    desc = torch.load(tmp_path / "emb" / "encoder.pt")
    twin = sgp_amd.SGPEncoder(**desc["kwargs"]); twin.load_state_dict(desc["state_dict"])
    x, _ = ds.get_tensors(["data", "u"], preprocess=True, cat_dim=-1)
    full = twin(x, ei, ew)
    assert torch.equal(sharded.load_steps(0, t), full)
    assert torch.equal(sharded.load_steps(10, 37), full[10:37])
    again = ShardedEmbedding.from_dir(str(tmp_path / "emb"))
    assert again.shape == sharded.shape and torch.equal(again.load_steps(40, 50), full[40:])


@pytest.mark.gpu
def test_sharded_embedding_from_rank_shards(tmp_path):
    n, t = 2200, 24
    ei, ew, _ = synthetic.knn_graph(n, 20, seed=4)
    enc = _encoder(seed=6, reservoir_layers=1)
    x = torch.randn(t, n, 3, generator=torch.Generator().manual_seed(3))
    ref = enc(x, ei, ew)
    sharded = enc(x, ei, ew, gpus=2, shard_dir=str(tmp_path))
    assert sharded.shape == tuple(ref.shape)
    got = sharded.load_steps(0, t)
    assert torch.allclose(got, ref, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("cut", [2, 3])
def test_sharded_iid_sampler_equals_the_reference_fixture(cut):
    """g7 fixtures (the reference's own ``IIDDataset.sample`` output) through node shards == g7: the index
    sequence is drawn once as the reference draws it, every shard gathers its own rows on the device."""
    import numpy as np
    from conftest import GOLDEN, golden_files
    from sgp_amd.datasets import ShardedIIDSampler
    for name in golden_files("g7_iid_"):
        z = np.load(os.path.join(GOLDEN, name))
        hz, delay, lag, nb = [int(v) for v in z["cfg"]]
        emb, y, u = (torch.from_numpy(z[k]) for k in ("emb", "y", "u"))
        n = emb.shape[1]
        perm = torch.randperm(n, generator=torch.Generator().manual_seed(cut))     # shards own scattered nodes
        ids = [perm[i::cut] for i in range(cut)]
        s = ShardedIIDSampler(emb.shape[0], n, hz, delay, lag)
        s.add_input_shards("x", [(i, emb[:, i].contiguous().cuda()) for i in ids])
        if bool(z["has_exo"]):
            s.add_input("u", u, "t f", preprocess=False)
        s.add_target_shards("y", [(i, y[:, i].contiguous().cuda()) for i in ids])
        torch.manual_seed(int(z["seed"]))
        out = s.sample(nb)
        assert np.array_equal(out["input"]["node_index"].numpy(), z["out_node_index"])
        assert np.array_equal(out["input"]["x"].cpu().numpy(), z["out_x"])
        if not bool(z["has_scaler"]):
            assert np.array_equal(out["target"]["y"].cpu().numpy(), z["out_y"])
        if bool(z["has_exo"]):
            assert np.array_equal(out["input"]["u"].cpu().numpy(), z["out_u"])

"""Synthetic workloads of SURVEY.md 8d / BASELINE.md 3 (there are no dataset files):
a geometric k-NN graph with Gaussian-kernel weights in Morton node order (mirrors
``experiments/run_largescale_sgp.py:167-170`` + ``tsl/ops/similarities.py:58-62,104-122``)
and an adversarial uniformly random graph.  Both return the reference's ``edge_index``
convention: row 0 = source j, row 1 = target i of the entry A[i, j]."""
import numpy as np
import torch


def morton_order(xy, bits=16):
    q = np.minimum((xy * (1 << bits)).astype(np.uint64), (1 << bits) - 1)

    def spread(v):
        v = v & np.uint64(0xFFFF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF)
        v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F)
        v = (v | (v << np.uint64(2))) & np.uint64(0x33333333)
        v = (v | (v << np.uint64(1))) & np.uint64(0x55555555)
        return v
    code = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1))
    return np.argsort(code, kind="stable")


def knn_graph(n, k=100, seed=1):
    """k nearest neighbours (self excluded) of n uniform points in the unit square; weights
    exp(-(d/theta)^2), theta = std of pairwise distances of a 4096-point sample; nodes in
    Morton order.  Exactly k in-edges per node, directed, weighted."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    xy = rng.random((n, 2))
    xy = xy[morton_order(xy)]
    k = min(k, n - 1)
    tree = cKDTree(xy)
    dist, idx = tree.query(xy, k=k + 1, workers=-1)
    dist, idx = dist[:, 1:], idx[:, 1:]                       # drop self
    m = min(n, 4096)
    s = xy[rng.choice(n, m, replace=False)]
    d = np.sqrt(((s[:, None, :] - s[None, :, :]) ** 2).sum(-1))
    theta = d.std()
    w = np.exp(-(dist / theta) ** 2).astype(np.float32)
    target = np.repeat(np.arange(n, dtype=np.int64), k)
    source = idx.reshape(-1).astype(np.int64)
    edge_index = torch.from_numpy(np.stack([source, target]))
    return edge_index, torch.from_numpy(w.reshape(-1)), torch.from_numpy(xy.astype(np.float32))


def threshold_graph(n, mean_degree, seed=1):
    """The reference's FULL large-scale graphs (``adj_knn=None``: experiments/run_largescale_sgp.py:167-170 with
    config/largescale/sgp_pv.yaml / sgp_cer.yaml): Gaussian-kernel similarities of ALL pairs with the small ones
    cut off (tsl/ops/similarities.py:58-62 + the connectivity threshold) -- PV-US 3 710 008 edges on 5 016 nodes (~740
    per row), CER-En 3 186 369 on 6 435 (~495).  Synthetic stand-in: n uniform points in the unit square in Morton
    order, weights exp(-(d/theta)^2), theta = std of the pairwise distances, every pair within the radius that
    gives ``mean_degree`` entries per row on average (self excluded); rows near the border are shorter, rows in
    the middle longer -- ragged lengths like the real graphs."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    xy = rng.random((n, 2))
    xy = xy[morton_order(xy)]
    m = min(n, 4096)
    smp = xy[rng.choice(n, m, replace=False)]
    d = np.sqrt(((smp[:, None, :] - smp[None, :, :]) ** 2).sum(-1))
    theta = d.std()
    radius = float(np.quantile(d[np.triu_indices(m, 1)], min(1.0, mean_degree / (n - 1))))
    tree = cKDTree(xy)
    pairs = tree.query_pairs(radius, output_type="ndarray")               # i < j
    src = np.concatenate([pairs[:, 0], pairs[:, 1]]).astype(np.int64)
    dst = np.concatenate([pairs[:, 1], pairs[:, 0]]).astype(np.int64)
    dist = np.sqrt(((xy[src] - xy[dst]) ** 2).sum(-1))
    w = np.exp(-(dist / theta) ** 2).astype(np.float32)
    return torch.from_numpy(np.stack([src, dst])), torch.from_numpy(w), torch.from_numpy(xy.astype(np.float32))


def random_graph(n, k=100, seed=1):
    """k distinct uniformly random in-neighbours per node, weights U(0, 1)."""
    rng = np.random.default_rng(seed)
    k = min(k, n)
    src = np.empty((n, k), dtype=np.int64)
    if n <= 4 * k:
        for i in range(n):
            src[i] = rng.choice(n, k, replace=False)
    else:
        src = rng.integers(0, n, (n, k))
        for _ in range(8):                                    # re-draw duplicates
            s = np.sort(src, axis=1)
            dup = np.zeros_like(src, dtype=bool)
            dup[:, 1:] = s[:, 1:] == s[:, :-1]
            if not dup.any():
                src = s
                break
            s[dup] = rng.integers(0, n, int(dup.sum()))
            src = s
    target = np.repeat(np.arange(n, dtype=np.int64), k)
    w = rng.random(n * k).astype(np.float32)
    return torch.from_numpy(np.stack([src.reshape(-1), target])), torch.from_numpy(w)


def sparse_traffic_graph(n, e, seed=1):
    """METR-LA / PEMS-BAY shaped: e random directed weighted edges, no self loops."""
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, e)
    dst = (src + 1 + rng.integers(0, n - 1, e)) % n
    w = (rng.random(e) * 0.9 + 0.1).astype(np.float32)
    return torch.from_numpy(np.stack([src, dst]).astype(np.int64)), torch.from_numpy(w)

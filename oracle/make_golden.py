"""Generate ``tests/golden/*.npz`` by running the UNMODIFIED reference files.

Container-only (needs ``/root/reference``); run as ``python oracle/make_golden.py``.
Each fixture stores inputs, explicit weights, flags and the reference's fp32
output (plus, where small enough, the same reference module evaluated in fp64
for tolerance context).  Fixtures are data only -- no reference source text.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def layer_dump(reservoir, prefix=""):
    layers = reservoir.reservoir_layers if hasattr(reservoir, "reservoir_layers") \
        else reservoir.rnn_cells
    d = {}
    for i, l in enumerate(layers):
        d[f"{prefix}w_ih_{i}"] = l.w_ih.detach().numpy().copy()
        d[f"{prefix}w_hh_{i}"] = l.w_hh.detach().numpy().copy()
        d[f"{prefix}b_ih_{i}"] = l.b_ih.detach().numpy().copy()
        d[f"{prefix}alpha_{i}"] = np.float64(l.alpha)
    d[f"{prefix}num_layers"] = np.int64(len(layers))
    return d


def graph(n, e, seed, isolated=True):
    """Random weighted digraph with duplicate edges, self loops and (optionally)
    one isolated node (no in- or out-edges) -- SURVEY.md 8c, fixture G1."""
    g = torch.Generator().manual_seed(seed)
    hi = n - 1 if isolated else n
    ei = torch.randint(0, hi, (2, e), generator=g)
    ei[:, 1] = ei[:, 0]                       # exact duplicate edge
    ei[0, 2] = ei[1, 2]                       # self loop
    ei[0, 3] = ei[1, 3] = 5                   # another self loop
    ew = torch.rand(e, generator=g) + 0.05
    return ei, ew


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    ref = ref_shim.load_reference()
    g = torch.Generator().manual_seed(1234)

    # ---- G0: reservoir only -------------------------------------------
    g0 = [
        (3, 64, 1, .9, .9, .7, False, "tanh"),
        (3, 64, 2, .9, .9, .7, True, "tanh"),
        (3, 16, 8, 1.0, .99, .7, True, "tanh"),
        (3, 128, 1, .8, .9, .7, True, "tanh"),
        (4, 32, 2, .9, .9, 1.0, False, "relu"),
        (4, 32, 2, .9, .9, 1.0, False, "self_norm"),
        (64, 64, 1, .9, .9, .9, False, "tanh"),
        (5, 24, 3, .7, .8, .5, True, "tanh"),
    ]
    T, N = 40, 12
    for idx, (f, r, L, a, rho, dens, dec, act) in enumerate(g0):
        torch.manual_seed(100 + idx)
        res = ref.Reservoir(input_size=f, hidden_size=r, num_layers=L,
                            leaking_rate=a, spectral_radius=rho, density=dens,
                            activation=act, alpha_decay=dec)
        x = torch.randn(T, N, f, generator=g)
        y = res(x[None])[0]
        y_last = res(x[None], return_last_state=True)[0]
        y64 = res.double()(x[None].double())[0]
        res.float()
        save(f"g0_reservoir_{idx}", x=x.numpy(), y=y.numpy(),
             y_last=y_last.numpy(), y64=y64.numpy(), activation=np.array(act),
             cfg=np.array([f, r, L, a, rho, dens, float(dec)]),
             **layer_dump(res))

    # ---- G1: spatial only ---------------------------------------------
    n, e = 50, 300
    ei, ew = graph(n, e, seed=7)
    x = torch.randn(6, n, 8, generator=g)
    flagsets = {
        "plain": dict(),
        "bidir": dict(bidirectional=True),
        "selfloops": dict(add_self_loops=True),
        "bidir_selfloops": dict(bidirectional=True, add_self_loops=True),
        "undirected": dict(undirected=True),
        "undirected_selfloops": dict(undirected=True, add_self_loops=True),
        "global": dict(global_attr=True),
        "bidir_global": dict(bidirectional=True, global_attr=True),
    }
    for fname, fl in flagsets.items():
        for k in (0, 1, 2, 4):
            enc = ref.SGPSpatialEncoder(
                receptive_field=k, bidirectional=fl.get("bidirectional", False),
                undirected=fl.get("undirected", False),
                global_attr=fl.get("global_attr", False),
                add_self_loops=fl.get("add_self_loops", False))
            y = enc(x, ei, ew)
            y64 = enc(x.double(), ei, ew.double())
            save(f"g1_spatial_{fname}_k{k}", x=x.numpy(), edge_index=ei.numpy(),
                 edge_weight=ew.numpy(), y=y.numpy(), y64=y64.numpy(), k=np.int64(k),
                 bidirectional=np.bool_(fl.get("bidirectional", False)),
                 undirected=np.bool_(fl.get("undirected", False)),
                 global_attr=np.bool_(fl.get("global_attr", False)),
                 add_self_loops=np.bool_(fl.get("add_self_loops", False)))
    # unit weights (edge_weight=None) and remove_self_loops via the function API
    outs = ref.sgp_spatial_embedding(x, n, ei, None, k=2)
    save("g1_embedding_noweight", x=x.numpy(), edge_index=ei.numpy(),
         y=torch.cat(outs, -1).numpy(), k=np.int64(2))
    outs = ref.sgp_spatial_embedding(x, n, ei, ew, k=2, remove_self_loops=True,
                                     bidirectional=True)
    save("g1_embedding_removeloops", x=x.numpy(), edge_index=ei.numpy(),
         edge_weight=ew.numpy(), y=torch.cat(outs, -1).numpy(), k=np.int64(2))

    # ---- G2: full SGPEncoder, shipped flag sets -----------------------
    shipped = {
        # config/traffic/sgp_la.yaml:7-21
        "la": dict(reservoir_size=64, reservoir_layers=2, leaking_rate=.9,
                   spectral_radius=.9, density=.7, alpha_decay=True,
                   bidirectional=True, receptive_field=4, undirected=False,
                   add_self_loops=False, global_attr=True),
        # config/traffic/sgp_bay.yaml:7-21
        "bay": dict(reservoir_size=128, reservoir_layers=1, leaking_rate=.8,
                    spectral_radius=.9, density=.7, alpha_decay=True,
                    bidirectional=True, receptive_field=4, undirected=False,
                    add_self_loops=False, global_attr=True),
        # config/largescale_100nn/sgp_pv.yaml:10-24
        "pv": dict(reservoir_size=16, reservoir_layers=8, leaking_rate=1.,
                   spectral_radius=.99, density=.7, alpha_decay=True,
                   bidirectional=False, receptive_field=2, undirected=False,
                   add_self_loops=False, global_attr=True),
        # BASELINE.json configs[0] (METR-LA, K=2, reservoir=64)
        "c1": dict(reservoir_size=64, reservoir_layers=1, leaking_rate=.9,
                   spectral_radius=.9, density=.7, alpha_decay=False,
                   bidirectional=False, receptive_field=2, undirected=False,
                   add_self_loops=False, global_attr=False),
        # ablations: config/traffic/sgp_la_abl1.yaml (K=0), K=1
        "abl_k0": dict(reservoir_size=32, reservoir_layers=2, leaking_rate=.9,
                       spectral_radius=.9, density=.7, alpha_decay=True,
                       bidirectional=True, receptive_field=0, undirected=False,
                       add_self_loops=False, global_attr=True),
        "abl_k1": dict(reservoir_size=32, reservoir_layers=1, leaking_rate=.9,
                       spectral_radius=.9, density=.7, alpha_decay=False,
                       bidirectional=False, receptive_field=1, undirected=True,
                       add_self_loops=True, global_attr=False),
    }
    n, e, T = 16, 80, 20
    ei, ew = graph(n, e, seed=11)
    for idx, (name, kw) in enumerate(shipped.items()):
        torch.manual_seed(200 + idx)
        enc = ref.SGPEncoder(input_size=3, input_scaling=1., **kw)
        x = torch.randn(T, n, 3, generator=g)
        y = enc(x, ei, ew)
        flags = {k: np.array(v) for k, v in kw.items()}
        save(f"g2_encoder_{name}", x=x.numpy(), edge_index=ei.numpy(),
             edge_weight=ew.numpy(), y=y.numpy(), **flags,
             **layer_dump(enc.reservoir))

    # ---- G3: GESNEncoder ----------------------------------------------
    torch.manual_seed(300)
    enc = ref.GESNEncoder(input_size=3, reservoir_size=32, reservoir_layers=3,
                          leaking_rate=.9, spectral_radius=.9, density=1.,
                          input_scaling=1., alpha_decay=True)
    x = torch.randn(T, n, 3, generator=g)
    y = enc(x, ei, ew)
    save("g3_gesn", x=x.numpy(), edge_index=ei.numpy(), edge_weight=ew.numpy(),
         y=y.numpy(), **layer_dump(enc.reservoir))

    # ---- G4: seed -> weights (RNG order) ------------------------------
    for idx, (seed, f, r, L, dens, scale) in enumerate(
            [(42, 3, 64, 2, .7, 1.), (7, 5, 16, 3, 1.0, 1.5), (0, 64, 32, 1, .9, 1.)]):
        torch.manual_seed(seed)
        res = ref.Reservoir(input_size=f, hidden_size=r, num_layers=L,
                            leaking_rate=.9, spectral_radius=.95, density=dens,
                            input_scaling=scale, alpha_decay=True)
        after = torch.rand(4)                   # RNG state after construction
        save(f"g4_seed_{idx}", seed=np.int64(seed),
             cfg=np.array([f, r, L, .9, .95, dens, scale]),
             rng_after=after.numpy(), **layer_dump(res))
    torch.manual_seed(5)
    ges = ref.GraphESN(input_size=3, hidden_size=16, num_layers=2, density=.8,
                       alpha_decay=True)
    save("g4_seed_gesn", seed=np.int64(5), rng_after=torch.rand(4).numpy(),
         **layer_dump(ges))

    # ---- G6: sgp_spatial_support (explicit supports, incl. its quirks) --
    n6, e6 = 30, 140
    ei6, ew6 = graph(n6, e6, seed=21)
    cases = {
        "plain_k1": dict(k=1), "plain_k3": dict(k=3),
        "selfloops_k2": dict(k=2, add_self_loops=True),
        "removeloops_k2": dict(k=2, remove_self_loops=True),
        "undirected_k2": dict(k=2, undirected=True),
        "bidir_k2": dict(k=2, bidirectional=True),
        "bidir_global_k3": dict(k=3, bidirectional=True, global_attr=True),
        "noweight_k2": dict(k=2),
    }
    for name, kw in cases.items():
        w6 = None if name.startswith("noweight") else ew6
        sup = ref.sgp_spatial_support(ei6, w6, num_nodes=n6, **kw)
        dense = [s_.numpy() if torch.is_tensor(s_) else s_.to_dense().numpy() for s_ in sup]
        save(f"g6_support_{name}", edge_index=ei6.numpy(),
             edge_weight=(ew6.numpy() if w6 is not None else np.zeros(0, np.float32)),
             has_weight=np.bool_(w6 is not None), supports=np.stack(dense).astype(np.float32),
             n=np.int64(n6), **{k_: np.array(v_) for k_, v_ in kw.items()})

    # ---- G5: encode_dataset harness -----------------------------------
    class FakeDataset:
        def __init__(self, data, u, ei, ew):
            self._t = {"data": data, "u": u}
            self.exogenous = {"u": u}
            self.edge_index, self.edge_weight = ei, ew
            self.calls = []

        def get_tensors(self, keys, preprocess=False, cat_dim=None):
            self.calls.append(("get_tensors", list(keys), preprocess, cat_dim))
            ts = [self._t[k] if self._t[k].dim() == 3 else
                  self._t[k][:, None].expand(-1, self._t["data"].shape[1], -1)
                  for k in keys]
            return torch.cat(ts, cat_dim), None

        def add_exogenous(self, name, value, add_to_input_map=True):
            self.calls.append(("add_exogenous", name, add_to_input_map))
            self._t[name] = value

        def set_input_map(self, m):
            self.calls.append(("set_input_map", m))
            self.input_map = m

    data = torch.randn(T, n, 1, generator=g)
    u = torch.randn(T, 2, generator=g)
    kw = dict(shipped["c1"], reservoir_size=16, input_scaling=1.)
    for idx, (enc_exo, keep_raw) in enumerate([(True, False), (True, True),
                                               (False, False), (False, True)]):
        ds = FakeDataset(data, u, ei, ew)
        torch.manual_seed(500 + idx)
        in_size = 3 if enc_exo else 1
        out = ref.encode_dataset(ds, ref.SGPEncoder,
                                 dict(kw, input_size=in_size),
                                 encode_exogenous=enc_exo, keep_raw=keep_raw)
        save(f"g5_harness_{idx}", data=data.numpy(), u=u.numpy(),
             edge_index=ei.numpy(), edge_weight=ew.numpy(),
             seed=np.int64(500 + idx), encode_exogenous=np.bool_(enc_exo),
             keep_raw=np.bool_(keep_raw),
             encoded_x=out._t["encoded_x"].numpy(),
             input_map_x=np.array(out.input_map["x"]),
             input_map_u=np.array(out.input_map.get("u", [])),
             n_calls=np.int64(len(out.calls)),
             get_tensors_keys=np.array(out.calls[0][1]))

    # ---- G3b: more DynGESN cases (own generator: appended after the others) ----
    g3 = torch.Generator().manual_seed(4321)
    gesn_cases = {
        "relu_r40": dict(reservoir_size=40, reservoir_layers=2, leaking_rate=.8,
                         spectral_radius=.9, density=.7, input_scaling=1.5,
                         alpha_decay=False, reservoir_activation="relu"),
        "selfnorm_r24": dict(reservoir_size=24, reservoir_layers=2, leaking_rate=.9,
                             spectral_radius=.8, density=.9, input_scaling=1.,
                             alpha_decay=True, reservoir_activation="self_norm"),
        "shipped_r80": dict(reservoir_size=80, reservoir_layers=3, leaking_rate=.9,
                            spectral_radius=.9, density=.7, input_scaling=1.,
                            alpha_decay=False, reservoir_activation="tanh"),
    }
    for idx, (name, kw) in enumerate(gesn_cases.items()):
        n3, t3, f3 = (21, 12, 2) if idx else (16, 20, 3)
        ei3, ew3 = graph(n3, 5 * n3, seed=30 + idx, isolated=(idx != 2))
        torch.manual_seed(310 + idx)
        enc = ref.GESNEncoder(input_size=f3, **kw)
        x3 = torch.randn(t3, n3, f3, generator=g3)
        w3 = ew3                      # edge_weight=None is a TypeError in the reference (:39)
        y3 = enc(x3, ei3, w3)
        save(f"g3_gesn_{name}", x=x3.numpy(), edge_index=ei3.numpy(),
             edge_weight=(ew3.numpy() if w3 is not None else np.zeros(0, np.float32)),
             has_weight=np.bool_(w3 is not None), y=y3.numpy(),
             activation=np.array(kw["reservoir_activation"]), seed=np.int64(310 + idx),
             cfg=np.array([f3, kw["reservoir_size"], kw["reservoir_layers"], kw["leaking_rate"],
                           kw["spectral_radius"], kw["density"], kw["input_scaling"],
                           float(kw["alpha_decay"])]),
             **layer_dump(enc.reservoir))


if __name__ == "__main__":
    main()

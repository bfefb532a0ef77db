"""Golden vectors for the on-the-fly spatial supports ("next" row f3 of SURVEY.md 8f) -- container only.

TEST INFRASTRUCTURE.  Builds the supports with the reference's own ``sgp_spatial_support``
(``lib/sgp_preprocessing.py:108-160``, unmodified, under ``oracle/ref_shim.py``) and applies them
the two ways the reference's loaders do:

* ``SGPLoader.collate`` (``lib/dataloader/sgp_dataloader.py:57-61``):
  ``torch.cat([x] + [adj @ x for adj in support], dim=-1)``
* ``IIDDataset._populate_input_frame`` (``lib/datasets/iid_dataset.py:111-114``), node subset:
  ``torch.cat([tens.index_select(1, node_index)] + [adj.index_select(0, node_index) @ tens ...], -1)``

    python oracle/make_golden_support.py    # writes tests/golden/g9_onthefly_*.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim  # noqa: E402
from oracle.make_golden import graph  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    ref = ref_shim.load_reference()
    g = torch.Generator().manual_seed(99)
    cases = {
        "k2": dict(k=2),
        "bidir_global_k3": dict(k=3, bidirectional=True, global_attr=True),
        "undirected_selfloops_k2": dict(k=2, undirected=True, add_self_loops=True),
    }
    n, e = 40, 200
    ei, ew = graph(n, e, seed=31)
    for name, kw in cases.items():
        sup = ref.sgp_spatial_support(ei, ew, num_nodes=n, **kw)
        x = torch.randn(6, n, 16, generator=g)                       # [steps, nodes, channels]
        full = torch.cat([x] + [adj @ x for adj in sup], dim=-1)     # sgp_dataloader.py:60-61
        node_index = torch.tensor([7, 0, 39, 7, 12])                 # repeats allowed
        sub = torch.cat([x.index_select(1, node_index)] +
                        [adj.index_select(0, node_index) @ x for adj in sup], dim=-1)
        np.savez_compressed(os.path.join(GOLDEN, f"g9_onthefly_{name}.npz"),
                            edge_index=ei.numpy(), edge_weight=ew.numpy(), n=np.int64(n),
                            x=x.numpy(), full=full.numpy(), node_index=node_index.numpy(),
                            sub=sub.numpy(), **{k_: np.array(v_) for k_, v_ in kw.items()})
        print("wrote g9_onthefly_" + name, tuple(full.shape), tuple(sub.shape))


if __name__ == "__main__":
    main()

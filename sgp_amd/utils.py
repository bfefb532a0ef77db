"""``lib/utils.py`` of the reference: the encode harness and the self-normalising
activation."""
import logging
from time import time

import torch
from torch import Tensor
from torch.nn import functional as F

from .sgp_preprocessing import ensure_list

logger = logging.getLogger("sgp_amd")


def encode_dataset(dataset, encoder_class, encoder_kwargs, encode_exogenous=True,
                   keep_raw=False, save_path=None, return_device=False):
    """lib/utils.py:10-47.  ``dataset`` is any object with the slice of the
    ``tsl.data.SpatioTemporalDataset`` interface the harness touches: ``exogenous``,
    ``get_tensors``, ``edge_index``, ``edge_weight``, ``add_exogenous``, ``set_input_map``.

    The reference leaves ``preprocess_exogenous`` unbound when ``encode_exogenous`` is not a
    bool (lib/utils.py:19-22 -> UnboundLocalError); a list is accepted here and treated as the
    list of exogenous keys to encode.

    Beyond the reference: ``return_device=True`` leaves ``encoded_x`` on the GPU (no 36-630 GB
    host round trip before the IID sampler); with ``save_path`` the encoder's
    constructor arguments, per-layer leaking rates and weights are written next to the tensor
    (``save_path + '.encoder.pt'``) so that the embedding can be re-derived (lib/utils.py:34-35
    saves the tensor only, the random weights are lost)."""
    if isinstance(encode_exogenous, bool):
        preprocess_exogenous = dataset.exogenous.keys() if encode_exogenous else []
    else:
        preprocess_exogenous = encode_exogenous
    preprocess_exogenous = ensure_list(preprocess_exogenous)

    x, _ = dataset.get_tensors(['data'] + preprocess_exogenous, preprocess=True, cat_dim=-1)

    encoder = encoder_class(**encoder_kwargs)

    start = time()
    if return_device:                 # every encoder answers on the device of its input
        from . import hip
        hip.require_gpu()
        x = x.cuda()
    encoded_x = encoder(x, edge_index=dataset.edge_index, edge_weight=dataset.edge_weight)
    elapsed = int(time() - start)

    if save_path is not None:
        torch.save(encoded_x, save_path)
        if hasattr(encoder, "describe"):
            torch.save(encoder.describe(), str(save_path) + ".encoder.pt")

    logger.info(f"Dataset encoded in {elapsed // 60}:{elapsed % 60:02d} minutes.")

    dataset.add_exogenous('encoded_x', encoded_x, add_to_input_map=False)

    input_map = {'x': ['encoded_x']}
    u = ([] if encode_exogenous else ['u']) + (['data'] if keep_raw else [])
    if len(u):
        input_map['u'] = u
    dataset.set_input_map(input_map)
    return dataset


def self_normalizing_activation(x: Tensor, r: float = 1.0):
    # lib/utils.py:50-51
    return r * F.normalize(x, p=2, dim=-1)

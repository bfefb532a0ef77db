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
                   keep_raw=False, save_path=None, return_device=False, gpus=None, shard_steps=None):
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
    saves the tensor only, the random weights are lost); ``gpus=N`` (default: SGP_AMD_GPUS, else 1)
    node-partitions the graph over N GPUs behind the same single-process call (``sgp_amd/multigpu.py``:
    one rank per GPU is started and joined inside, the embedding comes back as one host tensor in the
    dataset's node order); ``shard_steps=S`` with ``save_path`` a DIRECTORY streams the embedding to disk in
    time shards of S steps (never holding more than one on the host: embeddings larger than host RAM, the
    629 GB of ``run_largescale_sgp.py:208-212``) instead of the reference's single ``torch.save``
    (lib/utils.py:34-35); ``encoded_x`` is then a ``sgp_amd.datasets.ShardedEmbedding``."""
    exo_keys = _exogenous_to_encode(dataset, encode_exogenous)
    x, _ = dataset.get_tensors(['data'] + exo_keys, preprocess=True, cat_dim=-1)
    encoder = encoder_class(**encoder_kwargs)

    if shard_steps is not None and (save_path is None or return_device):
        raise ValueError("shard_steps needs save_path (a directory) and a host embedding")
    started = time()
    if return_device:                 # every encoder answers on the device of its input
        from . import hip
        hip.require_gpu()
        x = x.cuda()
    from .multigpu import resolve_gpus
    if shard_steps is not None:
        embedding = encoder(x, edge_index=dataset.edge_index, edge_weight=dataset.edge_weight, gpus=gpus,
                            shard_dir=str(save_path), shard_steps=int(shard_steps))
    elif resolve_gpus(gpus) > 1:
        if return_device:
            raise ValueError("return_device=True keeps ONE device tensor: use gpus=1")
        embedding = encoder(x, edge_index=dataset.edge_index, edge_weight=dataset.edge_weight, gpus=gpus)
    else:
        embedding = encoder(x, edge_index=dataset.edge_index, edge_weight=dataset.edge_weight)
    seconds = int(time() - started)
    logger.info(f"Dataset encoded in {seconds // 60}:{seconds % 60:02d} minutes.")

    if save_path is not None:
        if shard_steps is not None:
            import os
            if hasattr(encoder, "describe"):
                torch.save(encoder.describe(), os.path.join(str(save_path), "encoder.pt"))
        else:
            torch.save(embedding, save_path)
            if hasattr(encoder, "describe"):
                torch.save(encoder.describe(), str(save_path) + ".encoder.pt")

    # the embedding becomes the model input 'x'; what was not encoded stays available as 'u'
    dataset.add_exogenous('encoded_x', embedding, add_to_input_map=False)
    dataset.set_input_map(_input_map_after_encoding(encode_exogenous, keep_raw))
    return dataset


def _exogenous_to_encode(dataset, encode_exogenous):
    """True -> every exogenous variable of the dataset, False -> none, a name or list of names ->
    those (lib/utils.py:19-23; the reference only handles the two booleans)."""
    if encode_exogenous is True:
        return list(dataset.exogenous.keys())
    if encode_exogenous is False:
        return []
    return ensure_list(encode_exogenous)


def _input_map_after_encoding(encode_exogenous, keep_raw):
    """lib/utils.py:41-46: 'x' is the embedding; 'u' keeps the un-encoded exogenous 'u' (only when
    nothing exogenous was encoded) and, on request, the raw series."""
    side = []
    if not encode_exogenous:
        side.append('u')
    if keep_raw:
        side.append('data')
    mapping = {'x': ['encoded_x']}
    if side:
        mapping['u'] = side
    return mapping


def self_normalizing_activation(x: Tensor, r: float = 1.0):
    # lib/utils.py:50-51
    return r * F.normalize(x, p=2, dim=-1)

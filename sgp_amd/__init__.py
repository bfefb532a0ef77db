"""sgp_amd -- MI355X-native implementation of SGP's training-free spatiotemporal encoder
(reservoir over time + K-hop graph-shift propagation over nodes) behind the reference's own
Python surface.  Compute lives in ``csrc/libsgp_amd.so`` (hand-written HIP for gfx950); there
is no CPU fallback."""
from . import dataloader, datasets, hip
from .graph import ShiftOperator
from .nn.encoders import GESNEncoder, SGPEncoder, SGPSpatialEncoder, SGPTemporalEncoder
from .nn.reservoir import GESNLayer, GraphESN, Reservoir, ReservoirLayer
from .sgp_preprocessing import (preprocess_adj, preprocess_dataset, reservoir_preprocessing_,
                                sgp_spatial_embedding, sgp_spatial_support)
from .utils import encode_dataset, self_normalizing_activation

__version__ = "0.1.0"

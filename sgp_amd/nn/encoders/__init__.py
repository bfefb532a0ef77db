from .sgp_encoder import SGPEncoder
from .sgp_spatial_encoder import SGPSpatialEncoder
from .sgp_temporal_encoder import SGPTemporalEncoder
from .dyn_gesn_encoder import GESNEncoder

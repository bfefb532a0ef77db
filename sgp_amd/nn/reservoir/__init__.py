from .reservoir import Reservoir, ReservoirLayer
from .graph_reservoir import GraphESN, GESNLayer

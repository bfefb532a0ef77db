from .reservoir import Reservoir, ReservoirLayer

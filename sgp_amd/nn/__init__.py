from . import encoders, models, reservoir

from . import encoders, reservoir

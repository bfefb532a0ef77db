from .sgp_model import SGPInputEncoder

__all__ = ["SGPInputEncoder"]

from .sgp_dataloader import apply_supports

__all__ = ["apply_supports"]

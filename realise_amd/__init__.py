"""realise_amd: the ReaLiSe multimodal forward/backward hot path (DaDaMrX/ReaLiSe src/models.py)
rebuilt for MI355X (gfx950): hand-written HIP kernels behind a C ABI, driven from a drop-in nn.Module."""
from .config import RealiseConfig  # noqa: F401

__all__ = ["RealiseConfig"]

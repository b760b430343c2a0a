"""Optimiser side of the fine-tune path: the caller of the parameter classes' `update()`.  Mirror of the reference's
`bitorch_engine/optim/__init__.py` (same names)."""
from .diode_beta import DiodeMix
from .galore_projector import GaLoreProjector

__all__ = ["DiodeMix", "GaLoreProjector"]

from .layer import MPQWeightParameter, MPQLinearBase

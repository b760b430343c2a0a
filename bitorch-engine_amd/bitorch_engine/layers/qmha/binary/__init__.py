from .layer import BMHA, LearnableBias  # noqa: F401

from .binary import *  # noqa: F401,F403

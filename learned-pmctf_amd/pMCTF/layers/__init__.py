from .modules import *  # noqa: F401,F403

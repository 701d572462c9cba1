"""Import-satisfying stub (see ../README.md). Not gym."""
from . import spaces, error  # noqa: F401


class Env:
    def __init__(self):
        pass


class Wrapper:
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        return getattr(self.env, name)

"""Import-satisfying stub. Not pygame."""


class Surface:
    pass

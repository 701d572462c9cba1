from . import BaseGeom as BaseGeometry  # noqa: F401

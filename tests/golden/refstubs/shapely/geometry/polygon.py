from . import LinearRing, Polygon  # noqa: F401

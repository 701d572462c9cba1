"""Import-satisfying stub (see ../README.md). Not shapely."""

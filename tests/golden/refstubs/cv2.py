"""Import-satisfying stub. Not OpenCV."""

"""
Only the wavelength constants of tfrt/drawing.py:53-60 (scripts pass e.g. ``drawing.YELLOW``
as a wavelength).  The matplotlib / pyvista drawers are out of scope.
"""
VISIBLE_MIN = 380
VISIBLE_MAX = 780
RED = 680
ORANGE = 620
YELLOW = 575
GREEN = 510
BLUE = 450
PURPLE = 400
RAINBOW_6 = [RED, ORANGE, YELLOW, GREEN, BLUE, PURPLE]

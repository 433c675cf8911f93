"""Frame geometry and data-set split: the constants the reference keeps in config.py:3-10,
under the same names (callers star-import them)."""
import math

FRAME_SHAPE = (480, 640)                       # rows, columns of an OSD frame
height, width = FRAME_SHAPE

# field of view of the sensor in degrees -> radians, evaluated as degrees / 180 * pi
_FIELD_OF_VIEW_DEG = {"x": 62.0, "y": 48.6}
viewing_angle_x, viewing_angle_y = (_FIELD_OF_VIEW_DEG[axis] / 180 * math.pi for axis in ("x", "y"))

# the 15 held-out OSD frames; the other 96 of the 111 are the training split
test_indices = [int(t) for t in "2 20 22 27 30 33 37 62 79 90 92 98 101 105 107".split()]
train_indices = sorted(set(range(111)) - set(test_indices))

"""Frame geometry and data-set split (config.py:3-10 of the reference)."""
import math

height = 480
width = 640

# field of view of the sensor, radians
viewing_angle_x = 62.0 / 180 * math.pi
viewing_angle_y = 48.6 / 180 * math.pi

test_indices = [2, 20, 22, 27, 30, 33, 37, 62, 79, 90, 92, 98, 101, 105, 107]
train_indices = [i for i in range(111) if i not in test_indices]

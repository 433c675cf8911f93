"""Frame geometry, data-set split and feature count: the constants of the reference's config.py:3-13
under the same names (its callers star-import them)."""
import math

height, width = 480, 640                       # rows, columns of an OSD frame

viewing_angle_x = 62.0 / 180 * math.pi         # field of view of the sensor, radians
viewing_angle_y = 48.6 / 180 * math.pi

test_indices = [2, 20, 22, 27, 30, 33, 37, 62, 79, 90, 92, 98, 101, 105, 107]   # the 15 held-out OSD frames
train_indices = [i for i in range(111) if i not in test_indices]

num_pairwise_features = 14                     # inputs of the pairwise surface classifiers (downstream of this stage)

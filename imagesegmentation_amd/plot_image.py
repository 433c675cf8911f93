"""Display helper (plot_image.plot_surfaces of the reference); needs matplotlib."""
import numpy as np


def plot_surfaces(surfaces, show=True):
    try:
        import matplotlib.pyplot as plt
    except ImportError as e:      # plotting is optional and never on the compute path
        raise RuntimeError("plot_result=True needs matplotlib, which is not installed") from e
    labels = np.asarray(surfaces)
    rng = np.random.default_rng(0)
    palette = rng.integers(40, 256, size=(int(labels.max()) + 1, 3), dtype=np.uint8)
    palette[0] = 0
    plt.imshow(palette[labels])
    plt.axis("off")
    if show:
        plt.show()

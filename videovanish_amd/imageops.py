"""Host-side helpers for the uint8 image steps (K11): constants the kernels take from the host."""
import numpy as np


def gaussian_taps_21():
    """cv2.getGaussianKernel(21, 0): sigma = 0.3*((21-1)*0.5-1)+0.8 = 3.5; normalised; fp32 taps
    (third-party compose, SURVEY a5.7: GaussianBlur(mask,(21,21),0))."""
    sigma = 0.3 * ((21 - 1) * 0.5 - 1) + 0.8
    x = np.arange(21, dtype=np.float64) - 10
    k = np.exp(-(x * x) / (2 * sigma * sigma))
    return (k / k.sum()).astype(np.float32)

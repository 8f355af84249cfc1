from oracle.blocks import Downsample2D  # noqa: F401

Downsample1D = FirDownsample2D = KDownsample2D = None


def downsample_2d(*a, **k):
    raise NotImplementedError

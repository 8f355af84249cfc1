from oracle.blocks import CrossAttnDownBlock2D, DownBlock2D, UNetMidBlock2DCrossAttn, get_down_block  # noqa: F401


class UNetMidBlock2D:
    pass

from oracle.blocks import TemporalResnetBlock, AlphaBlender  # noqa: F401

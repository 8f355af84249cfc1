from oracle.blocks import BasicTransformerBlock, TemporalBasicTransformerBlock  # noqa: F401

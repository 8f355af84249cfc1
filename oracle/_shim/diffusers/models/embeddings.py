from oracle.blocks import TimestepEmbedding, Timesteps  # noqa: F401


class TextImageProjection:
    pass


class TextImageTimeEmbedding:
    pass


class TextTimeEmbedding:
    pass

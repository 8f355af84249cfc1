ADDED_KV_ATTENTION_PROCESSORS = ()
CROSS_ATTENTION_PROCESSORS = ()


class AttentionProcessor:
    pass


class AttnAddedKVProcessor:
    pass


class AttnProcessor:
    pass

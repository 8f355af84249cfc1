class FromOriginalControlNetMixin:
    pass

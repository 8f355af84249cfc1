class UNet2DConditionModel:
    pass

from dataclasses import dataclass

import torch

from ..utils import BaseOutput


@dataclass
class ControlNetOutput(BaseOutput):
    down_block_res_samples: tuple = None
    mid_block_res_sample: torch.Tensor = None

import enum
from dataclasses import dataclass

import torch


class KarrasDiffusionSchedulers(enum.Enum):
    DPMSolverMultistepScheduler = 1
    UniPCMultistepScheduler = 2


class SchedulerMixin:
    pass


@dataclass
class SchedulerOutput:
    prev_sample: torch.Tensor

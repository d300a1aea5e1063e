"""VHJBControllerConfig -- same fields as the reference (configs/controller/vhjb_controller_config.py:6-68)."""
from dataclasses import dataclass
from typing import Sequence

import numpy as np

from ..gin_lite import configurable

_ARRAY_FIELDS = ("normalization_mean", "normalization_std", "Q", "R", "xf", "uf", "interior_states_mean",
                 "interior_states_std", "boundary_states_mean", "boundary_states_std", "obs_min", "obs_max")


@configurable
@dataclass
class VHJBControllerConfig:
    seed: int
    epsilon: float
    features: Sequence[int]
    normalization_mean: Sequence[float]
    normalization_std: Sequence[float]
    epsilon_scalar: float
    using_batch_norm: bool
    lr: float
    epochs: int
    batch_size: int
    regularization_init_value: float
    regularization_peak_value: float
    regularization_end_value: float
    regularization_num_of_cycles: int
    regularization_warmup_steps_per_cycle: int
    regularization_total_steps_per_cycle: int
    num_of_interior_data: int
    num_of_boundary_data: int
    interior_states_mean: Sequence[float]
    interior_states_std: Sequence[float]
    boundary_states_mean: Sequence[float]
    boundary_states_std: Sequence[float]
    boundary_cost_clip: float
    num_of_trajectories_per_epoch: int
    maximum_step: int
    maximum_buffer_size: int
    Q: Sequence[Sequence[float]]
    R: Sequence[Sequence[float]]
    xf: Sequence[float]
    uf: Sequence[float]
    obs_min: Sequence[float]  # in error coordinates wrap(x - xf)
    obs_max: Sequence[float]

    def __post_init__(self):
        for k in _ARRAY_FIELDS:
            setattr(self, k, np.array(getattr(self, k), dtype=np.float32))

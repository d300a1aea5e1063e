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
        self.validate()

    # ---- not in the reference: consistency checks, so a bad config fails at construction instead of in a kernel ----
    def validate(self):
        n = int(self.xf.shape[0])
        m = int(self.uf.shape[0])
        if self.Q.shape != (n, n):
            raise ValueError(f"Q must be ({n}, {n}), got {self.Q.shape}")
        if self.R.shape != (m, m):
            raise ValueError(f"R must be ({m}, {m}), got {self.R.shape}")
        for name in ("normalization_mean", "normalization_std", "interior_states_mean", "interior_states_std",
                     "boundary_states_mean", "boundary_states_std", "obs_min", "obs_max"):
            if getattr(self, name).shape != (n,):
                raise ValueError(f"{name} must have {n} entries, got {getattr(self, name).shape}")
        if not np.all(self.obs_min < self.obs_max):
            raise ValueError("obs_min must be below obs_max in every coordinate")
        if np.any(self.normalization_std == 0):
            raise ValueError("normalization_std must be non-zero")
        if len(self.features) != 3:
            raise ValueError("features must list three layer widths")
        if self.batch_size <= 0 or self.maximum_step <= 0 or self.maximum_buffer_size <= 0:
            raise ValueError("batch_size, maximum_step and maximum_buffer_size must be positive")

    @property
    def state_dim(self) -> int:
        return int(self.xf.shape[0])

    @property
    def control_dim(self) -> int:
        return int(self.uf.shape[0])

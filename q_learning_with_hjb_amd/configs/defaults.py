"""Stock configurations: the values of the reference's seven .gin files and of
the 10-D quadcopter notebook, as constructors, so nothing has to be parsed to get the headline
set-ups.  (`gin_lite.parse_config_file` reads the reference's own .gin files if you have them.)"""
import numpy as np

from .controller.vhjb_controller_config import VHJBControllerConfig
from .dynamics.dynamics_config import (AcrobotDynamicsConfig, CartpoleDynamicsConfig, LinearDynamicsConfig,
                                       NearHoverQuadcopterConfig, Quadrotors2DConfig)


def linear_dynamics_config(**kw):  # configs/dynamics/linear.gin (double integrator)
    d = dict(seed=0, dt=0.02, A=[[0, 1], [0, 0]], B=[[0], [1]], umin=[-5], umax=[5], x0_mean=[0, 0], x0_std=[1, 1])
    d.update(kw)
    return LinearDynamicsConfig(**d)


def cartpole_dynamics_config(**kw):  # configs/dynamics/cartpole.gin
    d = dict(seed=0, mc=1, mp=0.1, l=1, g=9.81, dt=0.02, x0_mean=[0, 3.14, 0, 0], x0_std=[2.4, 0.05, 1, 0.05],
             umin=[-10], umax=[10])
    d.update(kw)
    return CartpoleDynamicsConfig(**d)


def acrobot_dynamics_config(**kw):  # dynamics/acrobot.py:8-16; start box = the demo start +-0.05
    d = dict(seed=0, dt=0.05, l1=0.5, l2=1, m1=8, m2=8, I1=2, I2=8, g=10, umin=[-25], umax=[25],
             x0_mean=[0.001, 0, 0, 0], x0_std=[0.05, 0.05, 0.05, 0.05])
    d.update(kw)
    return AcrobotDynamicsConfig(**d)


def quadrotors2d_dynamics_config(**kw):  # configs/dynamics/quadrotors2D.gin
    d = dict(seed=0, m=1, r=0.25, g=9.81, I=0.0625, dt=0.05, x0_mean=[0] * 6, x0_std=[1] * 6, umin=[-20, -20],
             umax=[20, 20])
    d.update(kw)
    return Quadrotors2DConfig(**d)


def near_hover_dynamics_config(**kw):  # configs/dynamics/near_hover_quadcopter.gin
    d = dict(seed=0, dt=0.05, g=9.81, m=1, kT=0.91, n0=10, umin=[0, -10, -10], umax=[14.715, 10, 10],
             x0_mean=[0] * 10, x0_std=[1, 1, 1, 0.5, 0.5, 1, 1, 1, 0.5, 0.5])
    d.update(kw)
    return NearHoverQuadcopterConfig(**d)


def _vhjb_common(n):
    return dict(seed=0, epsilon=1e-10, features=[128, 128, 64], normalization_mean=[0.0] * n,
                normalization_std=[1.0] * n, epsilon_scalar=1e-3, using_batch_norm=False, lr=1e-3, epochs=100,
                batch_size=256, regularization_init_value=0.0, regularization_peak_value=1e-5,
                regularization_end_value=0.0, regularization_num_of_cycles=10,
                regularization_warmup_steps_per_cycle=1000, regularization_total_steps_per_cycle=2000,
                num_of_interior_data=10, num_of_boundary_data=10, boundary_cost_clip=10000,
                num_of_trajectories_per_epoch=20, maximum_step=200, maximum_buffer_size=1000000)


def linear_vhjb_config(**kw):  # configs/controller/linear_vhjb_controller.gin
    d = _vhjb_common(2)
    d.update(interior_states_mean=[0, 0], interior_states_std=[1, 1], boundary_states_mean=[0, 0],
             boundary_states_std=[1, 1], Q=np.eye(2), R=[[1]], xf=[0, 0], uf=[0], obs_min=[-2, -3], obs_max=[2, 3])
    d.update(kw)
    return VHJBControllerConfig(**d)


def cartpole_vhjb_config(**kw):  # configs/controller/cartpole_vhjb_controller.gin
    d = _vhjb_common(4)
    xf = [0, 3.1415926, 0, 0]
    d.update(interior_states_mean=xf, interior_states_std=[1.0, 0.2, 4.0, 4.0], boundary_states_mean=xf,
             boundary_states_std=[1.0, 0.2, 4.0, 4.0], Q=np.eye(4), R=[[1]], xf=xf, uf=[0],
             obs_min=[-4.8, -0.418, -1000, -1000], obs_max=[4.8, 0.418, 1000, 1000])
    d.update(kw)
    return VHJBControllerConfig(**d)


def acrobot_vhjb_config(**kw):
    """Acrobot swing-up + balance at the upright (BASELINE configs[2]).  The reference ships NO acrobot VHJB config
    (dynamics/acrobot.py is stale upstream): the cartpole file's hyper-parameters, the cost of the energy-shaping demo
    (controller/acrobot_energy_shaping.py:13, Q = I, R = 1), target [pi, 0, 0, 0]; the observation box never cuts the wrapped
    angles (|e| <= pi), so a swing-up from the hanging position stays live, and bounds the rates."""
    d = _vhjb_common(4)
    d.update(Q=np.eye(4).tolist(), R=[[1.0]], xf=[float(np.pi), 0, 0, 0], uf=[0], obs_min=[-4, -4, -30, -30], obs_max=[4, 4, 30, 30],
             interior_states_mean=[float(np.pi), 0, 0, 0], interior_states_std=[0.5, 0.5, 2, 2],
             boundary_states_mean=[float(np.pi), 0, 0, 0], boundary_states_std=[0.5, 0.5, 2, 2])
    d.update(kw)
    return VHJBControllerConfig(**d)


def quadrotors2d_vhjb_config(**kw):  # configs/controller/quadrotors2DHovering_vhjb_controller.gin
    d = _vhjb_common(6)
    d.update(interior_states_mean=[0] * 6, interior_states_std=[1] * 6, boundary_states_mean=[0] * 6,
             boundary_states_std=[1] * 6, Q=np.eye(6), R=np.eye(2), xf=[0] * 6, uf=[4.905, 4.905],
             obs_min=[-2, -2, -1.5, -5, -5, -2], obs_max=[2, 2, 1.5, 5, 5, 2])
    d.update(kw)
    return VHJBControllerConfig(**d)


def near_hover_vhjb_config(**kw):  # examples/10D_quadcopte.ipynb cells 4, 9, 10 (no gin file upstream)
    d = _vhjb_common(10)
    g, m, kT = 9.81, 1.0, 0.91
    box = [2, 2, 2, 0.5, 0.5, 4, 4, 4, 2, 2]
    d.update(epochs=200, interior_states_mean=[0] * 10, interior_states_std=[1, 1, 1, .5, .5, 1, 1, 1, .5, .5],
             boundary_states_mean=[0] * 10, boundary_states_std=box, Q=np.eye(10), R=np.eye(3), xf=[0] * 10,
             uf=[g * m / kT, 0, 0], obs_min=[-b for b in box], obs_max=box)
    d.update(kw)
    return VHJBControllerConfig(**d)

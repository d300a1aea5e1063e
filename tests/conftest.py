import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # the HIP library and the oracle are build artefacts (git-ignored): make sure both exist and are current
    from q_learning_with_hjb_amd import build_library
    build_library()
    from oracle import oracle as O
    O.build()


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def spot():
    with open(os.path.join(GOLDEN, "spot_values.json")) as f:
        return json.load(f)


def make_dynamics(name):
    """Product Dynamics object with the stock (gin) constants. Construction needs no GPU."""
    from q_learning_with_hjb_amd.configs import defaults as D
    from q_learning_with_hjb_amd.dynamics.acrobot import Acrobot
    from q_learning_with_hjb_amd.dynamics.cartpole import Cartpole
    from q_learning_with_hjb_amd.dynamics.linear import LinearDynamics
    from q_learning_with_hjb_amd.dynamics.quadrotors import NearHoverQuadcopter, Quadrotors2D
    return {
        "linear": lambda: LinearDynamics(D.linear_dynamics_config()),
        "cartpole": lambda: Cartpole(D.cartpole_dynamics_config()),
        "acrobot": lambda: Acrobot(D.acrobot_dynamics_config()),
        "quad2d": lambda: Quadrotors2D(D.quadrotors2d_dynamics_config()),
        "nearhover": lambda: NearHoverQuadcopter(D.near_hover_dynamics_config()),
    }[name]()


def make_vhjb_config(name, **kw):
    from q_learning_with_hjb_amd.configs import defaults as D
    if name == "acrobot":
        # no upstream VHJB config for the acrobot: upright target, wide box
        base = D.cartpole_vhjb_config(xf=[np.pi, 0, 0, 0], interior_states_mean=[np.pi, 0, 0, 0], boundary_states_mean=[np.pi, 0, 0, 0],
                                      obs_min=[-1, -1, -8, -8], obs_max=[1, 1, 8, 8], **kw)
        return base
    return {"linear": D.linear_vhjb_config, "cartpole": D.cartpole_vhjb_config, "quad2d": D.quadrotors2d_vhjb_config,
            "nearhover": D.near_hover_vhjb_config}[name](**kw)


def orc_system(name):
    from oracle import oracle as O
    return O.System.from_dynamics(make_dynamics(name))


SYSTEMS = ["linear", "cartpole", "acrobot", "quad2d", "nearhover"]


def wrapped_diff(a, b, angle_idx):
    """a - b with the listed columns compared modulo 2 pi (wrap discontinuity)."""
    d = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    for i in angle_idx:
        d[..., i] = (d[..., i] + np.pi) % (2 * np.pi) - np.pi
    return d


ANGLE_IDX = {"linear": [], "cartpole": [1], "acrobot": [0, 1], "quad2d": [2], "nearhover": [3, 4]}


ARITHMETICS = {"f32": 0, "bf16x3": 1, "f16x2": 2}


@pytest.fixture(params=list(ARITHMETICS))
def arith(request):
    """Runs a GPU test in every value-network arithmetic of the fused kernels (hjbx.h HJBX_OPT_MLP_ARITHMETIC) against the SAME bounds:
    f32 = float32 MFMA (bitwise an fmaf chain); bf16x3 = each float32 operand split exactly into three bfloat16 pieces, six piece products
    on the bf16 matrix cores (csrc/hjbx_mlp_x3.hpp); f16x2 = each operand scaled per environment and rounded to two float16 pieces (22
    bits), three piece products on the f16 matrix cores (csrc/hjbx_mlp_h2.hpp); float32 accumulation in all of them."""
    from q_learning_with_hjb_amd import _abi
    prev = _abi.set_option(_abi.OPT_MLP_ARITHMETIC, ARITHMETICS[request.param])
    yield request.param
    _abi.set_option(_abi.OPT_MLP_ARITHMETIC, prev)

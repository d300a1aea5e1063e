"""Spong-style energy-shaping swing-up + LQR for the acrobot (reference
controller/acrobot_energy_shaping.py:9-121); CARE hoisted to construction as for the cartpole."""
import numpy as np
import scipy.linalg

from .. import _abi
from .feedback import DeviceFeedbackController


def wrap(q):
    return (q + np.pi) % (2 * np.pi) - np.pi


class AcrobotEnergyShapingController(DeviceFeedbackController):
    def __init__(self, acrobot_system, Q=np.eye(4), R=np.eye(1), eps=1000, K=np.array([1, 2, 1])):
        super().__init__()
        self.acrobot = self.dynamics = acrobot_system
        self.xf = np.array([np.pi, 0, 0, 0])
        self.K, self.Q, self.R, self.eps = np.asarray(K), np.asarray(Q), np.asarray(R), eps
        self._K_lqr, self._P_lqr = self._solve_lqr()

    def get_linearized_dynamics(self):
        """acrobot_energy_shaping.py:23-50"""
        a = self.acrobot
        Minv = np.linalg.inv(a.get_M(self.xf))
        pGpq1 = np.array([-a.m1 * a.g * a.l1 / 2 - a.m2 * a.g * a.l1 - a.m2 * a.g * a.l2 / 2, -a.m2 * a.g * a.l2 / 2])
        pGpq2 = np.array([-a.m2 * a.g * a.l2 / 2, -a.m2 * a.g * a.l2 / 2])
        Alin = np.vstack([np.array([0, 0, 1, 0]), np.array([0, 0, 0, 1]),
                          np.hstack([-Minv @ pGpq1.reshape(2, 1), -Minv @ pGpq2.reshape(2, 1), np.zeros((2, 2))])])
        Blin = np.hstack([np.zeros(2), Minv @ a.get_B()]).reshape(4, 1)
        return Alin, Blin

    def _solve_lqr(self):
        Alin, Blin = self.get_linearized_dynamics()
        P = scipy.linalg.solve_continuous_are(Alin, Blin, self.Q, self.R)
        return np.dot(scipy.linalg.inv(self.R), np.dot(Blin.T, P)), P

    def get_lqr_term(self):
        return self._K_lqr, self._P_lqr

    def _descriptor(self):
        return _abi.make_controller(_abi.CTRL_ACROBOT_ENERGY, 4, 1, self._K_lqr, xf=self.xf, P=self._P_lqr, Kes=self.K,
                                    eps_region=self.eps)

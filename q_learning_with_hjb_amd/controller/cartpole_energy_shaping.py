"""Energy-shaping swing-up + LQR catch for the cartpole (reference
controller/cartpole_energy_shaping.py:7-110).  The reference re-solves the CARE on every call
(:77 -> :58-61); here K, P are solved once at construction -- same numbers, and the per-step law
runs in the kernel."""
import numpy as np
import scipy.linalg

from .. import _abi
from .feedback import DeviceFeedbackController


class CartpoleEnergyShapingController(DeviceFeedbackController):
    def __init__(self, cartpole, Q=np.eye(4), R=np.eye(1), K=np.array([4, 4, 10]), eps_energy=1, eps_state=1) -> None:
        super().__init__()
        self.cartpole = self.dynamics = cartpole
        self.xf = np.array([0, np.pi, 0, 0])
        self.umin, self.umax = self.cartpole.get_control_limit()
        self.Q, self.R, self.K = np.asarray(Q), np.asarray(R), np.asarray(K)
        self.eps_energy, self.eps_state = eps_energy, eps_state
        self._K_lqr, self._P_lqr = self._solve_lqr()

    def get_linearized_dynamics(self):
        """x_dot ~ Alin dx + Blin u about the upright (cartpole_energy_shaping.py:21-44)."""
        c = self.cartpole
        Minv = np.linalg.inv(c.get_M(self.xf))
        pGpq = np.array([[0, 0], [0, -c.mp * c.g * c.l]])
        Alin = np.vstack([np.array([[0, 0, 1, 0], [0, 0, 0, 1]]), np.hstack([-Minv @ pGpq, np.zeros((2, 2))])])
        Blin = np.hstack([np.zeros(2), Minv @ c.get_B()]).reshape(4, 1)
        return Alin, Blin

    def _solve_lqr(self):
        Alin, Blin = self.get_linearized_dynamics()
        P = scipy.linalg.solve_continuous_are(Alin, Blin, self.Q, self.R)
        return np.dot(scipy.linalg.inv(self.R), np.dot(Blin.T, P)), P

    def get_lqr_term(self):
        return self._K_lqr, self._P_lqr

    def energy(self, x):
        """pole "energy" 0.5 theta_dot^2 - cos(theta) (cartpole_energy_shaping.py:90-95)"""
        return 0.5 * x[3] ** 2 - np.cos(x[1])

    def _descriptor(self):
        return _abi.make_controller(_abi.CTRL_CARTPOLE_ENERGY, 4, 1, self._K_lqr, xf=self.xf, Kes=self.K,
                                    eps_energy=self.eps_energy, eps_state=self.eps_state)

"""Host-side set-up helpers (reference utils/utils.py:30-80; its np_collate :7-14 has no counterpart: the replay buffer is device resident). Not on the device path."""
import numpy as np
import scipy.linalg


def solve_continuous_are(A, B, Q, R):
    """Stabilising solution of A'P + PA - P B R^-1 B' P + Q = 0 via the ordered real Schur form of
    the Hamiltonian [[A, -B R^-1 B'], [-Q, -A']] with the left-half-plane eigenvalues leading
    (utils/utils.py:67-80).  Used once at set-up for the terminal cost x'Px (vhjb.py:156-160)."""
    A = np.asarray(A, np.float64)
    B = np.asarray(B, np.float64)
    Q = np.asarray(Q, np.float64)
    R = np.asarray(R, np.float64)
    n = Q.shape[0]
    S = B @ np.linalg.inv(R) @ B.T
    H = np.block([[A, -S], [-Q, -A.T]])
    _, Z, sdim = scipy.linalg.schur(H, sort="lhp")
    if sdim != n:
        raise np.linalg.LinAlgError(f"Hamiltonian has {sdim} stable eigenvalues, expected {n}")
    return Z[n:, :n] @ np.linalg.inv(Z[:n, :n])


def linearize(dynamics, xf, uf, h=1e-6):
    """(A, B) = d x_dot / d(x, u) at (xf, uf) by central differences on the DEVICE dynamics_step in
    float64: one batched kernel call over the 2(n+m) perturbed points.  (The reference uses
    jax.jacobian, vhjb.py:159; for these smooth systems the two agree to ~1e-9.)"""
    n, m = dynamics.get_dimension()
    xf = np.asarray(xf, np.float64).reshape(n)
    uf = np.asarray(uf, np.float64).reshape(m)
    X = np.tile(xf, (2 * (n + m), 1))
    U = np.tile(uf, (2 * (n + m), 1))
    for i in range(n):
        X[2 * i, i] += h
        X[2 * i + 1, i] -= h
    for j in range(m):
        U[2 * (n + j), j] += h
        U[2 * (n + j) + 1, j] -= h
    XD = dynamics.dynamics_step(X, U)
    J = (XD[0::2] - XD[1::2]) / (2 * h)  # row k = derivative wrt coordinate k
    return J[:n].T.copy(), J[n:].T.copy()

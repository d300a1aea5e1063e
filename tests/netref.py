"""Float64 NumPy restatement of the ReLU value network of controller/vhjb.py:17-60 with the pieces the float32 parity tests need
beyond the oracle's plain evaluation (TEST INFRASTRUCTURE, like oracle/):

  * the ReLU derivative masks can be FORCED (a unit whose pre-activation is within rounding of zero may legitimately be taken on either
    side by a float32 evaluation: dV/dx of a ReLU network is discontinuous there) -- `grad(masks=...)`;
  * the units within a relative distance KINK of their kink, per environment -- `kink_candidates`;
  * per-element TERM SCALES: the network evaluated with |weights| along the active paths, i.e. the sum of the magnitudes of the terms that
    make up V and each component of dV/dx.  A float32 result is good to a few ulps of THIS, not of the result itself (the LQR-embedded
    networks carry +-q pairs that cancel).  With `split=True` every operand of the four 128 / 64-wide products is replaced by
    |operand| + 2^-17 max|operand| (max over the weight matrix / over the environment's inputs of that product): the absolute error floor
    of the f16x2 split arithmetic (include/hjbx.h, HJBX_OPT_MLP_ARITHMETIC = 2).
"""
import itertools

import numpy as np


class NetRef:
    def __init__(self, W, mean, std, xf, eps_scalar, wrap):
        """W: the three float32 weight matrices as float64 arrays (in, out); wrap(e) -> wrapped error coordinates (float64)."""
        self.W = [np.asarray(w, np.float64) for w in W]
        self.mean, self.std, self.xf = (np.asarray(v, np.float64)[None, :] for v in (mean, std, xf))
        self.eps, self.wrap = float(eps_scalar), wrap

    @classmethod
    def of(cls, ctl, W, orc_system):
        from oracle import oracle as O
        vf = ctl.value_function_approximator
        assert vf.activation == "relu"
        return cls(W, vf._np["mean"], vf._np["std"], vf._np["xf"], vf.epsilon_scalar, lambda e: O.wrap(orc_system, e))

    def forward(self, x):
        W1, W2, W3 = self.W
        e = self.wrap(np.asarray(x, np.float64) - self.xf)
        z = (e - self.mean) / self.std
        a1 = z @ W1
        h1 = np.maximum(a1, 0.0)
        a2 = h1 @ W2
        h2 = np.maximum(a2, 0.0)
        y = h2 @ W3
        return dict(e=e, z=z, a1=a1, h1=h1, a2=a2, h2=h2, y=y, V=(y * y).sum(1) + self.eps * (e * e).sum(1),
                    t1=np.abs(z) @ np.abs(W1), t2=h1 @ np.abs(W2))

    def grad(self, fw, m1=None, m2=None):
        """dV/dx with the ReLU derivative masks m1, m2 (bool (B, 128); default: pre-activation > 0, vhjb.py's jax.nn.relu convention)."""
        W1, W2, W3 = self.W
        m1 = fw["a1"] > 0 if m1 is None else m1
        m2 = fw["a2"] > 0 if m2 is None else m2
        d2 = np.where(m2, (2.0 * fw["y"]) @ W3.T, 0.0)
        d1 = np.where(m1, d2 @ W2.T, 0.0)
        return (d1 @ W1.T) / self.std + 2.0 * self.eps * fw["e"]

    def kink_margin(self, fw):
        """min over the 256 hidden units of |pre-activation| / sum |terms of that pre-activation|."""
        r1 = np.abs(fw["a1"]) / np.maximum(fw["t1"], 1e-300)
        r2 = np.abs(fw["a2"]) / np.maximum(fw["t2"], 1e-300)
        return np.minimum(r1.min(1), r2.min(1))

    def kink_candidates(self, fw, kink):
        """-> (c1, c2) bool (B, 128): units whose pre-activation is within `kink` x (sum of |terms|) of zero (units with no terms at all, e.g.
        behind an all-zero column, are not candidates: every evaluation gives exactly 0 there)."""
        c1 = (np.abs(fw["a1"]) < kink * fw["t1"]) & (fw["t1"] > 0)
        c2 = (np.abs(fw["a2"]) < kink * fw["t2"]) & (fw["t2"] > 0)
        return c1, c2

    def forced_grads(self, fw, rows, kink, max_units=3):
        """For the environments `rows` (indices): every dV/dx obtainable by taking each near-kink unit on either side.
        -> (list of (len(rows), n) arrays, one per combination), number of environments with more than `max_units` candidates (those
        keep only their first `max_units` candidates free)."""
        sub = {k: v[rows] for k, v in fw.items()}
        c1, c2 = self.kink_candidates(sub, kink)
        cand = np.concatenate([c1, c2], axis=1)                                # (R, 256)
        count = cand.sum(1)
        order = np.argsort(~cand, axis=1, kind="stable")[:, :max_units]        # the first candidates of each row (padding: non-candidates)
        is_c = np.take_along_axis(cand, order, axis=1)
        base = np.concatenate([sub["a1"] > 0, sub["a2"] > 0], axis=1)
        out = []
        rr = np.arange(len(rows))[:, None]
        for bits in itertools.product((False, True), repeat=max_units):
            m = base.copy()
            flip = is_c & np.asarray(bits)[None, :]
            m[rr, order] = np.where(flip, ~m[rr, order], m[rr, order])
            out.append(self.grad(sub, m[:, :128], m[:, 128:]))
        return out, int((count > max_units).sum())

    def term_scales(self, fw, split=False):
        """-> (TV (B,), Tg (B, n), Ty (B, 64)): sums of |terms| of V, of each component of dV/dx and of each y, along the active paths."""
        A1, A2, A3 = (np.abs(w) for w in self.W)
        m1, m2 = fw["a1"] > 0, fw["a2"] > 0

        def eff_w(A):
            return A + 2.0 ** -17 * A.max() if split else A

        def eff_a(v):
            return v + 2.0 ** -17 * v.max(1, keepdims=True) if split else v
        A2e, A3e = eff_w(A2), eff_w(A3)
        t1 = np.where(m1, np.abs(fw["z"]) @ A1, 0.0)                            # layer 1 is float32 in every arithmetic
        t2 = np.where(m2, eff_a(t1) @ A2e, 0.0)
        ty = eff_a(t2) @ A3e                                                   # >= |y|
        d2 = np.where(m2, eff_a(2.0 * ty) @ A3e.T, 0.0)
        d1 = np.where(m1, eff_a(d2) @ A2e.T, 0.0)
        tg = (d1 @ A1.T) / np.abs(self.std) + 2.0 * self.eps * np.abs(fw["e"])
        tv = 2.0 * (np.abs(fw["y"]) * ty).sum(1) + self.eps * (fw["e"] ** 2).sum(1)
        return tv, tg, ty


def smooth_term_scales(W, mean, std, eps_scalar, e, activation):
    """Per-element term scales of V and dV/dx for a network with a smooth activation ("tanh", "sin"), float64: the sums of the MAGNITUDES of the
    terms that make up each y_j, V and each component of dV/dx along the actual activations and activation derivatives (the analogue of
    NetRef's ReLU term scales).  W: float64 (in, out) matrices; e: wrapped error coordinates.  -> (scale of V (B,), scale of dV/dx (B, n))"""
    W1, W2, W3 = (np.asarray(w, np.float64) for w in W)
    f, df = {"tanh": (np.tanh, lambda a: 1.0 - np.tanh(a) ** 2), "sin": (np.sin, np.cos)}[activation]
    std = np.abs(np.asarray(std, np.float64)).reshape(1, -1)
    z = (e - np.asarray(mean, np.float64).reshape(1, -1)) / std
    a1 = z @ W1
    h1, s1 = f(a1), np.abs(df(a1))
    a2 = h1 @ W2
    h2, s2 = f(a2), np.abs(df(a2))
    y = h2 @ W3
    t_y = np.abs(h2) @ np.abs(W3)
    sV = (2.0 * np.abs(y) * t_y).sum(1) + (y * y).sum(1) + eps_scalar * (e * e).sum(1)
    G = (((((2.0 * t_y) @ np.abs(W3).T) * s2) @ np.abs(W2).T) * s1) @ np.abs(W1).T / std + 2.0 * eps_scalar * np.abs(e)
    return sV, G

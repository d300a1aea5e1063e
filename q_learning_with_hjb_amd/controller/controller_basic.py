"""Plugin surface #2: anything with `get_control_efforts(x)` can drive `Dynamics.simulate`
(the reference's 1-method interface, controller/controller_basic.py:1-5)."""
from __future__ import annotations


class Controller:
    """Feedback law u = pi(x).

    Subclasses implement `get_control_efforts`; in this package they evaluate on the device and accept a
    single state `(n,)` or a batch `(B, n)`, numpy or torch (see `controller.feedback.DeviceFeedbackController`
    for the closed-form laws and `controller.vhjb.VHJBController` for the learned one)."""

    def __init__(self) -> None:
        pass

    def get_control_efforts(self, x):
        """-> control `(m,)` / `(B, m)` for state(s) `x`."""
        raise NotImplementedError(f"{type(self).__name__} does not define a control law")

    def __call__(self, x):
        return self.get_control_efforts(x)

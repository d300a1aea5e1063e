"""Plugin surface #2 of the reference (controller/controller_basic.py:1-5)."""


class Controller():
    def __init__(self) -> None:
        pass

    def get_control_efforts(self, x):
        raise NotImplementedError

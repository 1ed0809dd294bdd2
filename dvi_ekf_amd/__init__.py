"""dvi_ekf_amd -- MI355X-native batched error-state Kalman filter behind the Python
``Filter`` / ``Simulator`` / ``Config`` surface of salehahr/dvi-ekf.

``Config`` and the host-side preparation import without a GPU; ``Filter``, ``Simulator`` and
``BatchedESKF`` need the HIP library (dvi_ekf_amd/libdvi_eskf.so) and a HIP device."""
from .config import Config  # noqa: F401


def __getattr__(name):
    if name == "Simulator":
        from .simulator import Simulator
        return Simulator
    if name == "Filter":
        from .filter import Filter
        return Filter
    if name == "BatchedESKF":
        from .batch import BatchedESKF
        return BatchedESKF
    if name in ("State", "ErrorState"):
        from . import state
        return getattr(state, name)
    if name == "Quaternion":
        from .quaternion import Quaternion
        return Quaternion
    raise AttributeError(name)

"""dvi_ekf_amd -- MI355X-native batched error-state Kalman filter behind the
Python ``Filter`` / ``Simulator`` / ``Config`` surface of salehahr/dvi-ekf."""
from .config import Config  # noqa: F401

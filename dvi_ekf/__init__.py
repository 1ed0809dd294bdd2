"""The reference's package name for this build: ``from dvi_ekf import Config, Simulator`` -- the first line of the
reference's ``main.py`` -- resolves to ``dvi_ekf_amd`` (same classes, same objects), so that script runs unedited."""
from dvi_ekf_amd import Config  # noqa: F401


def __getattr__(name):
    import dvi_ekf_amd
    return getattr(dvi_ekf_amd, name)

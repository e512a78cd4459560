"""vins-mobile_amd — MI355X-native VIO hot path (KLT front-end + sliding-window solve).

The directory name is not a Python identifier; import it with
``importlib.import_module("vins-mobile_amd")`` (see ``__graft_entry__.py``).
"""
from . import abi, backend, estimator, frontend, loop, multi, pnp, posegraph, replay, synth, window  # noqa: F401

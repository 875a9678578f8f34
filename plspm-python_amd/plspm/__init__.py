"""plspm -- drop-in MI355X (gfx950) estimator backend for PLS path modelling.

Same public surface as GoogleCloudPlatform/plspm-python (``plspm.plspm.Plspm``, ``plspm.config.Config`` /
``Structure`` / ``MV``, ``plspm.mode.Mode``, ``plspm.scheme.Scheme``, ``plspm.scale.Scale``), but the
iterative weight solver and the bootstrap loop run as hand-written HIP kernels in ``libplspm_hip.so``
(C-ABI: ``include/plspm_hip.h``), reached through ctypes.  There is no CPU fallback: importing the
estimator without the built library, or running it without a HIP device, raises.
"""
name = "plspm"
__version__ = "0.1.0+mi355x"

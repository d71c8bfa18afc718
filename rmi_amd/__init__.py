"""rmi_amd -- MI355X-native trainer for the leaf-fitting hot path of two-layer RMIs.

The compute lives in ``rmi_amd/csrc`` (hand-written HIP for gfx950) behind the C ABI declared
in ``include/rmi_hip.h``.  This package is the host-side mirror of the reference's
``rmi_lib::train`` surface; it never imports anything from ``oracle/``.
"""
__version__ = "0.1.0"

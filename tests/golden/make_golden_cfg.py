"""Configs shared by the make_golden*.py scripts and the tests: they live in the package (ptv3_hip/configs.py)."""
import os
import sys

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                    "pointcept-keypointdetection_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from ptv3_hip.configs import ORDERS, TINY_CFG, FORK_CFG, SEMSEG_CFG, TINY_M2_CFG  # noqa: E402,F401

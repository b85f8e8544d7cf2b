#!/usr/bin/env python3
"""Tiny launches for a rocprofv3 kernel trace: what a 100k x 1 pricing costs on the device itself."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optionslab_amd import _hip  # noqa: E402

for n, m in ((256, 1), (10_000, 1), (100_000, 1), (100_000, 16), (1_000_000, 1)):
    for i in range(12):
        _hip.european(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, True, n, m, i, True)

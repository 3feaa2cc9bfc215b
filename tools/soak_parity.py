#!/usr/bin/env python3
"""Launcher of the randomised differential soak (tests/soak_parity.py -- it drives the CPU oracle, so it lives with the tests).
  python tools/soak_parity.py [seconds=300] [seed=1] [checkpoint.json] > gpurun_out/soak.json"""
import os
import runpy
import sys

if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    runpy.run_path(os.path.join(root, "tests", "soak_parity.py"), run_name="__main__")

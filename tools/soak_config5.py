#!/usr/bin/env python3
"""Launcher of tests/soak_config5.py (the soak drives the ORACLE as its checker, so its code lives under tests/).
python tools/soak_config5.py [seconds=60]"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
runpy.run_module("tests.soak_config5", run_name="__main__")

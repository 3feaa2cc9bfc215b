"""kitti_motion_compensation_amd -- MI355X (gfx950) native per-point LiDAR deskew engine.

The product is two shared libraries built from csrc/:
  lib/libkmc_hip.so                         C-ABI + hand-written HIP kernels   (include/kmc_hip.h)
  lib/libkitti_motion_compensation_lib.so   C++ drop-in kmc::MotionCompensateFrame(Frame const&, Time) API

This Python package only holds the ctypes plumbing that tests/ and bench.py use to drive the C-ABI
(`capi`) and the frame-range sharding helper for the one-process-per-GPU launch (`sharding`).
"""
from . import capi  # noqa: F401

__all__ = ["capi"]

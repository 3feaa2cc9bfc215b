"""The C-ABI library loads on a GPU-less box and exports every symbol include/kmc_hip.h declares; the product never
touches the oracle.  No compute calls here."""
import os
import re
import subprocess

import pytest

from kitti_motion_compensation_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    with open(os.path.join(ROOT, "include", "kmc_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kmc_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    declared = _declared_symbols()
    assert len(declared) >= 20
    assert sorted(capi.SIGNATURES) == declared


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    for name in _declared_symbols():
        assert hasattr(L, name), name
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\sT\s+(kmc_[a-z0-9_]+)", out))
    assert set(_declared_symbols()) <= exported
    assert L.kmc_abi_version() == 7  # 7: the direct queue is opt-in (kmc_hip_set_direct_dispatch, kmc_hip_direct_dispatch_active), kmc_host_pool_alloc_near, KMC_LIST_ROUTE / KMC_DIRECT_LANES / KMC_MAPPED_WAVES / KMC_DIRECT_DEBUG retired; 6: kmc_hip_completion_word_fallbacks (an in-place wait that finds the stream idle without the word synchronises the stream and carries on); 5: kmc_hip_frame_queue_dropped + the sticky join error, a list call launches once per tier present; 4: kmc_device_info.any_order_dispatch (run-time probe), queued kmc_hip_deskew_f64cols_begin calls, kmc_hip_deskew_frames_f32 as one launch


def test_library_contains_gfx950_code_object():
    out = subprocess.run(["strings", "-n", "6", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "gfx950" in out


def test_status_strings():
    assert capi.status_string(capi.OK) == "KMC_OK"
    assert "no CPU fallback" in capi.status_string(capi.ERR_NO_DEVICE)


def test_no_device_means_loud_failure_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.KmcError) as e:
        capi.Context(0)
    assert e.value.status == capi.ERR_NO_DEVICE


def test_product_never_references_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    bad = []
    pkg = os.path.join(ROOT, "kitti_motion_compensation_amd")
    for base in (pkg, os.path.join(ROOT, "include")):
        for dp, _, files in os.walk(base):
            for fn in files:
                if fn.endswith((".so", ".o", ".pyc")):
                    continue
                with open(os.path.join(dp, fn), errors="ignore") as f:
                    txt = f.read()
                if re.search(r"kmc_oracle|kmo_|from oracle|import oracle|libkmc_oracle", txt):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad
    # ... nor do the measurement tools: a tool that needs the checker lives under tests/ (soak_*.py, stress_direct_queue_process.py)
    tools = os.path.join(ROOT, "tools")
    for fn in os.listdir(tools):
        if fn.endswith((".py", ".hip", ".cpp", ".sh")):
            with open(os.path.join(tools, fn), errors="ignore") as f:
                if re.search(r"kmc_oracle|kmo_|from oracle|import oracle|libkmc_oracle", f.read()):
                    bad.append(os.path.join(tools, fn))
    assert not bad, bad
    out = subprocess.run(["nm", "-D", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "kmo_" not in out
    ldd = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in ldd


def test_python_plumbing_rejects_strided_or_mistyped_buffers():
    """The C-ABI takes bare pointers; the ctypes layer must refuse views it would misread."""
    import numpy as np

    a = np.zeros((16, 8), dtype=np.float32)[:, :4]  # strided view
    with pytest.raises(ValueError):
        capi._ptr(a, np.float32)
    with pytest.raises(TypeError):
        capi._ptr(np.zeros((4, 4), dtype=np.float64), np.float32)
    assert capi._ptr(np.zeros((4, 4), dtype=np.float32), np.float32) != 0
    assert capi._ptr(None) is None


def test_host_pool_declines_without_a_device_and_never_claims_foreign_memory():
    """The page-locked pool behind the drop-in's containers (kmc_host_pool_*): without a HIP device (this CPU box; or KMC_HOST_POOL=0)
    kmc_host_pool_alloc fails with KMC_ERR_NO_DEVICE so that the containers fall back to ordinary memory -- an allocation is not a
    computation, there is still no CPU fallback for the deskew --, and the pool never claims a pointer it did not hand out."""
    import ctypes as C

    import numpy as np

    from kitti_motion_compensation_amd import capi

    L = capi.lib()
    a = np.zeros(1 << 16)
    assert L.kmc_host_pool_owns(a.ctypes.data, a.nbytes) == 0
    assert L.kmc_host_pool_free(a.ctypes.data) == 0 and L.kmc_host_pool_free(None) == 0
    p = C.c_void_p()
    rc = L.kmc_host_pool_alloc(1 << 20, C.byref(p))
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        assert rc == capi.OK and p.value and L.kmc_host_pool_owns(p, 1 << 20) == 1 and L.kmc_host_pool_owns(p, (1 << 21) + 1) == 0
        assert L.kmc_host_pool_free(p) == 1 and L.kmc_host_pool_owns(p, 16) == 0 and L.kmc_host_pool_trim() >= 1
    else:
        assert rc == capi.ERR_NO_DEVICE and not p.value and L.kmc_host_pool_trim() == 0
    assert L.kmc_host_pool_alloc(0, C.byref(p)) in (capi.OK, capi.ERR_NO_DEVICE)


def test_every_environment_knob_the_product_reads_is_in_the_headers_table():
    """the knobs had grown to 15, several of them measurement switches that kept dead routes alive.  Every getenv("KMC_*")
    in the product sources must be listed in include/kmc_hip.h's table (with its default and the test that covers the non-default value),
    the retired ones must be gone from the code, and the count stays below round 5's."""
    import re

    srcs = []
    for d in (os.path.join(ROOT, "kitti_motion_compensation_amd", "csrc"), os.path.join(ROOT, "kitti_motion_compensation_amd", "csrc", "api"), os.path.join(ROOT, "include"),
              os.path.join(ROOT, "include", "kitti_motion_compensation")):
        srcs += [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hip", ".h", ".hpp", ".cpp"))]
    srcs.append(os.path.join(ROOT, "tools", "motion_compensate_runs.cpp"))
    read = set()
    for path in srcs:
        with open(path) as fh:
            read |= set(re.findall(r'getenv\("(KMC_[A-Z0-9_]+)"\)', fh.read()))
    with open(os.path.join(ROOT, "include", "kmc_hip.h")) as fh:
        header = fh.read()
    table = header[header.index("Environment (read at kmc_hip_create"):header.index("Data conventions")]
    missing = sorted(k for k in read if k not in table)
    assert not missing, f"read by the product but not in kmc_hip.h's table: {missing}"
    retired = {"KMC_LIST_ROUTE", "KMC_DIRECT_LANES", "KMC_MAPPED_WAVES", "KMC_DIRECT_DEBUG"}
    assert not (read & retired), read & retired
    assert len(read) < 15, sorted(read)

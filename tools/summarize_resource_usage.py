#!/usr/bin/env python3
"""hipcc -Rpass-analysis=kernel-resource-usage output (make -C kitti_motion_compensation_amd/csrc resource-usage) -> one line per kernel
instantiation: demangled name, VGPRs, AGPRs, SGPRs, SGPR spills, scratch bytes per lane, occupancy (waves per SIMD), LDS bytes.
  make -C kitti_motion_compensation_amd/csrc resource-usage 2>&1 | python tools/summarize_resource_usage.py > profiles/r04_resource_usage.txt"""
import re
import subprocess
import sys


def parse(text):
    rows = []
    for b in re.split(r"remark: [^\n]*Function Name: ", text)[1:]:
        name = b.split("\n")[0].split(" [")[0].strip()

        def g(k):
            m = re.search(k + r": (\S+)", b)
            return m.group(1) if m else "?"

        rows.append(dict(mangled=name, vgprs=g("VGPRs"), agprs=g("AGPRs"), sgprs=g("SGPRs"), sgpr_spills=g("SGPRs Spill"),
                         scratch=g(r"ScratchSize \[bytes/lane\]"), occupancy=g(r"Occupancy \[waves/SIMD\]"), lds=g(r"LDS Size \[bytes/block\]")))
    return rows


def demangle(names):
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        try:
            out = subprocess.run([tool], input="\n".join(names) + "\n", capture_output=True, text=True, check=True).stdout.splitlines()
            if len(out) == len(names):
                return out
        except Exception:
            pass
    return names


def main():
    rows = parse(sys.stdin.read())
    names = demangle([r["mangled"] for r in rows])
    print("# kernel instantiation | VGPRs | AGPRs | SGPRs | SGPR spills | scratch B/lane | occupancy (waves/SIMD) | LDS B/block")
    for r, n in zip(rows, names):
        short = re.sub(r"\(.*$", "", n.replace("void ", "").replace("kmc_dev::", ""))
        print(f"{short} | {r['vgprs']} | {r['agprs']} | {r['sgprs']} | {r['sgpr_spills']} | {r['scratch']} | {r['occupancy']} | {r['lds']}")


if __name__ == "__main__":
    main()

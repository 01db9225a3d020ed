"""Build libmmn_hip.so in-tree with hipcc for gfx950.  `python -m multimodn_amd.build`."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "mmn_kernels.hip")
INC = os.path.join(os.path.dirname(HERE), "include")
OUT = os.path.join(HERE, "libmmn_hip.so")


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm)")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    csrc = os.path.dirname(SRC)                           # mmn_kernels.hip includes the *.inc files next to it
    sources = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".inc"))]
    newest = max(os.path.getmtime(p) for p in sources + [os.path.join(INC, "mmn_hip.h")])
    return os.path.getmtime(OUT) < newest


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    # -amdgpu-mfma-vgpr-form: accumulators stay in VGPRs (gfx950's register file is unified).  With AGPR accumulators
    # hipcc rotated k_wgrad's sixteen accumulator tiles through v_accvgpr_read / _write pairs in every unrolled loop
    # body (reads that wait for the MFMA that produced the tile): k_wgrad 18.1 -> 17.5 us, the step -1.4 us.
    cmd = [hipcc_path(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-amdgpu-mfma-vgpr-form",
           f"-I{INC}", SRC, "-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

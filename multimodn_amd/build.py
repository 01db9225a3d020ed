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


def source_files():
    csrc = os.path.dirname(SRC)                           # mmn_kernels.hip includes the *.inc files next to it
    return sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".inc"))) + [os.path.join(INC, "mmn_hip.h")]


def source_hash() -> str:
    """sha256 over the library's sources (names and contents, fixed order), first 16 hex digits: baked into the library at
    build time (`mmn_source_hash()`), compared by `needs_build()` and by `hip.load()` - a library that was not built from
    the sources next to it does not load (modification times say nothing on a box that received the tree by copy)."""
    import hashlib
    h = hashlib.sha256()
    for p in source_files():
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def built_hash(path: str = OUT):
    """What `mmn_source_hash()` of an existing library says (None: no such library / symbol).  Asked in a CHILD process:
    loading a stale library here and the rebuilt one afterwards would leave two copies mapped in the building process."""
    if not os.path.exists(path):
        return None
    code = ("import ctypes,sys\n"
            "try:\n"
            "    lib = ctypes.CDLL(sys.argv[1]); fn = lib.mmn_source_hash; fn.restype = ctypes.c_char_p; print(fn().decode())\n"
            "except (OSError, AttributeError):\n"
            "    print('')\n")
    try:
        out = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=120)
    except (OSError, subprocess.TimeoutExpired):
        return None
    h = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else ""
    return h or None


def needs_build() -> bool:
    return not os.path.exists(OUT) or built_hash() != source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    # -amdgpu-mfma-vgpr-form: accumulators stay in VGPRs (gfx950's register file is unified).  With AGPR accumulators
    # hipcc rotated k_wgrad's sixteen accumulator tiles through v_accvgpr_read / _write pairs in every unrolled loop
    # body (reads that wait for the MFMA that produced the tile): k_wgrad 18.1 -> 17.5 us, the step -1.4 us.
    cmd = [hipcc_path(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-amdgpu-mfma-vgpr-form",
           f'-DMMN_SOURCE_HASH="{source_hash()}"', f"-I{INC}", SRC, "-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

"""Register / scratch / LDS use of every kernel in libmmn_hip.so (from the code object's metadata notes).
`python tools/kernel_resources.py [substring ...]`; MMN_LIB_PATH names another library (tools/ab variants)."""
import os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.environ.get("MMN_LIB_PATH") or os.path.join(REPO, "multimodn_amd", "libmmn_hip.so")
llvm = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as td:
    tmp = os.path.join(td, "lib.so")
    subprocess.run(["cp", so, tmp], check=True)
    subprocess.run([f"{llvm}/llvm-objdump", "--offloading", tmp], check=True, capture_output=True)
    co = [f for f in os.listdir(td) if "gfx950" in f][0]
    txt = subprocess.run([f"{llvm}/llvm-readelf", "--notes", os.path.join(td, co)], check=True, capture_output=True, text=True).stdout
    dem = {}
    rows = []
    for blk in txt.split("- .agpr_count")[1:]:
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
        rows.append((g("name"), int(g("vgpr_count")), int(g("sgpr_count")), int(g("private_segment_fixed_size")),
                     int(g("group_segment_fixed_size")), int(g("vgpr_spill_count"))))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    print(f"{'kernel':58s} vgpr sgpr scratch_B static_lds_B spilled_vgprs")
    for (n, v, s, sc, lds, sp), d in zip(rows, names):
        d = re.sub(r"\(anonymous namespace\)::", "", d).split("(")[0].replace("void ", "")
        if len(sys.argv) > 1 and not any(a in d for a in sys.argv[1:]):
            continue
        print(f"{d[:58]:58s} {v:4d} {s:4d} {sc:7d} {lds:12d} {sp:6d}")

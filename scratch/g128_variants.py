"""Lab builds of the library with other settings of k_gemm128x's knobs (csrc/gemm.hip: PTX_G128_NM / _ORDER / _PIPE ...):
    python scratch/g128_variants.py "nm4:-DPTX_G128_NM=4" "pipe:-DPTX_G128_PIPE=1" ...   ->  scratch/lab/lib_g128_<tag>.so
then on the GPU box: python scratch/gemm128_lab.py --libs scratch/lab/lib_g128_*.so"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(R, "proxytransformation_amd", "csrc")
out = os.path.join(R, "scratch", "lab")
os.makedirs(out, exist_ok=True)
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -ffp-contract=off -fno-fast-math -Wno-unused-function".split()
objs = [os.path.join(C, f) for f in sorted(os.listdir(C)) if f.endswith(".o") and f not in ("gemm.o", "api_testhooks.o")]
procs = []
for spec in sys.argv[1:]:
    tag, _, defs = spec.partition(":")
    o = os.path.join(out, f"gemm_{tag}.o")
    procs.append((tag, o, subprocess.Popen(["/opt/rocm/bin/hipcc"] + flags + defs.split() + ["-c", os.path.join(C, "gemm.hip"), "-o", o])))
for tag, o, p in procs:
    assert p.wait() == 0, tag
    so = os.path.join(out, f"lib_g128_{tag}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(C, "exports.map"),
                           "-o", so, o] + objs)
    os.remove(o)
    print("built", so)

"""Lab builds: plain instead of streaming (nt) loads of the features in k_img_pool / k_img_mean16.
python scratch/ld_policy_variants.py -> scratch/lab/lib_{poolld,meanld,bothld}.so ; bash scratch/ab_interleaved.sh real poolld 32"""
import os, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(R, "proxytransformation_amd", "csrc")
out = os.path.join(R, "scratch", "lab"); os.makedirs(out, exist_ok=True)
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function".split()
def plain(src):
    return src.replace("__builtin_nontemporal_load(reinterpret_cast<const u4u2 *>(", "*(reinterpret_cast<const u4u2 *>(")
def build(name, files):
    objs = []
    for f in sorted(os.listdir(C)):
        if not f.endswith(".hip"): continue
        if f in files:
            src = open(os.path.join(C, f)).read()
            if f == "imgproxy16.hip":       # only the mean pass
                a = src.index("void k_img_mean16("); b = src.index("\n}\n", a)
                src = src[:a] + plain(src[a:b]) + src[b:]
            else:
                src = plain(src)
            lab = os.path.join(C, "_lab_" + f)
            open(lab, "w").write(src)
            o = os.path.join(out, "%s_%s.o" % (name, f[:-4]))
            try: subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", lab, "-o", o])
            finally: os.remove(lab)
            objs.append(o)
        else:
            objs.append(os.path.join(C, f[:-4] + ".o"))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "lib_%s.so" % name)] + objs)
    print("built", name)
build("poolld", {"imgpool.hip"}); build("meanld", {"imgproxy16.hip"}); build("bothld", {"imgpool.hip", "imgproxy16.hip"})

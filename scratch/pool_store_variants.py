"""Lab builds: cache policy of k_img_pool's write-out of the per-tile partials (shipped: sc1 = write-through).
python scratch/pool_store_variants.py -> scratch/lab/lib_st_{plain,nt,sc1nt,sc0sc1}.so ; bash scratch/pool_var_ab.sh real st_plain ..."""
import os, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(R, "proxytransformation_amd", "csrc")
src = open(os.path.join(C, "imgpool.hip")).read()
needle = 'asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + i), "v"(v) : "memory");'
assert needle in src
out = os.path.join(R, "scratch", "lab"); os.makedirs(out, exist_ok=True)
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function".split()
objs = [os.path.join(C, f) for f in sorted(os.listdir(C)) if f.endswith(".o") and f != "imgpool.o"]
for name, mod in (("plain", ""), ("nt", " nt"), ("sc1nt", " sc1 nt"), ("sc0sc1", " sc0 sc1")):
    lab = os.path.join(C, "_imgpool_st.hip")
    open(lab, "w").write(src.replace(needle, needle.replace(" off sc1", " off" + mod)))
    try:
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", lab, "-o", os.path.join(out, "imgpool_st.o")])
    finally:
        os.remove(lab)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "lib_st_%s.so" % name),
                           os.path.join(out, "imgpool_st.o")] + objs)
    print("built", name)

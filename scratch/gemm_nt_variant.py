"""Lab build: the o-projection GEMM (k_gemm32 AMODE 1) reads the pooling partials with streaming (nt) loads.
python scratch/gemm_nt_variant.py -> scratch/lab/lib_ont.so ; bash scratch/ab_interleaved.sh real ont 32"""
import os, re, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(R, "proxytransformation_amd", "csrc")
src = open(os.path.join(C, "gemm.hip")).read()
n = 0
def rep(m):
    global n
    n += 1
    return "ld_nt4(%s)" % m.group(1)
src2 = re.sub(r"\*reinterpret_cast<const float4 \*>\((pr\.p[ge] \+ [^)]*\)?[^;]*?)\);", lambda m: rep(m) + ";", src)
helper = '''
typedef float f32x4nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_nt4(const float *p)
{
    const f32x4nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt *>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
'''
i = src2.index("template <int SK, int AMODE, bool CHAIN = false>")
src2 = src2[:i] + helper + src2[i:]
print("replaced", n, "loads")
out = os.path.join(R, "scratch", "lab"); os.makedirs(out, exist_ok=True)
lab = os.path.join(C, "_gemm_ont.hip")
open(lab, "w").write(src2)
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function".split()
try:
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", lab, "-o", os.path.join(out, "gemm_ont.o")])
finally:
    os.remove(lab)
objs = [os.path.join(C, f) for f in sorted(os.listdir(C)) if f.endswith(".o") and f != "gemm.o"]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "lib_ont.so"), os.path.join(out, "gemm_ont.o")] + objs)
print("built ont")

"""Lab builds of the library with parts of k_img_pool removed (timing only -- the results are wrong): which phase of the
co-resident work-group slows a unit's requests?   python scratch/pool_variants.py  ->  scratch/lab/lib_{nostore,nostage3,loadsonly}.so
then on the GPU box:  bash scratch/bench_ab.sh real nostore nostage3 loadsonly   (the line's roofline object times the kernel)"""
import os, re, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(R, "proxytransformation_amd", "csrc")
src = open(os.path.join(C, "imgpool.hip")).read()
i_k = src.index("void k_img_pool(PoolArgs a)")
i_e = src.index("bool img_pool_supported(int dt")
head, k, tail = src[:i_k], src[i_k:i_e], src[i_e:]

def variant(name):
    # every variant keeps the shipped residency -- 4 waves per SIMD = two work-groups per CU: without the attribute the variants
    # without stage 3 compile to 138-142 VGPRs (ONE work-group per CU) and the loads-only one to 79 (three)
    t = k.replace("void k_img_pool(PoolArgs a)", "__attribute__((amdgpu_waves_per_eu(4, 4))) void k_img_pool(PoolArgs a)", 1)
    if name in ("nostore", "loadsonly", "nostore_nostage3"):
        # no write-through stores of G, no E / ML stores
        t = t.replace('asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + i), "v"(v) : "memory");',
                      'if (v[0] == 1.2345e-30f) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + i), "v"(v) : "memory");')
        t = t.replace("erow[t] = t <= hw ? e[u] : 0.0f;", "if (e[u] == 1.2345e-30f) erow[t] = t <= hw ? e[u] : 0.0f;")
        t = t.replace("            ml[2 * T] = m;\n            ml[2 * T + 1] = l;\n            if (T == 0) { ml[4] = s0; erow[0] = 1.0f; }",
                      "            if (m == 1.2345e-30f) { ml[2 * T] = m; ml[2 * T + 1] = l; if (T == 0) { ml[4] = s0; erow[0] = 1.0f; } }")
    if name in ("nostage3", "loadsonly", "nostore_nostage3"):
        a = t.index("    // ---- 3. weighted sums over the tile's pixels from the registers.")
        b = t.index("    __syncthreads();\n    {\n        const float *G = partial;")
        keep = "    { unsigned x_ = 0;\n"
        keep += "".join("      x_ ^= L[%d][%d][0] ^ L[%d][%d][3];\n" % (kb, i, kb, i) for kb in range(2) for i in range(8))
        keep += "      if (x_ == 0x12345678u) partial[threadIdx.x] = 1.0f; }\n"
        t = t[:a] + keep + t[b:]
    if name == "s12_notrail":
        # stages 1 + 2, no stage 3, no stores, and WITHOUT the small loads behind the tile loads (positional terms, q / k0 of token 0)
        t = variant("nostore_nostage3")
        t = t.replace("ev[u] = p < hw ? wim[(size_t)wid * a.KT1 + in_dim + 1 + p] : 0.0f;", "ev[u] = p < hw ? 0.25f * (float)p : 0.0f;")
        t = t.replace("if (lane < hd) { sq = qv[lane]; sk = qv[a.C + lane]; }", "if (lane < hd) { sq = (float)lane; sk = 0.5f; (void)qv; }")
        return t
    if name == "s1only":
        # stage 1 stays (scores into the LDS slices, barrier); no stage 2 / 3, no stores
        a = t.index("    // ---- 2. wave = head: sum the eight channel slices")
        b = t.index("    __syncthreads();\n    {\n        const float *G = partial;")
        keep = "    { unsigned x_ = 0;\n"
        keep += "".join("      x_ ^= L[%d][%d][0] ^ L[%d][%d][3];\n" % (kb, i, kb, i) for kb in range(2) for i in range(8))
        keep += "      if (x_ == 0x12345678u) partial[threadIdx.x] = 1.0f; }\n"
        t = t[:a] + keep + t[b:]
        t = t.replace('asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + i), "v"(v) : "memory");',
                      'if (v[0] == 1.2345e-30f) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + i), "v"(v) : "memory");')
    if name == "loadsonly":
        # no stage 1 / 2 either: everything between the first barrier and the fold above
        a = t.index("    // ---- 1. scores of this wave's 64 channels")
        b = t.index("    { unsigned x_ = 0;")
        t = t[:a] + t[b:]
    return t

out = os.path.join(R, "scratch", "lab"); os.makedirs(out, exist_ok=True)
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable".split()
objs = [os.path.join(C, f) for f in sorted(os.listdir(C)) if f.endswith(".o") and f != "imgpool.o"]
import sys
for name in (sys.argv[1:] or ["nostore", "nostage3", "loadsonly", "nostore_nostage3"]):
    lab = os.path.join(C, "_imgpool_%s.hip" % name)
    open(lab, "w").write(head + variant(name) + tail)
    try:
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", lab, "-o", os.path.join(out, "imgpool_%s.o" % name)])
    finally:
        os.remove(lab)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "lib_%s.so" % name),
                           os.path.join(out, "imgpool_%s.o" % name)] + objs)
    print("built", name)

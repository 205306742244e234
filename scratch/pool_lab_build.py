"""Lab build of the library with s_memrealtime (100 MHz, chip-wide) phase stamps in k_img_pool: a COPY of
csrc/imgpool.hip gets the stamps (never the product), the other objects are the product's.
    python scratch/pool_lab_build.py   ->  scratch/lab/lib_poolstamp.so      (then scratch/pool_stamp.py on the GPU box)"""
import os, re, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(R, "proxytransformation_amd", "csrc")
src = open(os.path.join(C, "imgpool.hip")).read()
head = '''
__device__ unsigned long long *g_pool_dbg = nullptr;
#define STAMP(k) do { if (threadIdx.x == 0 && g_pool_dbg) g_pool_dbg[(size_t)dbg_unit * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define STAMP_ID() do { if (threadIdx.x == 0 && g_pool_dbg) { g_pool_dbg[(size_t)dbg_unit * 16 + 12] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); g_pool_dbg[(size_t)dbg_unit * 16 + 13] = blockIdx.x; } } while (0)
'''
src = src.replace("struct PoolArgs {", head + "struct PoolArgs {", 1)
i_e = src.index("bool img_pool_supported(int dt")
fresh, pers, tail = src[:i_e], "", src[i_e:]

def stamp_barriers(txt, first):
    k = [first]
    def rep(m):
        k[0] += 1
        return "__syncthreads(); STAMP(%d);" % min(k[0], 15)
    return re.sub(r"__syncthreads\(\);", rep, txt)

# 0 start, 1 weights in LDS, 2 scores, 3 soft-max, 4 weighted sums, 5 end  (slots 12 / 13: HW_ID | XCC_ID << 32, blockIdx)
a0 = "const int slab = im * 2 + T;"
i_k = fresh.index("__global__ __launch_bounds__(512) void k_img_pool(PoolArgs a)")
fk = fresh[i_k:]
fk = fk.replace(a0, a0 + " const int dbg_unit = blockIdx.x; STAMP(0); STAMP_ID();", 1)
fk = stamp_barriers(fk, 0)
fk = fk.replace("""            asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + i), "v"(v) : "memory");
        }
    }
}""", """            asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + i), "v"(v) : "memory");
        }
    }
    STAMP(5);
}""", 1)
fresh = fresh[:i_k] + fk
tail += '''
extern "C" int ptx_lab_pool_dbg(void *buf)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(ptx::g_pool_dbg), &buf, sizeof(buf));
}
'''
out = os.path.join(R, "scratch", "lab")
os.makedirs(out, exist_ok=True)
lab = os.path.join(C, "_imgpool_lab.hip")      # next to common.h for the includes
open(lab, "w").write(fresh + pers + tail)
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function".split()
try:
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", lab, "-o", os.path.join(out, "imgpool_lab.o")])
finally:
    os.remove(lab)
objs = [os.path.join(C, f) for f in sorted(os.listdir(C)) if f.endswith(".o") and f != "imgpool.o"]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "lib_poolstamp.so"),
                       os.path.join(out, "imgpool_lab.o")] + objs)
print("built", os.path.join(out, "lib_poolstamp.so"))

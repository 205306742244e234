// Three-way bf16 operand split shared by the "x3" matrix-core kernels (gemm.hip k_gemm64x, fattn.hip k_proxy_attn).
// An fp32 operand is written x = x1 + x2 + x3 with each part a bf16 (round to nearest: |x - x1| <= 2^-9 |x|,
// |x - x1 - x2| <= 2^-18 |x|, the third part carries the rest), and a product of two operands is the six bf16
// matrix-core products x1 y1 + (x1 y2 + x2 y1) + (x1 y3 + x2 y2 + x3 y1), accumulated in fp32; the dropped terms
// are <= 2^-25 |x y|, below the fp32 rounding of the sum.  Six v_mfma_f32_32x32x16_bf16 do the work of eight
// v_mfma_f32_32x32x2_f32 in 3/8 of the matrix-pipe cycles.
#pragma once
#include <hip/hip_runtime.h>

namespace ptx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split3_pair(float x, float y, unsigned &p1, unsigned &p2, unsigned &p3)
{
    // (forcing the residuals into v_pk_add_f32 -- inline assembly; the compiler takes a two-wide subtraction of freshly built
    //  halves apart again -- halves these instructions and made k_proxy_attn 8 % SLOWER: r03, scratch/README.md)
    const f32x2 v = {x, y};
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 r1 = {x - __uint_as_float(p1 << 16), y - __uint_as_float(p1 & 0xffff0000u)};
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
    const f32x2 r2 = {r1[0] - __uint_as_float(p2 << 16), r1[1] - __uint_as_float(p2 & 0xffff0000u)};
    p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}
// one float4 (four consecutive k) -> the three planes' 8-byte pieces at dst, dst + plane, dst + 2 plane (bytes)
__device__ __forceinline__ void stash_split3(char *dst, int plane, const float4 &v)
{
    unsigned a1, a2, a3, b1, b2, b3;
    split3_pair(v.x, v.y, a1, a2, a3);
    split3_pair(v.z, v.w, b1, b2, b3);
    *reinterpret_cast<uint2 *>(dst) = make_uint2(a1, b1);
    *reinterpret_cast<uint2 *>(dst + plane) = make_uint2(a2, b2);
    *reinterpret_cast<uint2 *>(dst + 2 * plane) = make_uint2(a3, b3);
}
// eight consecutive k of one matrix row / column -> the lane's three MFMA fragments (parts 1, 2, 3)
__device__ __forceinline__ void split3_frag(const float (&x)[8], bf16x8 (&f)[3])
{
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 q1, q2, q3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned a, b, c;
        split3_pair(x[2 * i], x[2 * i + 1], a, b, c);
        q1[i] = a; q2[i] = b; q3[i] = c;
    }
    f[0] = __builtin_bit_cast(bf16x8, q1); f[1] = __builtin_bit_cast(bf16x8, q2); f[2] = __builtin_bit_cast(bf16x8, q3);
}
// acc += A B^T for split operands: the six products, small terms first
__device__ __forceinline__ f32x16 mfma_split6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 acc)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}
// NP = 3: the exact split above; NP = 1: plain bf16 operands (round to nearest), one product per 16 k -- the opt-in
// reduced-precision mode of the proxy blocks (PtxForwardOpts::compute_dtype = 1, what autocast gives the reference's
// linears under --amp)
template <int NP>
__device__ __forceinline__ void stash_parts(char *dst, int plane, const float4 &v)
{
    if (NP == 3) { stash_split3(dst, plane, v); return; }
    const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
    *reinterpret_cast<uint2 *>(dst) = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2)),
                                                 __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2)));
}
template <int NP>
__device__ __forceinline__ void frag_parts(const float (&x)[8], bf16x8 (&f)[3])
{
    if (NP == 3) { split3_frag(x, f); return; }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 v = {x[2 * i], x[2 * i + 1]};
        q[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    }
    f[0] = __builtin_bit_cast(bf16x8, q);
}
template <int NP>
__device__ __forceinline__ f32x16 mfma_parts(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 acc)
{
    if (NP == 3) return mfma_split6(a, b, acc);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

// The same on raw registers (four dwords = eight bf16).  Kernels that carry fragments around loops keep them in this type: a
// bf16x8 value that crosses a conditional reload is taken apart into sixteen-bit halves by the compiler and put together
// again with a v_lshrrev + v_perm per dword (48 VALU instructions per 32 x 32 x 32 step of k_proxy_attn, measured r03).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NP>
__device__ __forceinline__ void frag_parts(const float (&x)[8], u32x4 (&f)[3])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (NP == 3) {
            unsigned a, b, c;
            split3_pair(x[2 * i], x[2 * i + 1], a, b, c);
            f[0][i] = a; f[1][i] = b; f[2][i] = c;
        } else {
            const f32x2 v = {x[2 * i], x[2 * i + 1]};
            f[0][i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
        }
    }
}
template <int NP>
__device__ __forceinline__ f32x16 mfma_parts(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 acc)
{
    auto bf = [](const u32x4 &q) { return __builtin_bit_cast(bf16x8, q); };
    if (NP == 3) {                                  // the six products of mfma_split6, small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a[2]), bf(b[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a[1]), bf(b[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a[0]), bf(b[2]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a[1]), bf(b[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a[0]), bf(b[1]), acc, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a[0]), bf(b[0]), acc, 0, 0, 0);
}

// LDS rows of the split tiles: 32 bf16 = 64 bytes, unpadded; the four 16-byte pieces of row r sit at piece ^ ((r >> 2) & 3),
// which makes both the 8-byte stash writes (four rows x 64 B per half wave) and the 16-byte fragment reads (sixteen rows,
// one piece each) bank-conflict free
constexpr int XROW = 64;
__device__ __forceinline__ int xswz(int row, int piece) { return (piece ^ ((row >> 2) & 3)) * 16; }

}  // namespace ptx

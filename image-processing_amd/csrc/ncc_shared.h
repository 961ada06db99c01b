// Pieces of the dense-NCC kernels shared by the 8-bit (ncc_kernels.hip) and 16-bit
// (ncc16_kernels.hip) translation units.
#pragma once

#include "mis_common.h"

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

struct NccArgs {
    int W, H, pitch;
    int dmin, dmax, nchunk;
    int row0, row1;      // output rows [row0,row1), already clipped to [r, H-r)
    int band_rows;       // rows per workgroup
    int out_pitch;       // elements per output row (== W)
    int nstrips, nbands; // fast path: work decomposition (strips of columns, bands of rows)
};

// 16-bit pairs (ncc16_kernels.hip); a: geometry filled in by the caller except pitch/nstrips/nbands
int mis_ncc16_run(mis_ctx* ctx, int window, NccArgs a, float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2);

// key of a score for the batched argmax: upper 26 bits of the float, 63 - lane below
__device__ __forceinline__ float ncc_make_key(float sc, uint32_t tag)
{
    return __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, sc) & 0xFFFFFFC0u) | tag);
}

#define NCC_DPP_MAX(dst, src, mods) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 " mods : "+v"(dst) : "v"(src))

// ---- transposing max network, fed one column at a time ----
// (a lane swap needs two wait states behind the VALU write of its operands, none in front of the
// VALU read of its results)
__device__ __forceinline__ float ncc_merge32(float a, float b)
{
    float r;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %1, %2\n\tv_max_f32 %0, %1, %2" : "=v"(r), "+v"(a), "+v"(b));
    return r;                     // lanes 0-31: column of a, lanes 32-63: column of b
}
__device__ __forceinline__ float ncc_merge16(float a, float b)
{
    float r;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %1, %2\n\tv_max_f32 %0, %1, %2" : "=v"(r), "+v"(a), "+v"(b));
    return r;                     // rows 0,2 from a, rows 1,3 from b
}
// Both DPP steps of a merge sit in ONE asm statement: its inputs exist before the statement starts,
// so the single s_nop 1 (VALU write -> DPP read needs two wait states) covers both reads, and the
// early-clobber output keeps the result register apart from the inputs.
__device__ __forceinline__ float ncc_merge8(float a, float b)
{
    float o;
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc"
                 : "=&v"(o) : "v"(a), "v"(b));
    return o;                     // lanes 0-7 of every row from a, lanes 8-15 from b
}
__device__ __forceinline__ float ncc_merge4(float a, float b)
{
    float o;
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_max_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa"
                 : "=&v"(o) : "v"(a), "v"(b));
    return o;                     // banks 0,2 from a, banks 1,3 from b
}

// Scores of four columns -> keys -> both DPP levels, as one hand-scheduled statement: every DPP read
// sits at least two instructions behind the write of its source, which leaves a single wait state
// to pad.  The score registers are overwritten with their keys.
__device__ __forceinline__ float ncc_quad4(float s0, float s1, float s2, float s3, uint32_t tag)
{
    float o0, o1, p;
    asm volatile("v_and_or_b32 %3, %3, %7, %8\n\t"
                 "v_and_or_b32 %4, %4, %7, %8\n\t"
                 "v_and_or_b32 %5, %5, %7, %8\n\t"
                 "v_and_or_b32 %6, %6, %7, %8\n\t"
                 "v_max_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_max_f32_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_max_f32_dpp %2, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "s_nop 0\n\t"
                 "v_max_f32_dpp %2, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa"
                 : "=&v"(o0), "=&v"(o1), "=&v"(p), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)
                 : "s"(0xFFFFFFC0u), "v"(tag));
    return p;
}

// Same, and the four keys are handed back (the second-peak pass masks and reduces them again).
__device__ __forceinline__ float ncc_quad4_keep(float& s0, float& s1, float& s2, float& s3, uint32_t tag)
{
    float o0, o1, p;
    asm volatile("v_and_or_b32 %3, %3, %7, %8\n\t"
                 "v_and_or_b32 %4, %4, %7, %8\n\t"
                 "v_and_or_b32 %5, %5, %7, %8\n\t"
                 "v_and_or_b32 %6, %6, %7, %8\n\t"
                 "v_max_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_max_f32_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_max_f32_dpp %2, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "s_nop 0\n\t"
                 "v_max_f32_dpp %2, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa"
                 : "=&v"(o0), "=&v"(o1), "=&v"(p), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)
                 : "s"(0xFFFFFFC0u), "v"(tag));
    return p;
}
// The two DPP levels on four ready-made keys.
__device__ __forceinline__ float ncc_quad4_keys(float k0, float k1, float k2, float k3)
{
    float o0, o1, p;
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_max_f32_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_max_f32_dpp %2, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "s_nop 0\n\t"
                 "v_max_f32_dpp %2, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa"
                 : "=&v"(o0), "=&v"(o1), "=&v"(p) : "v"(k0), "v"(k1), "v"(k2), "v"(k3));
    return p;
}

// Keys of 16 columns (lane = candidate) -> one register in which every 4-lane bank holds the 4
// partial maxima of one column.  Columns are pushed one at a time (jj = 0..15, a compile-time
// constant after unrolling); pairs are merged as soon as both halves exist, so at most five
// partial registers are alive.
struct NccTree {
    float t0, t1, t2, t3, q;
    __device__ __forceinline__ void push(int jj, float k)
    {
        // the levels with the most merges use the cheapest steps (two DPP maxima); the lane swaps,
        // which cost half as much again, are left for the last two levels
        if ((jj & 1) == 0) { t0 = k; return; }
        const float m1 = ncc_merge8(t0, k);
        if ((jj & 2) == 0) { t1 = m1; return; }
        const float m2 = ncc_merge4(t1, m1);
        if ((jj & 4) == 0) { t2 = m2; return; }
        const float m3 = ncc_merge32(t2, m2);
        if ((jj & 8) == 0) { t3 = m3; return; }
        q = ncc_merge16(t3, m3);
    }
    // four columns at once (jj = 0, 4, 8, 12), from their scores
    __device__ __forceinline__ void push4(int jj, float s0, float s1, float s2, float s3, uint32_t tag)
    {
        push_m2(jj, ncc_quad4(s0, s1, s2, s3, tag));
    }
    // same; the scores are replaced by their keys
    __device__ __forceinline__ void push4_keep(int jj, float& s0, float& s1, float& s2, float& s3, uint32_t tag)
    {
        push_m2(jj, ncc_quad4_keep(s0, s1, s2, s3, tag));
    }
    // four columns at once from ready-made keys
    __device__ __forceinline__ void push4_keys(int jj, float k0, float k1, float k2, float k3)
    {
        push_m2(jj, ncc_quad4_keys(k0, k1, k2, k3));
    }
    __device__ __forceinline__ void push_m2(int jj, float m2)
    {
        if ((jj & 4) == 0) { t2 = m2; return; }
        const float m3 = ncc_merge32(t2, m2);
        if ((jj & 8) == 0) { t3 = m3; return; }
        q = ncc_merge16(t3, m3);
    }
};

// ---- second peak: the best key of a column among the lanes further than one lane from its winner ----
// In a finished 16-column register (NccTree::q) column jj occupies the four lanes 4 * NCC_BANK16[jj] ..
// + 3, each holding a partial maximum (the network's fixed permutation; ncc_reduce_quads later picks
// one lane per column).  ncc_quad_full makes all four lanes hold the column maximum.
__device__ constexpr int NCC_BANK16[16] = { 0, 2, 1, 3, 8, 10, 9, 11, 4, 6, 5, 7, 12, 14, 13, 15 };
__device__ __forceinline__ float ncc_quad_full(float q)
{
    float t = q;
    NCC_DPP_MAX(t, q, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    float u = t;
    NCC_DPP_MAX(u, t, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    return u;
}
// key with the lanes winner-1 .. winner+1 replaced by `low`; qfull: ncc_quad_full of the register that
// holds column jj (a compile-time constant).  The winner's key is read into a scalar register, the
// three-lane mask is scalar arithmetic, and one v_cndmask per column applies it.
template <int JJ>
__device__ __forceinline__ float ncc_mask_near_winner(float key, float qfull, float low)
{
    const uint32_t kw = (uint32_t)__builtin_amdgcn_readlane(__builtin_bit_cast(int, qfull), 4 * NCC_BANK16[JJ]);
    const uint32_t lw = ~kw & 63u;                                   // winner's lane: the tag is 63 - lane
    const unsigned long long m = lw == 0 ? 3ull : (7ull << (lw - 1));
    float out;
    asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(out) : "v"(key), "v"(low), "s"(m));
    return out;
}


// four such registers -> one register with 64 column maxima (one per lane)
__device__ __forceinline__ float ncc_reduce_quads(float q0, float q1, float q2, float q3, int lane)
{
    float t0 = q0, t1 = q1, t2 = q2, t3 = q3;
    NCC_DPP_MAX(t0, q0, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    NCC_DPP_MAX(t1, q1, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    NCC_DPP_MAX(t2, q2, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    NCC_DPP_MAX(t3, q3, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    float e0 = (lane & 2) ? t1 : t0;
    float e1 = (lane & 2) ? t3 : t2;
    float u0 = e0, u1 = e1;
    NCC_DPP_MAX(u0, e0, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    NCC_DPP_MAX(u1, e1, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    return (lane & 1) ? u1 : u0;
}


// second pass over a block of 16 columns: keys kk[0..15], qfull = ncc_quad_full of their finished register
__device__ __forceinline__ void ncc_second_pass16(const float* kk, float qfull, float low, NccTree& tr2)
{
    tr2.push4_keys(0, ncc_mask_near_winner<0>(kk[0], qfull, low), ncc_mask_near_winner<1>(kk[1], qfull, low),
                   ncc_mask_near_winner<2>(kk[2], qfull, low), ncc_mask_near_winner<3>(kk[3], qfull, low));
    tr2.push4_keys(4, ncc_mask_near_winner<4>(kk[4], qfull, low), ncc_mask_near_winner<5>(kk[5], qfull, low),
                   ncc_mask_near_winner<6>(kk[6], qfull, low), ncc_mask_near_winner<7>(kk[7], qfull, low));
    tr2.push4_keys(8, ncc_mask_near_winner<8>(kk[8], qfull, low), ncc_mask_near_winner<9>(kk[9], qfull, low),
                   ncc_mask_near_winner<10>(kk[10], qfull, low), ncc_mask_near_winner<11>(kk[11], qfull, low));
    tr2.push4_keys(12, ncc_mask_near_winner<12>(kk[12], qfull, low), ncc_mask_near_winner<13>(kk[13], qfull, low),
                   ncc_mask_near_winner<14>(kk[14], qfull, low), ncc_mask_near_winner<15>(kk[15], qfull, low));
}

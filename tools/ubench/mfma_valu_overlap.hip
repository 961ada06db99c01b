// Do matrix-core products and vector instructions of different wavefronts on one SIMD overlap?
// Workgroups of 512 threads (8 wavefronts, 2 per SIMD); the role of a wavefront depends on the mode:
//   mode 0: all wavefronts issue MFMAs            mode 1: all issue vector ops
//   mode 2: per SIMD one wavefront MFMAs, one vector ops (waves w and w+4 share a SIMD)
//   mode 3: every wavefront interleaves 1 MFMA with NV vector ops (one stream)
// Prints the time per mode; overlap shows as mode 2 ~ max(mode 0, mode 1) / 2-wave share.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ITER 2000

template <int KIND>   // vector op kind: 0 v_mul_f32, 1 v_and_or_b32, 2 v_max3_f32, 3 v_pk_mul_f32
__device__ __forceinline__ void valu8(float (&r)[8], float s)
{
    if (KIND == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
    } else if (KIND == 1) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_and_or_b32 %0, %0, %1, 5" : "+v"(r[i]) : "v"(s));
    } else if (KIND == 2) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(s));
    } else {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2* p = (f2*)r;
        f2 ss = { s, s };
#pragma unroll
        for (int i = 0; i < 4; i++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(ss));
#pragma unroll
        for (int i = 0; i < 4; i++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(ss));
    }
}

template <int ACCA>
__device__ __forceinline__ void mfma16(f32x16& acc, half8 va, half8 vb)
{
    if (ACCA) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(va), "v"(vb));
    else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, vb, acc, 0, 0, 0);
}

template <int KIND, int F32, int NVX>   // NVX: groups of 8 vector instructions per product
__global__ __launch_bounds__(512) void k(int mode, float* out, float s)
{
    const int wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int a = 0; a < 4; a++) for (int i = 0; i < 16; i++) acc[a][i] = 0.f;
    half8 va, vb;
    for (int i = 0; i < 8; i++) { va[i] = (_Float16)(threadIdx.x & 7); vb[i] = (_Float16)(i); }
    float r[8];
    for (int i = 0; i < 8; i++) r[i] = s + i;
    const bool do_mfma = (mode == 0) || (mode == 2 && wave < 4) || mode == 3;
    const bool do_valu = (mode == 1) || (mode == 2 && wave >= 4) || mode == 3;
    if (mode == 3) {
        for (int it = 0; it < ITER; it++) {
#pragma unroll
            for (int a = 0; a < 4; a++) {
                if (F32 == 1) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(s, r[0], acc[a], 0, 0, 0);
                else mfma16<(F32 == 2)>(acc[a], va, vb);
#pragma unroll
                for (int q = 0; q < NVX; q++) valu8<KIND>(r, s);
            }
        }
    } else if (do_mfma) {
        for (int it = 0; it < ITER; it++) {
#pragma unroll
            for (int a = 0; a < 4; a++) {
                if (F32 == 1) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(s, r[0], acc[a], 0, 0, 0);
                else mfma16<(F32 == 2)>(acc[a], va, vb);
            }
        }
    } else if (do_valu) {
        for (int it = 0; it < ITER; it++) {
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int q = 0; q < NVX; q++) valu8<KIND>(r, s);
        }
    }
    float t = 0.f;
    for (int a = 0; a < 4; a++) for (int i = 0; i < 16; i++) t += acc[a][i];
    for (int i = 0; i < 8; i++) t += r[i];
    out[blockIdx.x * 512 + threadIdx.x] = t;
}

template <int KIND, int F32, int NVX>
static void run(const char* name)
{
    float* out; hipMalloc(&out, 256 * 8 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[4];
    for (int mode = 0; mode < 4; mode++) {
        hipLaunchKernelGGL((k<KIND, F32, NVX>), dim3(256), dim3(512), 0, 0, mode, out, 1.0001f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((k<KIND, F32, NVX>), dim3(256), dim3(512), 0, 0, mode, out, 1.0001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[mode], e0, e1); ms[mode] /= 3;
    }
    // per SIMD: mode 0: 2 waves x ITER x 4 MFMA; mode 1: 2 waves x ITER x 32 valu; mode 2: 1 wave each; mode 3: 2 waves of both
    const double n_m = (double)ITER * 4, n_v = (double)ITER * 32 * NVX;
    printf("%-28s mfma-only %.3f ms (%.1f ns/MFMA/SIMD)  valu-only %.3f ms (%.2f ns/op/SIMD)  split waves %.3f ms  one stream (1 MFMA : %d ops) %.3f ms\n",
           name, ms[0], ms[0] * 1e6 / (2 * n_m), ms[1], ms[1] * 1e6 / (2 * n_v), ms[2], 8 * NVX, ms[3]);
    hipFree(out);
}

int main()
{
    run<0, 0, 1>("f16 MFMA + v_mul_f32");
    run<1, 0, 1>("f16 MFMA + v_and_or_b32");
    run<2, 0, 1>("f16 MFMA + v_max3_f32");
    run<3, 0, 1>("f16 MFMA + v_pk_mul_f32");
    run<0, 1, 1>("f32 MFMA + v_mul_f32");
    run<1, 1, 1>("f32 MFMA + v_and_or_b32");
    run<2, 1, 1>("f32 MFMA + v_max3_f32");
    run<0, 0, 1>("f16 MFMA + v_mul_f32 (again)");
    // accumulators in AGPRs: no difference
    run<0, 2, 1>("f16 MFMA in AGPRs + v_mul_f32");
    run<1, 2, 1>("f16 MFMA in AGPRs + v_and_or_b32");
    // 24 vector instructions per product (the argmax kernel's ratio): v_mul_f32 / v_fma_f32 do not overlap with the products at all,
    // v_and_or_b32 / v_max3_f32 hide about two thirds of them
    run<0, 0, 3>("f16 MFMA + 3x v_mul_f32");
    run<1, 0, 3>("f16 MFMA + 3x v_and_or_b32");
    run<2, 0, 3>("f16 MFMA + 3x v_max3_f32");
    return 0;
}

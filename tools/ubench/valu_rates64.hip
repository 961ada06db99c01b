// Throughput of the 64-bit and integer vector instructions the 16-bit dense-NCC kernel can be built
// from, measured like valu_rates.hip: one wave per workgroup, W waves per SIMD, ITER x 16 independent
// copies of the instruction per wave.  Usage: valu_rates64 [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define ITER 16000
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(64) void k(long long* out, float seed)
{
    float r[16]; float b = seed + threadIdx.x, c = seed * 0.5f;
    double dr[16]; double db = b, dc = c;
#pragma unroll
    for (int i = 0; i < 16; i++) { r[i] = seed + i; dr[i] = seed + i; }
    double dsg = 1.5; unsigned sm = 77u;
    for (int it = 0; it < ITER; it++) {
#define F64(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(dr[i]) : "v"(db), "v"(dc));
#define F64S(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(dr[i]) : "s"(dsg), "v"(dc));
#define A64(i) asm volatile("v_add_f64 %0, %1, %0" : "+v"(dr[i]) : "v"(db));
#define M64(i) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(dr[i]) : "v"(db));
#define CVT(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(dr[i]) : "v"(b));
#define CVTD(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r[i]) : "v"(db));
#define MADU(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(dr[i]) : "v"(b), "v"(c) : "vcc");
#define MADI(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(dr[i]) : "v"(b), "v"(c) : "vcc");
#define MADIS(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(dr[i]) : "s"(sm), "v"(c) : "vcc");
#define D2I(i) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(r[i]) : "v"(b), "v"(c));
#define D2U(i) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(r[i]) : "v"(b), "v"(c));
#define M24(i) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(r[i]) : "v"(b), "v"(c));
#define MLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define MHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define ADC(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %1, vcc" : "+v"(r[i]), "+v"(b), "+v"(r[(i + 8) & 15]) : : "vcc");
#define LA64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(dr[i]) : "v"(db));
#define FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(b), "v"(c));
#define MADU32(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(r[i]) : "v"(b), "v"(c));
#define CVTU(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(dr[i]) : "v"(b));
#define CVTF(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(r[i]) : "v"(b));
        if (OP == 0) { REP16(F64) } if (OP == 1) { REP16(F64S) } if (OP == 2) { REP16(A64) } if (OP == 3) { REP16(M64) }
        if (OP == 4) { REP16(CVT) } if (OP == 5) { REP16(CVTD) } if (OP == 6) { REP16(MADU) } if (OP == 7) { REP16(MADI) }
        if (OP == 8) { REP16(MADIS) } if (OP == 9) { REP16(D2I) } if (OP == 10) { REP16(D2U) } if (OP == 11) { REP16(M24) }
        if (OP == 12) { REP16(MLO) } if (OP == 13) { REP16(MHI) } if (OP == 14) { REP16(ADC) } if (OP == 15) { REP16(LA64) }
        if (OP == 16) { REP16(FMA) } if (OP == 17) { REP16(MADU32) } if (OP == 18) { REP16(CVTU) } if (OP == 19) { REP16(CVTF) }
    }
    float s = 0; double ds = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { s += r[i]; ds += dr[i]; }
    if (s + (float)ds + b == 12345.678f) out[0] = 1;
}

template <int OP> void run(const char* name, int wps, long long* d, int per = 1)
{
    const int nblk = 256 * 4 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(nblk), dim3(64), 0, 0, d, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(nblk), dim3(64), 0, 0, d, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)ITER * 16 * per;
    printf("%-26s %.2f ns per wave64 instruction per SIMD (kernel %.3f ms, %d waves/SIMD)\n", name, ms * 1e6 / (n * wps), ms, wps);
}

int main(int argc, char** argv)
{
    const int wps = argc > 1 ? atoi(argv[1]) : 8;
    long long* d; hipMalloc(&d, 64); hipMemset(d, 0, 8);
    for (int rep = 0; rep < 3; rep++) {
    printf("-- pass %d (the first pass runs while the clock ramps up)\n", rep);
    run<16>("v_fma_f32", wps, d);
    run<0>("v_fma_f64 vgpr", wps, d); run<1>("v_fma_f64 sgpr", wps, d); run<2>("v_add_f64", wps, d); run<3>("v_mul_f64", wps, d);
    run<4>("v_cvt_f64_i32", wps, d); run<18>("v_cvt_f64_u32", wps, d); run<5>("v_cvt_f32_f64", wps, d); run<19>("v_cvt_f32_i32", wps, d);
    run<6>("v_mad_u64_u32", wps, d); run<7>("v_mad_i64_i32 vgpr", wps, d); run<8>("v_mad_i64_i32 sgpr", wps, d);
    run<9>("v_dot2_i32_i16", wps, d); run<10>("v_dot2_u32_u16", wps, d); run<11>("v_mad_i32_i24", wps, d); run<17>("v_mad_u32_u24", wps, d);
    run<12>("v_mul_lo_u32", wps, d); run<13>("v_mul_hi_u32", wps, d);
    run<14>("v_add_co + v_addc_co", wps, d, 2); run<15>("v_lshl_add_u64", wps, d);
    }
    return 0;
}

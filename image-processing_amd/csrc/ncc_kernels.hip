// Dense windowed zero-mean NCC over a horizontal disparity range, gfx950.
//
// What is computed (per left pixel, per disparity d):
//   score(d) = cov(L window, R window shifted by d) / sqrt(var L * var R)
// with the first maximum over d, its two neighbours (parabolic subpixel) and optionally the best
// score further than 1 px away (second peak).  This is the dense mode the reference leaves disabled
// (image3.cpp:775-811) and names as a TODO (image3.cpp:866-870); sign and tie-break follow
// image3.cpp:696-699 and image3.cpp:486-501.
//
// Two argmax paths feed one refinement kernel.  Windows up to 21 x 21 run on the matrix cores: ncc_mfma.h, where
// the window sums of 32 x 32 tiles of (left column, right column) pairs slide down the rows as MFMA accumulators.
// Wider windows run the vector kernels of this file:
//
// Mapping onto the machine (vector kernels)
//   lane        <-> disparity (64 consecutive d per wavefront, one "chunk")
//   wavefront   <-> one chunk of disparities for a strip of WT pixel columns, walking down a band of rows
//   workgroup   <-> one wavefront; wavefronts never synchronise with each other
// so every window sum is a running sum held in one lane's registers: the column sums of the
// product image slide down the rows (one v_dot2c_f32_f16 per column, row and disparity: it adds
// the entering row's product and subtracts the leaving row's), the window sum slides along the
// strip.  The left-image operand of every multiply is the same for all lanes (scalar loads); the
// right-image operand sits in LDS, where lane d reads element x-d.
//
// Exactness: pixels are shifted by 128 into [-128,127] (NCC does not change); every product,
// column sum and window sum is an integer below 2^24 and therefore exact in f32.
#include "ncc_ws.h"
#include "ncc_mfma.h"          // NCC_MF_MAX_R

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <cstdlib>


// side stream of the current call (created on first use), ordered behind everything enqueued on the main stream so far
int ncc_fork(mis_ctx* ctx, NccWs* w)
{
    if (w->forked) return MIS_OK;
    if (!w->side) {
        // Normal priority, like the main stream.  A high-priority side stream that waits for an event of a
        // normal-priority stream can stall for good once a process holds more queues than the device has
        // hardware queues (the waiting queue keeps its slot and the queue it waits for is never scheduled).
        MIS_HIP(ctx, hipStreamCreateWithFlags(&w->side, hipStreamNonBlocking));
        MIS_HIP(ctx, hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming));
        MIS_HIP(ctx, hipEventCreateWithFlags(&w->ev_tables, hipEventDisableTiming));
        MIS_HIP(ctx, hipEventCreateWithFlags(&w->ev_join, hipEventDisableTiming));
        MIS_HIP(ctx, hipEventCreateWithFlags(&w->ev_main, hipEventDisableTiming));
    }
    MIS_HIP(ctx, hipEventRecord(w->ev_fork, ctx->stream));
    MIS_HIP(ctx, hipStreamWaitEvent(w->side, w->ev_fork, 0));
    w->forked = true;
    return MIS_OK;
}
// main stream waits for the side stream's work of this call
int ncc_join(mis_ctx* ctx, NccWs* w)
{
    if (!w->forked) return MIS_OK;
    w->forked = false;
    MIS_HIP(ctx, hipEventRecord(w->ev_join, w->side));
    MIS_HIP(ctx, hipStreamWaitEvent(ctx->stream, w->ev_join, 0));
    return MIS_OK;
}

// end of a call that built the tables (side stream joined): calls that reuse them on another stream wait for this point
static int ncc_mark_built(mis_ctx* ctx, NccWs* w)
{
    if (!w->building) return MIS_OK;
    w->building = false;
    if (!w->ev_built) MIS_HIP(ctx, hipEventCreateWithFlags(&w->ev_built, hipEventDisableTiming));
    MIS_HIP(ctx, hipEventRecord(w->ev_built, ctx->stream));
    w->built_stream = ctx->stream;
    return MIS_OK;
}

void mis_ncc_release(mis_ctx* ctx)
{
    NccWs* w = ctx->ncc;
    if (!w) return;
    void* bufs[] = { w->Lpack, w->negc, w->negg, w->iL, w->Rpack, w->Rsr, w->Rir, w->VL, w->VR, w->keys };
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (w->side) {
        (void)hipStreamSynchronize(w->side);
        (void)hipEventDestroy(w->ev_fork); (void)hipEventDestroy(w->ev_tables); (void)hipEventDestroy(w->ev_join); (void)hipEventDestroy(w->ev_main);
        (void)hipStreamDestroy(w->side);
    }
    if (w->ev_built) (void)hipEventDestroy(w->ev_built);
    delete w;
    ctx->ncc = nullptr;
}

// ---------------------------------------------------------------------------------------------
// Pre-pass 1: window statistics and the packed row pairs, one pass over the raster.
// A workgroup owns 256 columns and a band of rows.  It walks down the band (plus r rows above and
// below): each row is staged in LDS as one word per pixel, (p'^2 << 13) | (p' + 128) with p' = p - 128,
// so that one integer add per tap sums p' and p'^2 together (31 taps: the low field stays below 2^13,
// the high one below 2^19).  Every thread sums its column's horizontal window, keeps the last 2r+1 such
// row sums in an LDS ring (a thread only touches its own ring column) and slides the vertical window
// over them.
// ---------------------------------------------------------------------------------------------
#define NCC_STATS_BAND 64
template <typename T, bool LEFT, int R>
__global__ __launch_bounds__(256) void ncc_tables_kernel(const T* __restrict__ img, size_t stride_px,
                                                         int W, int H, int pitch, int band0,
                                                         uint32_t* __restrict__ pack,
                                                         float* __restrict__ o0,   // LEFT: negc   RIGHT: S_R
                                                         float* __restrict__ o1,   // LEFT: negg   RIGHT: iR
                                                         float* __restrict__ o2)   // LEFT: iL
{
    constexpr int r = R, Wn = 2 * R + 1;
    constexpr long long N = (long long)Wn * Wn;
    __shared__ uint32_t rowbuf[2][256 + 32];
    __shared__ uint32_t ring[Wn][256];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * 256, x = x0 + tid;
    const int y0 = (band0 + blockIdx.y) * NCC_STATS_BAND;      // the launch covers bands band0 .. band0 + gridDim.y - 1
    const int y1 = min(y0 + NCC_STATS_BAND, H);
    const bool colok = (x >= r && x < W - r);
    uint32_t s1 = 0, s2 = 0;            // vertical sums of the ring, rows yy-2r .. yy: of (p' + 128) and of p'^2
    int slot = 0;
    // Rows are fetched into registers two iterations ahead and published to LDS one iteration ahead,
    // after the arithmetic.  Every load of an iteration is issued unconditionally at its top (coordinates
    // clamped into the raster, the value replaced afterwards where the pixel does not exist): a load
    // inside a conditional block is waited for at the end of that block, which turns the walk down the
    // band into one exposed memory round trip per row.
    const int Wm = W - 1, Hm = H - 1;
    // the fetched value stays as loaded until it is used: arithmetic on it here would be scheduled right behind
    // the load and wait for it
    auto raw = [&](int yy, int xs) -> int {
        return (int)img[(size_t)min(max(yy, 0), Hm) * stride_px + min(max(xs, 0), Wm)];
    };
    auto inside = [&](int yy, int xs) -> bool { return yy >= 0 && yy < H && xs >= 0 && xs < W; };
    auto word = [&](int v) -> uint32_t { return ((uint32_t)__mul24(v, v) << 13) | (uint32_t)(v + 128); };
    const int xa = x0 - r + tid, xb = x0 - r + 256 + (tid & 31);   // staged columns of this thread (xb: halo, threads 0..31)
    const bool halo = tid < 2 * r;
    struct Row { int n0, n1, pa, pb; };
    // fetch(yy): the staged pixels of row yy + 1 and, for the packed pair written at iteration yy, the own
    // pixel of row yy - r and the one Wn rows above it
    auto fetch = [&](int yy, int z) -> Row { return Row{ raw(yy + 1, xa + z), raw(yy + 1, xb + z), raw(yy - r, x + z), raw(yy - r - Wn, x + z) }; };
    const int ya = y0 - r, yz = y1 + r;
    {
        const int v0 = raw(ya, xa), v1 = raw(ya, xb);
        rowbuf[0][tid] = word(inside(ya, xa) ? v0 - 128 : 0);
        if (tid < 32) rowbuf[0][256 + tid] = word((inside(ya, xb) && halo) ? v1 - 128 : 0);
    }
    Row f0 = fetch(ya, 0), f1 = fetch(ya + 1, 0);
    __syncthreads();
    auto step = [&](int yy, Row& cur) {             // cur: fetched two iterations ago, refilled here for iteration yy + 2
        const int buf = (yy - ya) & 1;
        // everything that depends on the fetched bytes is derived first, so that the refill can land in
        // the same registers and nothing waits for it before the iteration after next
        uint32_t w0 = word(inside(yy + 1, xa) ? cur.n0 - 128 : 0);
        uint32_t w1 = word((inside(yy + 1, xb) && halo) ? cur.n1 - 128 : 0);
        float pa = (x < W) ? (float)(cur.pa - 128) : 0.f;
        float pb = (x < W && yy - r - Wn >= 0) ? (float)(cur.pb - 128) : 0.f;
        int z = 0;                      // the refill's addresses hang on this statement: it is issued after the lines above
        asm volatile("" : "+v"(z), "+v"(w0), "+v"(w1), "+v"(pa), "+v"(pb));
        cur = fetch(yy + 2, z);
        uint32_t hs = 0;
#pragma unroll
        for (int i = 0; i < Wn; i++) hs += rowbuf[buf][tid + i];
        if (yy - ya >= Wn) {            // the ring is full: drop the row that leaves the window
            const uint32_t old = ring[slot][tid];
            s1 -= old & 0x1FFFu;
            s2 -= old >> 13;
        }
        ring[slot][tid] = hs;
        slot = (slot + 1 == Wn) ? 0 : slot + 1;
        s1 += hs & 0x1FFFu; s2 += hs >> 13;

        const int y = yy - r;           // the window centred at y is complete now
        if (y >= y0 && y < y1 && x < pitch) {
            const size_t o = (size_t)y * pitch + x;
            // packed pair: this row and the row Wn above
            half2_t h;
            h.x = (_Float16)pa;
            h.y = (_Float16)(LEFT ? -pb : pb);
            pack[o] = __builtin_bit_cast(uint32_t, h);

            const bool rowok = (y >= r && y < H - r);
            float v0 = 0.f, v1 = LEFT ? 0.f : (colok ? 0.f : __builtin_nanf("")), v2 = 0.f;
            if (colok && rowok) {
                const int t1 = (int)s1 - 128 * (int)N;                          // sum of p'
                const long long var = N * (long long)s2 - (long long)t1 * t1;   // N * sum (p-mean)^2, exact
                // 1/sigma to float accuracy (a few ulp): scores only need 1e-7 relative
                const float inv = var > 0 ? sqrtf((float)N / (float)var) : 0.f;
                if (LEFT) {
                    // c: the integer nearest to t1/N (N is odd, so there are no ties; float division is
                    // accurate enough to pick it, and the remainder below is exact whatever c is)
                    const int cc = (int)__builtin_rintf((float)t1 / (float)N);
                    const int rem = t1 - (int)N * cc;
                    v0 = -(float)cc;
                    v1 = -((float)rem / (float)N);
                    v2 = inv;
                } else {
                    v0 = (float)t1;
                    v1 = inv;
                }
            }
            if (LEFT) {
                // left-image tables are stored shifted left by r columns (entry x-r describes the window
                // centred at x), so that a strip's first output column x0 = r + 64*strip lands on a
                // 256-byte boundary for the scalar loads of the correlation kernels
                if (x >= r) { o0[o - r] = v0; o1[o - r] = v1; o2[o - r] = v2; }
            } else {
                o0[o] = v0;
                o1[o] = v1;
            }
        }
        if (yy + 1 < yz) {
            rowbuf[buf ^ 1][tid] = w0;
            if (tid < 32) rowbuf[buf ^ 1][256 + tid] = w1;
        }
        __syncthreads();                // the next row is staged, this row's buffer may be overwritten
    };
    for (int yy = ya; yy < yz; yy += 2) {
        step(yy, f0);
        step(yy + 1, f1);               // unconditional (a conditional refill costs register copies that wait for the loads); with an odd row count the last call writes nothing
    }
}

// ---------------------------------------------------------------------------------------------
// Pre-pass 3: four vertically adjacent pixels per dword, shifted to int8, for v_dot4_i32_i8.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ncc_vpack_kernel(const T* __restrict__ img, size_t stride_px, int W, int H,
                                                        int pitch, int q0, uint32_t* __restrict__ out)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int q = q0 + blockIdx.y;
    if (x >= pitch) return;
    // the four rows are requested together (clamped coordinates) and tested afterwards: a load inside a condition is
    // waited for on the spot, four memory round trips in a row
    const int xc = min(x, W - 1);
    int p[4];
#pragma unroll
    for (int k = 0; k < 4; k++) p[k] = (int)img[(size_t)min(4 * q + k, H - 1) * stride_px + xc];
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (x < W && 4 * q + k < H) v |= (uint32_t)((p[k] - 128) & 0xFF) << (8 * k);
    out[(size_t)q * pitch + x] = v;
}

// ---------------------------------------------------------------------------------------------
// Integer argmax first, exact scores of the winner and its two neighbours afterwards.
//
// Argmax kernel: lane <-> disparity, and the per-column reduction over the 64 lanes is batched.
// Every score is turned into a key (low 6 mantissa bits replaced by 63 - lane,
// so a float maximum also picks the lowest disparity among scores that agree in their upper 26
// bits); 64 key registers (one per strip column) go through a transposing max network
// (v_permlane32_swap / v_permlane16_swap, then DPP row and quad steps with write masks) that ends
// with ONE register holding the 64 column maxima, about 2 vector instructions per column instead
// of a dependent 6-step reduction per column.  The chunk wavefronts of a workgroup then merge
// their keys through LDS.
// ---------------------------------------------------------------------------------------------
// ---- scheduling fences for the hand-pipelined loops below ----
// The loops load one group of columns ahead of the arithmetic: wait for everything issued so far
// (LDS and scalar loads share one counter), issue the next group's loads, then compute.  The
// compiler keeps that order only if it is told to: NCC_WAIT() is a wait it knows about followed by
// a scheduling barrier; ncc_pin() is an empty asm that reads and writes a value, so work on that
// value cannot move across it.
#define NCC_WAIT()                                      \
    do {                                                \
        __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */ \
        __builtin_amdgcn_sched_barrier(0);              \
    } while (0)
__device__ __forceinline__ void ncc_pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void ncc_pin_s(uint32_t& v) { asm volatile("" : "+s"(v)); }
__device__ __forceinline__ void ncc_pin_s(float& v) { asm volatile("" : "+s"(v)); }
// (the compiler pads a run of asm statements with a wait state, so the pins of a group come first)
__device__ __forceinline__ void ncc_pin_b4(float* sv, float* iv, float* cv, float* gv)
{
    asm volatile("" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(iv[0]), "+v"(iv[1]), "+v"(iv[2]), "+v"(iv[3]),
                      "+s"(cv[0]), "+s"(cv[1]), "+s"(cv[2]), "+s"(cv[3]), "+s"(gv[0]), "+s"(gv[1]), "+s"(gv[2]), "+s"(gv[3]));
}
typedef float float2_t __attribute__((ext_vector_type(2)));

// Argmax kernel: ONE wavefront per workgroup = one chunk of 64 disparities for one strip of WT
// columns and one band of rows; wavefronts never synchronise with each other.
// LDS image of the current right-image row, private to the wavefront: 3 arrays (packed pair, S_R,
// 1/sigma_R), each kept in 4 copies shifted by 0..3 elements so that every lane finds its x-d run
// 16-byte aligned in one of them (ds_read_b128); LENP % 64 == 16 staggers the copies over the
// banks.  Output: one key per pixel and chunk (upper 26 bits of the best score, 63 - lane below).
template <int R, int WT>
struct NccWaveCfg {
    static constexpr int NC = WT + 2 * R;
    static constexpr int NCP = (NC + 3) & ~3;
    static constexpr int SEG = NCP + 64 + 4;                       // staged right-image columns
    static constexpr int LENP = ((SEG + 7 + 47) / 64) * 64 + 16;   // >= SEG + 7, == 16 (mod 64)
};

// WANT2: a second key plane per chunk, oKeys2: the best key among the lanes further than one lane
// from the column's winner (the refinement builds the second-peak output from both planes).  The keys
// of 16 columns stay in registers until their winners are known, so this variant runs with one
// wavefront per SIMD less and the short pipeline stages.
template <int R, int WT, int MW, bool WANT2>
__global__ __launch_bounds__(64, MW) void ncc_argmax_kernel(
    NccArgs a,
    const uint32_t* __restrict__ gLpack, const float* __restrict__ gNegc, const float* __restrict__ gNegg,
    const uint32_t* __restrict__ gRpack, const float* __restrict__ gRsr, const float* __restrict__ gRir,
    float* __restrict__ oKeys, float* __restrict__ oKeys2)
{
    static_assert(WT == 64 || WT == 48 || WT == 32, "strip width");
    using Cfg = NccWaveCfg<R, WT>;
    constexpr int WN = 2 * R + 1;
    constexpr int NCP = Cfg::NCP, SEG = Cfg::SEG, LENP = Cfg::LENP;
    constexpr int GA = (MW >= 5 || WANT2) ? 8 : 16, GB = (MW >= 5 || WANT2) ? 4 : 8;      // columns per software-pipeline stage
    __shared__ __attribute__((aligned(16))) float sArr[12 * LENP];   // [3 arrays][4 copies][LENP]
    // Copy c of an array holds element i at i + ((4 - c) & 3): a lane whose run starts at element e
    // reads copy e & 3 from a 16-byte aligned position.  The 16 lanes that one ds_read_b128 cycle
    // serves use all four copies; copies 1..3 read 4 words further in than copy 0, so they start
    // 4 words early (LENP % 64 == 16 then puts the four copies on disjoint bank quarters).
#define NCC_COPY(arr, c) (((arr) * 4 + (c)) * LENP - ((c) ? 4 : 0))

    const int lane = threadIdx.x;
    // Workgroups are dealt to the 8 XCDs round-robin (ids b and b+8 share an L2).  Remap the id so
    // that every XCD works through a contiguous run of work items, ordered chunk-fastest: the
    // wavefronts resident on one XCD then read the same left-image rows and overlapping
    // right-image columns (an affinity for speed only, nothing depends on it).
    int chunk, strip, band;
    {
        const int total = a.nstrips * a.nbands * a.nchunk;
        const int id = blockIdx.x, xcd = id & 7, local = id >> 3;
        const int q = total >> 3, rem = total & 7;
        const int wid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
        chunk = wid % a.nchunk;
        strip = (wid / a.nchunk) % a.nstrips;
        band = wid / (a.nchunk * a.nstrips);
    }
    const int x0 = R + strip * WT;
    const int yb0 = a.row0 + band * a.band_rows;
    const int yb1 = min(yb0 + a.band_rows, a.row1);
    if (yb0 >= yb1) return;
    const int nrows = yb1 - yb0;
    const int T = 2 * R + nrows;

    const int dtop = a.dmin + chunk * 64 + 63;      // largest disparity of this chunk
    const int d = dtop - 63 + lane;
    const int loff = 63 - lane;                     // LDS element of (column k, this lane) is k + loff
    const int xr0 = x0 - R - dtop;                  // raster column of LDS element 0
    // lanes past dmax get a huge negative FINITE bias: -inf would turn into a signalling-NaN bit
    // pattern once the lane tag is inserted, and v_max_f32 propagates signalling NaNs
    const float bias = (d <= a.dmax) ? 0.f : -3.0e38f;
    const uint32_t tag = 63u - (uint32_t)lane;
    const float2_t bias2 = { bias, bias };

    const int cA = loff & 3;
    const int offB = loff + R, cB = offB & 3;

    // which strip column ends up in this lane of the reduced register: run the network on column ids
    int mycol;
    {
        float qi[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            NccTree tr;
#pragma unroll
            for (int jj = 0; jj < 16; jj++) tr.push(jj, (float)(b * 16 + jj));
            qi[b] = tr.q;
        }
        mycol = (int)ncc_reduce_quads(qi[0], qi[1], qi[2], qi[3], lane);
    }

    float C[NCP];
#pragma unroll
    for (int k = 0; k < NCP; k++) C[k] = 0.f;
    const int P = a.pitch;

    // right-image row -> LDS (this wavefront's columns only); element i <-> raster column xr0 + i
    constexpr int NSTG = (SEG + 63) / 64;      // staged elements per lane
    uint32_t nbp[NSTG]; float nsr[NSTG], nir[NSTG];
    // fetch only issues the loads (uniform row base + 32-bit lane offset); nothing touches the loaded
    // registers before publish, a whole row of arithmetic later, so the memory latency is hidden
    auto fetch = [&](int ynew, int yout, bool stats) {
        const uint32_t* rp = gRpack + (size_t)ynew * P;
        const float* sp = gRsr + (size_t)(stats ? yout : 0) * P;
        const float* ip = gRir + (size_t)(stats ? yout : 0) * P;
#pragma unroll
        for (int q = 0; q < NSTG; q++) {
            const int i = lane + 64 * q;
            const unsigned xr = (unsigned)(xr0 + i);
            const bool in = (xr < (unsigned)a.W && i < SEG);       // also false for negative columns
            nbp[q] = in ? rp[xr] : 0u;
            nsr[q] = (stats && in) ? sp[xr] : 0.f;
            nir[q] = stats ? (in ? ip[xr] : __builtin_nanf("")) : 0.f;
        }
    };
    auto publish = [&](bool stats, bool warm) {
#pragma unroll
        for (int q = 0; q < NSTG; q++) {
            const int i = lane + 64 * q;
            if (i < SEG) {
                const uint32_t bp = warm ? (nbp[q] & 0x0000FFFFu) : nbp[q];   // warm-up rows: nothing leaves the window yet
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int pos = i + ((4 - c) & 3);
                    sArr[NCC_COPY(0, c) + pos] = __builtin_bit_cast(float, bp);
                    if (stats) {
                        sArr[NCC_COPY(1, c) + pos] = nsr[q];
                        sArr[NCC_COPY(2, c) + pos] = nir[q];
                    }
                }
            }
        }
    };
    fetch(yb0 - R, yb0 - 2 * R, false);
    publish(false, true);

    // Operand buffers of the two software-pipelined loops (two sets each).  The first group of a
    // loop is requested before the previous loop's last group is consumed (part B's inside part A,
    // the next row's part A right after the row image is rewritten), so no wait sees a fresh request.
    constexpr int NGA = (NCP + GA - 1) / GA, NGB = WT / GB;
    float bv[2][GA];
    uint32_t av[2][GA];
    float sv[2][GB], iv[2][GB], cv[2][GB], gv[2][GB];
    const float4* bp4 = (const float4*)(sArr + NCC_COPY(0, cA) + loff + ((4 - cA) & 3));
    const float4* sr4 = (const float4*)(sArr + NCC_COPY(1, cB) + offB + ((4 - cB) & 3));
    const float4* ir4 = (const float4*)(sArr + NCC_COPY(2, cB) + offB + ((4 - cB) & 3));
    auto loadA = [&](const uint32_t* lrow, int g, int set) {
#pragma unroll
        for (int v = 0; v < GA / 4; v++) {
            if (GA * g + 4 * v < NCP) {
                const float4 f = bp4[(GA / 4) * g + v];
                bv[set][4 * v] = f.x; bv[set][4 * v + 1] = f.y; bv[set][4 * v + 2] = f.z; bv[set][4 * v + 3] = f.w;
            }
        }
#pragma unroll
        for (int e = 0; e < GA; e++)
            if (GA * g + e < NCP) av[set][e] = lrow[GA * g + e];
    };
    auto loadB = [&](const float* ncrow, const float* ngrow, int g, int set) {
#pragma unroll
        for (int v = 0; v < GB / 4; v++) {
            const float4 f = sr4[(GB / 4) * g + v], h = ir4[(GB / 4) * g + v];
            sv[set][4 * v] = f.x; sv[set][4 * v + 1] = f.y; sv[set][4 * v + 2] = f.z; sv[set][4 * v + 3] = f.w;
            iv[set][4 * v] = h.x; iv[set][4 * v + 1] = h.y; iv[set][4 * v + 2] = h.z; iv[set][4 * v + 3] = h.w;
        }
#pragma unroll
        for (int e = 0; e < GB; e++) { cv[set][e] = ncrow[GB * g + e]; gv[set][e] = ngrow[GB * g + e]; }
    };
    loadA(gLpack + (size_t)(yb0 - R) * P + (x0 - R), 0, 0);

    for (int t = 0; t < T; t++) {
        const int ynew = yb0 - R + t;
        const bool score = (t >= 2 * R);
        const int yout = ynew - R;
        const bool have_next = (t + 1 < T);
        if (have_next) fetch(ynew + 1, yout + 1, t + 1 >= 2 * R);
        const float* ncrow = gNegc + (size_t)(score ? yout : 0) * P + (x0 - R);
        const float* ngrow = gNegg + (size_t)(score ? yout : 0) * P + (x0 - R);

        // ---- column sums: add the entering row, drop the leaving row (one v_dot2c each) ----
        {
            const uint32_t* lrow = gLpack + (size_t)ynew * P + (x0 - R);
#pragma unroll
            for (int g = 0; g < NGA; g++) {
                const int set = g & 1;
                NCC_WAIT();
                if (g + 1 < NGA) loadA(lrow, g + 1, set ^ 1);
                else loadB(ncrow, ngrow, 0, 0);          // unconditional: warm-up rows request row 0's tables and drop them
                __builtin_amdgcn_sched_barrier(0);
                static_assert(GB % 4 == 0, "part B pins its values four columns at a time");
#pragma unroll
                for (int e = 0; e < GA; e++) { ncc_pin(bv[set][e]); ncc_pin_s(av[set][e]); }
#pragma unroll
                for (int e = 0; e < GA; e++) {
                    const int k = GA * g + e;
                    if (k < NCP) {
                        C[k] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, av[set][e]),
                                                      __builtin_bit_cast(half2_t, bv[set][e]), C[k], false);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        float red = 0.f, red2 = 0.f;
        if (score) {
            float S = 0.f;
#pragma unroll
            for (int k = 0; k < 2 * R; k++) S += C[k];
            float q[4] = { -3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f };
            float q2[4] = { -3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f };
            NccTree tr, tr2;
            float kk[WANT2 ? 16 : 1];                           // keys of the current block of 16 columns
#pragma unroll
            for (int g = 0; g < NGB; g++) {
                const int set = g & 1;
                NCC_WAIT();
                if (g + 1 < NGB) loadB(ncrow, ngrow, g + 1, set ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                // The packed fmas of the group's column pairs are written interleaved (all tt, then all
                // cov, then all scores): a dependent v_pk_fma_f32 right behind its producer costs a wait
                // state, an independent one does not.
                constexpr int NP = GB / 2;
                float2_t S2[NP], s2[NP], i2[NP], c2[NP], g2[NP], tt[NP], cov[NP], sc[NP];
#pragma unroll
                for (int e = 0; e < GB; e += 4) ncc_pin_b4(&sv[set][e], &iv[set][e], &cv[set][e], &gv[set][e]);
#pragma unroll
                for (int pq = 0; pq < NP; pq++) {
                    const int e = 2 * pq, j = GB * g + e;
                    s2[pq] = float2_t{ sv[set][e], sv[set][e + 1] }; i2[pq] = float2_t{ iv[set][e], iv[set][e + 1] };
                    c2[pq] = float2_t{ cv[set][e], cv[set][e + 1] }; g2[pq] = float2_t{ gv[set][e], gv[set][e + 1] };
                    const float Sa = S + C[j + 2 * R];
                    const float Sb = (Sa - C[j]) + C[j + 1 + 2 * R];
                    S = Sb - C[j + 1];
                    S2[pq] = float2_t{ Sa, Sb };
                }
#pragma unroll
                for (int pq = 0; pq < NP; pq++) tt[pq] = __builtin_elementwise_fma(c2[pq], s2[pq], S2[pq]);     // S - c*S_R, exact
#pragma unroll
                for (int pq = 0; pq < NP; pq++) cov[pq] = __builtin_elementwise_fma(g2[pq], s2[pq], tt[pq]);    // - (rem/N)*S_R
#pragma unroll
                for (int pq = 0; pq < NP; pq++) sc[pq] = __builtin_elementwise_fma(cov[pq], i2[pq], bias2);
                static_assert(GB % 4 == 0, "columns are keyed four at a time");
#pragma unroll
                for (int pq = 0; pq < NP; pq += 2) {
                    const int jj = (GB * g + 2 * pq) & 15;
                    if constexpr (WANT2) {
                        float k0 = sc[pq].x, k1 = sc[pq].y, k2 = sc[pq + 1].x, k3 = sc[pq + 1].y;
                        tr.push4_keep(jj, k0, k1, k2, k3, tag);
                        kk[jj] = k0; kk[jj + 1] = k1; kk[jj + 2] = k2; kk[jj + 3] = k3;
                    } else {
                        tr.push4(jj, sc[pq].x, sc[pq].y, sc[pq + 1].x, sc[pq + 1].y, tag);
                    }
                }
                if (((GB * (g + 1)) & 15) == 0) {
                    q[(GB * g) >> 4] = tr.q;
                    if constexpr (WANT2) {
                        // the 16 winners are known now: mask the three lanes around each and reduce again
                        const float qf = ncc_quad_full(tr.q);
                        const float low = -3.0e38f;
                        ncc_second_pass16(kk, qf, low, tr2);
                        q2[(GB * g) >> 4] = tr2.q;
                    }
                }
                ncc_pin(S);
                __builtin_amdgcn_sched_barrier(0);
            }
            red = ncc_reduce_quads(q[0], q[1], q[2], q[3], lane);
            if constexpr (WANT2) red2 = ncc_reduce_quads(q2[0], q2[1], q2[2], q2[3], lane);
        }

        // the LDS image is single-buffered: every read of this row was issued above, LDS executes a
        // wavefront's operations in order, so the next row may be written now
        if (have_next) publish(t + 1 >= 2 * R, t + 1 < WN);
        // first group of the next row (unconditional, so that the scalar operands stay in SGPRs; after
        // the last row it re-reads the current one and is dropped)
        loadA(gLpack + (size_t)(have_next ? ynew + 1 : ynew) * P + (x0 - R), 0, 0);
        // The key store comes after publish: vector memory operations complete in order, so waiting
        // for the prefetched row (inside publish) would otherwise also wait for this store's
        // acknowledgement, a full memory round trip per row.
        if (score) {
            const int x = x0 + mycol;
            if (mycol < WT && x < a.W - R) {
                oKeys[((size_t)chunk * a.H + yout) * a.out_pitch + x] = red;
                if constexpr (WANT2) oKeys2[((size_t)chunk * a.H + yout) * a.out_pitch + x] = red2;
            }
        }
    }
}

// Tail kernel: the few disparities left over when the search range is not a multiple of 64
// (+-256 is 513 disparities: eight full chunks and one disparity).  A whole chunk wavefront for them
// would idle 63 of its 64 lanes, so they are scored the other way round: lane <-> column, a loop
// over the nt <= NCC_TAIL_MAX disparities.  Column sums slide down a band of rows in registers
// (the same v_dot2c on the packed row pairs), the horizontal window sum goes through LDS.  Output:
// one more key plane in the format of the argmax kernel (plane `chunk`, tags relative to it).
// The kernel runs beside the argmax kernel on a second stream.  Its workgroups are three wavefronts
// (192 - 2R output columns): a workgroup can only start on a CU with that many wave slots free at
// once, and while the argmax grid lasts slots free up one wavefront at a time.  Measured step times
// at 8192x8192, +-256: 64 threads 16.27 ms (fits anywhere, but a third of its lanes are halo),
// 128: 15.95, 192: 15.93, 320: 16.05 (mostly runs after the argmax kernel); 16.40 with the tail in
// front of the argmax kernel on one stream.
#define NCC_TAIL_MAX 4
#ifndef NCC_TAIL_NT
#define NCC_TAIL_NT 192
#endif
#define NCC_TAIL_TW(RR) (NCC_TAIL_NT - 2 * (RR))
#define NCC_TAIL_RB 4        // rows whose operands are fetched together (one memory round trip per batch)
template <int R>
__global__ __launch_bounds__(NCC_TAIL_NT) void ncc_tail_kernel(
    NccArgs a, int d0, int nt, int chunk,
    const uint32_t* __restrict__ gLpack, const float* __restrict__ gNegc, const float* __restrict__ gNegg,
    const uint32_t* __restrict__ gRpack, const float* __restrict__ gRsr, const float* __restrict__ gRir,
    float* __restrict__ oKeys, float* __restrict__ oKeys2)      // oKeys2 (nullable): second-peak plane, as in the argmax kernel
{
    constexpr int WN = 2 * R + 1, TW = NCC_TAIL_TW(R), NCOL = TW + 2 * R, RB = NCC_TAIL_RB;
    __shared__ float sC[RB][NCC_TAIL_MAX][NCOL + 2];
    const int tid = threadIdx.x;
    const int x0 = R + blockIdx.x * TW;               // first output column of the workgroup
    const int xc = x0 - R + tid;                      // column whose sums this thread carries
    const bool carry = tid < NCOL && xc < a.W;
    const int yb0 = a.row0 + blockIdx.y * a.band_rows;
    const int yb1 = min(yb0 + a.band_rows, a.row1);
    if (yb0 >= yb1) return;
    const int T = 2 * R + (yb1 - yb0);
    const int P = a.pitch;
    const int x = x0 + tid;
    const bool outcol = tid < TW && x < a.W - R;
    float C[NCC_TAIL_MAX];
#pragma unroll
    for (int j = 0; j < NCC_TAIL_MAX; j++) C[j] = 0.f;

    // The walk down the band is latency bound (little arithmetic per row), so the operands of RB
    // rows are requested together and the rows are then processed back to back.
    for (int t0 = 0; t0 < T; t0 += RB) {
        uint32_t lp[RB], rp[RB][NCC_TAIL_MAX];
        float nc[RB], ng[RB], srv[RB][NCC_TAIL_MAX], irv[RB][NCC_TAIL_MAX];
#pragma unroll
        for (int b = 0; b < RB; b++) {
            const int t = t0 + b;
            const int ynew = min(yb0 - R + t, a.H - 1);            // rows past the band are fetched but not used
            const bool live = t < T;
            lp[b] = (live && carry) ? gLpack[(size_t)ynew * P + xc] : 0u;
#pragma unroll
            for (int j = 0; j < NCC_TAIL_MAX; j++) {
                const unsigned u = (unsigned)(xc - (d0 + j));
                rp[b][j] = (live && j < nt && carry && u < (unsigned)a.W) ? gRpack[(size_t)ynew * P + u] : 0u;
            }
            nc[b] = ng[b] = 0.f;
            const bool sc_row = live && t >= 2 * R && outcol;
            const int yout = ynew - R;
            if (sc_row) { nc[b] = gNegc[(size_t)yout * P + (x - R)]; ng[b] = gNegg[(size_t)yout * P + (x - R)]; }
#pragma unroll
            for (int j = 0; j < NCC_TAIL_MAX; j++) {
                const int d = d0 + j, u = x - d;
                const bool ok = (sc_row && j < nt && d <= a.dmax && u >= R && u < a.W - R);
                srv[b][j] = ok ? gRsr[(size_t)yout * P + u] : 0.f;
                irv[b][j] = ok ? gRir[(size_t)yout * P + u] : __builtin_nanf("");
            }
        }
        // column sums of the RB rows, published to LDS
#pragma unroll
        for (int b = 0; b < RB; b++) {
            const int t = t0 + b;
            const uint32_t keep = (t < WN) ? 0x0000FFFFu : 0xFFFFFFFFu;   // warm-up rows: nothing leaves the window yet
#pragma unroll
            for (int j = 0; j < NCC_TAIL_MAX; j++)
                if (j < nt) {
                    C[j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, lp[b]), __builtin_bit_cast(half2_t, rp[b][j] & keep), C[j], false);
                    if (tid < NCOL) sC[b][j][tid] = C[j];
                }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < RB; b++) {
            const int t = t0 + b;
            if (t < 2 * R || t >= T || !outcol) continue;
            const int yout = yb0 - R + t - R;
            float best = -3.0e38f;
            float keyj[NCC_TAIL_MAX];
            int jb = -9;
#pragma unroll
            for (int j = 0; j < NCC_TAIL_MAX; j++) {
                keyj[j] = -3.0e38f;
                if (j < nt) {
                    float S = 0.f;
#pragma unroll
                    for (int i = 0; i < WN; i++) S += sC[b][j][tid + i];       // exact: integers below 2^24
                    const float tt = __builtin_fmaf(nc[b], srv[b][j], S);
                    const float cov = __builtin_fmaf(ng[b], srv[b][j], tt);
                    const float sc = __builtin_fmaf(cov, irv[b][j], 0.f);      // NaN where this disparity is not valid
                    const float key = ncc_make_key(sc, 63u - (uint32_t)(d0 + j - (a.dmin + chunk * 64)));
                    if (key > best) { best = key; jb = j; }                    // NaN never wins
                    keyj[j] = key;
                }
            }
            oKeys[((size_t)chunk * a.H + yout) * a.out_pitch + x] = best;
            if (oKeys2) {                                                      // best key further than one disparity from the winner
                float second = -3.0e38f;
#pragma unroll
                for (int j = 0; j < NCC_TAIL_MAX; j++)
                    if (j < nt && (j < jb - 1 || j > jb + 1) && keyj[j] > second) second = keyj[j];
                oKeys2[((size_t)chunk * a.H + yout) * a.out_pitch + x] = second;
            }
        }
        __syncthreads();
    }
}

// Refinement: exact window correlation of the winner and its two neighbours from vertically packed
// int8 pixels (v_dot4_i32_i8: four rows per instruction), then the same float expression as the
// argmax kernel, the parabola vertex and the outputs.

// fills the frame border (pixels whose window does not fit) of rows [row0,row1)
__global__ __launch_bounds__(256) void ncc_border_kernel(float* disp, float* peak, int16_t* idisp, float* peak2,
                                                         int W, int H, int r, int row0, int row1)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = row0 + blockIdx.y;
    if (x >= W || y >= row1) return;
    const bool inside = (x >= r && x < W - r && y >= r && y < H - r);
    if (inside) return;
    const size_t o = (size_t)y * W + x;
    disp[o] = (float)MIS_INVALID_DISP;
    peak[o] = 0.f;
    if (idisp) idisp[o] = (int16_t)MIS_INVALID_DISP;
    if (peak2) peak2[o] = -1.f;
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
template <bool LEFT>
static void ncc_launch_tables(int r, dim3 grid, hipStream_t s, const uint8_t* img, size_t stride_px, int W, int H, int pitch,
                              int band0, uint32_t* pack, float* o0, float* o1, float* o2)
{
#define NCC_TCASE(RR) case RR: hipLaunchKernelGGL((ncc_tables_kernel<uint8_t, LEFT, RR>), grid, dim3(256), 0, s, img, stride_px, \
                                                  W, H, pitch, band0, pack, o0, o1, o2); break;
    switch (r) {
        NCC_TCASE(1) NCC_TCASE(2) NCC_TCASE(3) NCC_TCASE(4) NCC_TCASE(5) NCC_TCASE(6) NCC_TCASE(7) NCC_TCASE(8)
        NCC_TCASE(9) NCC_TCASE(10) NCC_TCASE(11) NCC_TCASE(12) NCC_TCASE(13) NCC_TCASE(14) NCC_TCASE(15)
        default: break;                 // ncc_run rejects other windows before anything is launched
    }
#undef NCC_TCASE
}

// Builds (or reuses) the per-pair tables for raster rows [need_lo, need_hi): the rows the correlation of
// the requested output rows reads.  The tables of rows built by earlier calls on the same pair are kept,
// so a frame computed strip by strip builds every band of rows once, and a rank that computes one strip
// of a frame builds that strip's rows only.
static int ncc_prepare(mis_ctx* ctx, int window, bool use_side, int need_lo, int need_hi)
{
    NccWs* w = ctx->ncc;
    const int W = ctx->width, H = ctx->height, r = window / 2;
    const int pitch = ((W + 64 + 63) / 64) * 64 + 64;   // room for one strip past the right edge
    const size_t npx = (size_t)pitch * H;
    if (w->cap_px < npx) {
        void** bufs[] = { (void**)&w->Lpack, (void**)&w->negc, (void**)&w->negg, (void**)&w->iL,
                          (void**)&w->Rpack, (void**)&w->Rsr, (void**)&w->Rir,
                          (void**)&w->VL, (void**)&w->VR };
        for (void** b : bufs) {
            if (*b) { MIS_HIP(ctx, hipFree(*b)); *b = nullptr; }
        }
        w->cap_px = 0;
        for (void** b : bufs) MIS_HIP(ctx, hipMalloc(b, npx * 4));
        w->cap_px = npx;
        w->pair_serial = 0;
    }
    // whole bands of rows
    need_lo = need_lo < 0 ? 0 : need_lo / NCC_STATS_BAND * NCC_STATS_BAND;
    need_hi = (need_hi + NCC_STATS_BAND - 1) / NCC_STATS_BAND * NCC_STATS_BAND;
    if (need_hi > H) need_hi = H;
    const bool same = (w->pair_serial == ctx->pair_serial && w->window == window && w->pitch == pitch && w->rows == H);
    int lo[2], hi[2], nparts = 0;                        // row ranges to build now
    if (same && need_lo >= w->built_lo && need_hi <= w->built_hi) {
        if (w->built_stream != ctx->stream && w->ev_built) MIS_HIP(ctx, hipStreamWaitEvent(ctx->stream, w->ev_built, 0));
        return MIS_OK;
    }
    if (same && need_lo <= w->built_hi && need_hi >= w->built_lo) {     // touches what exists: build what is missing on either side
        if (need_lo < w->built_lo) { lo[nparts] = need_lo; hi[nparts] = w->built_lo; nparts++; }
        if (need_hi > w->built_hi) { lo[nparts] = w->built_hi; hi[nparts] = need_hi; nparts++; }
        if (w->built_stream != ctx->stream && w->ev_built) MIS_HIP(ctx, hipStreamWaitEvent(ctx->stream, w->ev_built, 0));
        w->built_lo = need_lo < w->built_lo ? need_lo : w->built_lo;
        w->built_hi = need_hi > w->built_hi ? need_hi : w->built_hi;
    } else {
        lo[0] = need_lo; hi[0] = need_hi; nparts = 1;
        w->built_lo = need_lo; w->built_hi = need_hi;
    }
    w->building = true;
    w->pitch = pitch; w->rows = H;
    w->pair_serial = 0;               // the cache claims nothing until every launch below has been accepted
    dim3 block(256);
    hipStream_t s = ctx->stream, sv = ctx->stream;       // sv: side stream when there is one
    if (use_side) {
        int rc = ncc_fork(ctx, w);
        if (rc) return rc;
        sv = w->side;
    }
    {   // a pair still on its way from page-locked host rasters: the rows the table kernels read (whole bands of rows
        // plus the window's rows below them) must have arrived, on both streams (the side stream forked earlier)
        const int upto = need_hi + r + 4 < H ? need_hi + r + 4 : H;
        int rcu = mis_wait_upload(ctx, upto, s);
        if (!rcu && sv != s) rcu = mis_wait_upload(ctx, upto, sv);
        if (rcu) return rcu;
    }
    if (ctx->dtype != MIS_U8)
        return mis_fail(ctx, MIS_ERR_UNSUPPORTED, "dense NCC: only 8-bit rasters are supported by this build");
    const uint8_t* pl = (const uint8_t*)ctx->img[0];
    const uint8_t* pr = (const uint8_t*)ctx->img[1];
    // the two table kernels run side by side (each alone leaves wave slots idle); the main stream
    // then waits for the right image's tables only, not for the vertical packs queued behind them
    for (int k = 0; k < nparts; k++) {
        const int band0 = lo[k] / NCC_STATS_BAND;
        const dim3 sgrid((pitch + 255) / 256, (hi[k] - lo[k] + NCC_STATS_BAND - 1) / NCC_STATS_BAND);
        ncc_launch_tables<false>(r, sgrid, sv, pr, ctx->stride, W, H, pitch, band0, w->Rpack, w->Rsr, w->Rir, (float*)nullptr);
    }
    if (use_side) MIS_HIP(ctx, hipEventRecord(w->ev_tables, w->side));
    for (int k = 0; k < nparts; k++) {
        const int band0 = lo[k] / NCC_STATS_BAND;
        const dim3 sgrid((pitch + 255) / 256, (hi[k] - lo[k] + NCC_STATS_BAND - 1) / NCC_STATS_BAND);
        ncc_launch_tables<true>(r, sgrid, s, pl, ctx->stride, W, H, pitch, band0, w->Lpack, w->negc, w->negg, w->iL);
    }
    if (use_side) MIS_HIP(ctx, hipStreamWaitEvent(ctx->stream, w->ev_tables, 0));
    for (int k = 0; k < nparts; k++) {
        const int q0 = lo[k] / 4, q1 = (hi[k] + 3) / 4;  // bands are multiples of 4 rows, except the last one of the raster
        const dim3 vgrid((pitch + 255) / 256, q1 - q0);
        hipLaunchKernelGGL(ncc_vpack_kernel<uint8_t>, vgrid, block, 0, sv, pl, ctx->stride, W, H, pitch, q0, w->VL);
        hipLaunchKernelGGL(ncc_vpack_kernel<uint8_t>, vgrid, block, 0, sv, pr, ctx->stride, W, H, pitch, q0, w->VR);
    }
    MIS_HIP(ctx, hipGetLastError());
    w->pair_serial = ctx->pair_serial;
    w->window = window;
    return MIS_OK;
}

template <int R, int WT, int MW, bool WANT2>
static int ncc_launch_fast(mis_ctx* ctx, NccArgs a, float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2)
{
    NccWs* w = ctx->ncc;
    // a short remainder of the range (nd % 64 disparities) goes to the tail kernel instead of a
    // chunk wavefront that would idle most of its lanes
    const int nd = a.dmax - a.dmin + 1;
    const int rem = nd % 64, nfull = nd / 64;
    const bool tail = (rem > 0 && rem <= NCC_TAIL_MAX && nfull >= 1);
    const int nplanes = a.nchunk;                    // key planes the refinement merges
    const size_t plane_px = (size_t)a.H * a.W;
    const size_t keys_bytes = (size_t)nplanes * plane_px * sizeof(float) * (WANT2 ? 2 : 1);
    int rc = mis_reserve(ctx, (void**)&w->keys, &w->keys_bytes, keys_bytes);
    if (rc) return rc;
    float* keys2 = WANT2 ? w->keys + (size_t)nplanes * plane_px : nullptr;      // second-peak planes behind the first ones
    a.nstrips = (a.W - 2 * R + WT - 1) / WT;
    a.nbands = (a.row1 - a.row0 + a.band_rows - 1) / a.band_rows;
    if (tail) a.nchunk = nfull;
    mis_prof_mark(ctx);                            // tables done
    // the tail writes its own key plane, beside the argmax kernel (side stream, behind the tables)
    if (tail) {
        rc = ncc_fork(ctx, w);
        if (rc) return rc;
        MIS_HIP(ctx, hipEventRecord(w->ev_tables, ctx->stream));
        MIS_HIP(ctx, hipStreamWaitEvent(w->side, w->ev_tables, 0));
        NccArgs ta = a;                              // shorter bands: the tail walk is latency bound and wants more workgroups
        ta.band_rows = 64;
        ta.nbands = (a.row1 - a.row0 + ta.band_rows - 1) / ta.band_rows;
        hipLaunchKernelGGL(ncc_tail_kernel<R>, dim3((a.W - 2 * R + NCC_TAIL_TW(R) - 1) / NCC_TAIL_TW(R), ta.nbands), dim3(NCC_TAIL_NT), 0, w->side,
                           ta, a.dmin + nfull * 64, rem, nfull, w->Lpack, w->negc, w->negg, w->Rpack, w->Rsr, w->Rir, w->keys, keys2);
    }
    hipLaunchKernelGGL((ncc_argmax_kernel<R, WT, MW, WANT2>), dim3((unsigned)(a.nstrips * a.nbands * a.nchunk)), dim3(64), 0, ctx->stream,
                       a, w->Lpack, w->negc, w->negg, w->Rpack, w->Rsr, w->Rir, w->keys, keys2);
    mis_prof_mark(ctx);
    rc = ncc_join(ctx, w);                         // vertical packs, border, tail
    if (rc) return rc;
    rc = ncc_refine_launch(ctx, R, false, a, nplanes, keys2, d_disp, d_peak, d_idisp, d_peak2);
    if (rc) return rc;
    MIS_HIP(ctx, hipGetLastError());
    return MIS_OK;
}


static int ncc_run(mis_ctx* ctx, int window, int dmin, int dmax, int row0, int nrows,
                   float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2)
{
    if (!ctx) return MIS_ERR_ARG;
    if (!ctx->img[0]) return mis_fail(ctx, MIS_ERR_STATE, "dense NCC: no image pair bound");
    if (!d_disp || !d_peak) return mis_fail(ctx, MIS_ERR_ARG, "dense NCC: disp and peak outputs are required");
    if (window < 3 || window > 31 || !(window & 1)) return mis_fail(ctx, MIS_ERR_ARG, "dense NCC: window %d not odd in 3..31", window);
    if (dmax < dmin || dmax - dmin + 1 > 2048) return mis_fail(ctx, MIS_ERR_ARG, "dense NCC: bad disparity range [%d,%d]", dmin, dmax);
    if (dmin < -16000 || dmax > 16000) return mis_fail(ctx, MIS_ERR_ARG, "dense NCC: disparity out of the int16 output range");
    const int W = ctx->width, H = ctx->height, r = window / 2;
    if (row0 < 0 || nrows < 0 || row0 + nrows > H) return mis_fail(ctx, MIS_ERR_ARG, "dense NCC: rows [%d,%d) outside the raster", row0, row0 + nrows);
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    if (nrows == 0) return MIS_OK;

    mis_prof_mark(ctx);
    // the 8-bit path runs its small kernels on a second stream
    const bool fast = (ctx->dtype == MIS_U8 && W >= window && H >= window);
    if (fast && !ctx->ncc) ctx->ncc = new NccWs();
    {   // frame border
        hipStream_t sb = ctx->stream;
        if (fast) {
            int rcf = ncc_fork(ctx, ctx->ncc);
            if (rcf) return rcf;
            sb = ctx->ncc->side;
        }
        dim3 grid((W + 255) / 256, nrows), block(256);
        hipLaunchKernelGGL(ncc_border_kernel, grid, block, 0, sb, d_disp, d_peak, d_idisp, d_peak2, W, H, r, row0, row0 + nrows);
    }
    if (W < window || H < window) { mis_prof_mark(ctx); mis_prof_mark(ctx); mis_prof_mark(ctx); return MIS_OK; }
    if (ctx->dtype == MIS_U16) {
        NccArgs a;
        a.W = W; a.H = H; a.pitch = 0;
        a.dmin = dmin; a.dmax = dmax;
        a.row0 = row0 > r ? row0 : r;
        a.row1 = (row0 + nrows < H - r) ? row0 + nrows : H - r;
        a.out_pitch = W;
        a.nstrips = a.nbands = 0;
        a.nchunk = (dmax - dmin + 1 + 63) / 64;
        if (a.row0 >= a.row1) { mis_prof_mark(ctx); mis_prof_mark(ctx); mis_prof_mark(ctx); return MIS_OK; }
        {   // taller bands (less warm-up work per output row) when the grid still fills the 3072 wave slots several times
            const long long strips = (W - 2 * r + 31) / 32, bands256 = (a.row1 - a.row0 + 255) / 256;
            a.band_rows = strips * bands256 * a.nchunk >= 6 * 3072 ? 256 : 128;
        }
        return mis_ncc16_run(ctx, window, a, d_disp, d_peak, d_idisp, d_peak2);
    }
    if (ctx->dtype != MIS_U8)
        return mis_fail(ctx, MIS_ERR_UNSUPPORTED, "dense NCC: 8-bit and 16-bit rasters only");
    if (!ctx->ncc) ctx->ncc = new NccWs();
    int rc = MIS_OK;
    {
        // rows the kernels of this call read: packed pairs and vertical packs of the window rows of every
        // output row, statistics of the output rows themselves
        const int o0 = row0 > r ? row0 : r, o1 = (row0 + nrows < H - r) ? row0 + nrows : H - r;
        if (o0 < o1) rc = ncc_prepare(ctx, window, fast, o0 - r, o1 + r);
    }
    if (rc) { (void)ncc_join(ctx, ctx->ncc); return rc; }

    NccWs* w = ctx->ncc;
    NccArgs a;
    a.W = W; a.H = H; a.pitch = w->pitch;
    a.dmin = dmin; a.dmax = dmax;
    a.row0 = row0 > r ? row0 : r;
    a.row1 = (row0 + nrows < H - r) ? row0 + nrows : H - r;
    a.out_pitch = W;
    a.nstrips = a.nbands = 0;
    if (a.row0 >= a.row1) {
        mis_prof_mark(ctx); mis_prof_mark(ctx); mis_prof_mark(ctx);
        rc = ncc_join(ctx, w);
        const int rcb = ncc_mark_built(ctx, w);
        return rc ? rc : rcb;
    }
    const int nd = dmax - dmin + 1;
    const int total_chunks = (nd + 63) / 64;
    // Rows per workgroup: every band re-runs the column sums of the 2r rows above it, so taller bands
    // do less work (256 rows: 12.0 ms at 8192x8192, +-256 against 12.15 with 128; 384 and 512 leave too
    // few, uneven rounds of wavefronts: 12.5 and 12.35), as long as the grid still has several rounds of
    // wavefronts for the 5120 wave slots of the chip.
    a.band_rows = 128;
    {
        const long long strips = (W - 2 * r + 31) / 32, bands256 = (a.row1 - a.row0 + 255) / 256;
        if (strips * bands256 * (nd / 64 > 0 ? nd / 64 : 1) >= 6 * 5120) a.band_rows = 256;
    }
    a.nchunk = total_chunks;
    static const int mfma_env = [] { const char* e = getenv("MISTEREO_NCC_MFMA"); return e ? atoi(e) : 1; }();
    const bool use_mfma = (mfma_env != 0 && r <= NCC_MF_MAX_R);
    // wavefronts per SIMD the register budget is set for: 5 up to 23 x 23 windows, 4 beyond; the
    // second-peak variant keeps 16 more keys in registers: 4 wavefronts for every window
#define NCC_MW(RR) ((RR) <= 11 ? 5 : 4)
#define NCC_CASE(RR) case RR: rc = use_mfma ? ncc_mfma_launch(ctx, RR, a, d_disp, d_peak, d_idisp, d_peak2) \
                               : d_peak2 ? ncc_launch_fast<RR, 32, 4, true>(ctx, a, d_disp, d_peak, d_idisp, d_peak2) \
                               : ncc_launch_fast<RR, 32, NCC_MW(RR), false>(ctx, a, d_disp, d_peak, d_idisp, nullptr); break;
    switch (r) {
        NCC_CASE(1) NCC_CASE(2) NCC_CASE(3) NCC_CASE(4) NCC_CASE(5) NCC_CASE(6) NCC_CASE(7) NCC_CASE(8)
        NCC_CASE(9) NCC_CASE(10) NCC_CASE(11) NCC_CASE(12) NCC_CASE(13) NCC_CASE(14) NCC_CASE(15)
        default: rc = mis_fail(ctx, MIS_ERR_ARG, "dense NCC: unsupported window %d", window);
    }
#undef NCC_CASE
    if (w->forked) {                               // a failed launch path: leave no work behind on the side stream
        const int rcj = ncc_join(ctx, w);
        if (!rc) rc = rcj;
    }
    const int rcb = ncc_mark_built(ctx, w);
    if (!rc) rc = rcb;
    mis_prof_mark(ctx);
    return rc;
}

extern "C" {

int mis_ncc_dense_rows_device(mis_ctx* ctx, int window, int dmin, int dmax, int row0, int nrows,
                              float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2)
{
    return ncc_run(ctx, window, dmin, dmax, row0, nrows, d_disp, d_peak, d_idisp, d_peak2);
}

int mis_ncc_dense_device(mis_ctx* ctx, int window, int dmin, int dmax,
                         float* d_disp, float* d_peak, int16_t* d_idisp, float* d_peak2)
{
    if (!ctx) return MIS_ERR_ARG;
    return ncc_run(ctx, window, dmin, dmax, 0, ctx->height, d_disp, d_peak, d_idisp, d_peak2);
}

int mis_ncc_dense(mis_ctx* ctx, int window, int dmin, int dmax,
                  float* disp, float* peak, int16_t* idisp, float* peak2)
{
    if (!ctx) return MIS_ERR_ARG;
    return mis_ncc_dense_rows(ctx, window, dmin, dmax, 0, ctx->height, disp, peak, idisp, peak2);
}

int mis_ncc_dense_rows(mis_ctx* ctx, int window, int dmin, int dmax, int row0, int nrows,
                       float* disp, float* peak, int16_t* idisp, float* peak2)
{
    if (!ctx) return MIS_ERR_ARG;
    if (!ctx->img[0]) return mis_fail(ctx, MIS_ERR_STATE, "dense NCC: no image pair bound");
    if (!disp || !peak) return mis_fail(ctx, MIS_ERR_ARG, "dense NCC: disp and peak outputs are required");
    if (row0 < 0 || nrows < 0 || row0 + nrows > ctx->height)
        return mis_fail(ctx, MIS_ERR_ARG, "dense NCC: rows [%d,%d) outside the raster", row0, row0 + nrows);
    if (nrows == 0) return MIS_OK;
    MIS_HIP(ctx, hipSetDevice(ctx->device));
    const int H = ctx->height;
    const size_t n = (size_t)ctx->width * H, rowpx = (size_t)ctx->width;
    // device-side rasters, kept between calls
    const size_t a4 = (n * 4 + 255) & ~(size_t)255, a2 = (n * 2 + 255) & ~(size_t)255;
    int rc = mis_reserve(ctx, &ctx->out_buf, &ctx->out_bytes, 3 * a4 + a2);
    if (rc) return rc;
    float* dd = (float*)ctx->out_buf;
    float* dp = (float*)((char*)ctx->out_buf + a4);
    float* dp2 = peak2 ? (float*)((char*)ctx->out_buf + 2 * a4) : nullptr;
    int16_t* di = idisp ? (int16_t*)((char*)ctx->out_buf + 3 * a4) : nullptr;
    // The frame is computed in row strips and every finished strip is copied back on a second stream
    // while the next ones are computed (the copy to host memory is longer than the computation).
    // (page-locked outputs travel at the link's rate: more, shorter strips leave less of the last copy exposed)
    constexpr int NSMAX = 8;
    const int NS = mis_host_pinned(disp, n * 4) && mis_host_pinned(peak, n * 4) ? 8 : 4;
    if (!ctx->strip_done[0])
        for (int i = 0; i < NSMAX; i++) MIS_HIP(ctx, hipEventCreateWithFlags(&ctx->strip_done[i], hipEventDisableTiming));
    int r0[NSMAX + 1];
    for (int i = 0; i <= NS; i++) r0[i] = row0 + ((int)((long long)nrows * i / NS) & ~3);
    r0[NS] = row0 + nrows;
    for (int i = 0; i < NS && !rc; i++) {
        if (r0[i + 1] <= r0[i]) continue;
        rc = ncc_run(ctx, window, dmin, dmax, r0[i], r0[i + 1] - r0[i], dd, dp, di, dp2);
        if (!rc) MIS_HIP(ctx, hipEventRecord(ctx->strip_done[i], ctx->stream));
    }
    hipError_t e = hipSuccess;
    if (!rc) {
        MisCopyJob jobs[4 * NSMAX];
        int nj = 0;
        for (int i = 0; i < NS; i++) {
            if (r0[i + 1] <= r0[i]) continue;
            const size_t o = (size_t)r0[i] * rowpx, m = (size_t)(r0[i + 1] - r0[i]) * rowpx;
            jobs[nj++] = { disp + o, dd + o, m * 4, ctx->strip_done[i] };
            jobs[nj++] = { peak + o, dp + o, m * 4, nullptr };
            if (idisp) jobs[nj++] = { idisp + o, di + o, m * 2, nullptr };
            if (peak2) jobs[nj++] = { peak2 + o, dp2 + o, m * 4, nullptr };
        }
        rc = mis_copy_back(ctx, jobs, nj);
    }
    if (ctx->copy_stream) e = hipStreamSynchronize(ctx->copy_stream);
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = e2;
    if (!rc && e != hipSuccess) rc = mis_fail(ctx, MIS_ERR_DEVICE, "dense NCC: copy back failed: %s", hipGetErrorString(e));
    return rc;
}

}  // extern "C"

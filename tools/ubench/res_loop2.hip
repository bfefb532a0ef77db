// Inner loop of a register-resident row-group SpMM, second form: 256-byte staged rows, 4 column
// classes of 16 lanes (the layout of spmm_pipe), the two phases of a step walk the two halves
// (A | B) of every group's column list and the accumulators of a wave's G groups persist across
// them, so a group is folded ONCE per step (24 VALU) instead of once per phase.
//
//   lane = 16 q + li: class q (0..3), chunk li = 16 B of the 256-byte row
//   slot  = 4 columns x 4 rows x 64 features = 4 MFMAs (cbsz = 2, abid = slot & 3: one weight VGPR
//           per 4 slots), ONE ds_read_b128 through a resident per-lane address, no VALU
//
// Also checks the arithmetic of one step against the host.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o res_loop2 res_loop2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SH = 18;              // slots per half (72 columns)
constexpr int WH = (SH + 3) / 4;    // weight registers per half
constexpr int UROWS = 480;          // staged rows (both regions)

__host__ __device__ inline unsigned hash3(unsigned a, unsigned b, unsigned c) {
    unsigned h = a * 2654435761u ^ (b + 0x9e3779b9u) * 40503u ^ (c + 77u) * 2246822519u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return h;
}
__host__ __device__ inline int urow(int wave, int g, int ph, int s, int q) {
    return (int)(hash3(wave * 8 + g, s * 2 + ph, q) % (UROWS / 2)) + ph * (UROWS / 2);
}
__host__ __device__ inline float wval(int wave, int g, int ph, int s, int q, int i) {
    return (float)((int)(hash3(wave * 131 + g, (s * 2 + ph) * 4 + q, i + 9) % 2001) - 1000) * 1e-3f;
}
__host__ __device__ inline float xval(int u, int f) {
    return (float)((int)(hash3(u, f, 5) % 4001) - 2000) * 5e-4f;
}

__device__ __forceinline__ void dma16_saddr(unsigned voff, const void* sbase, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}

// VAR bit0: no fold/store   bit1: no operand reads (ring holds constants)   bit2: no MFMAs
//     bit3: DMA staging on  bit4: DMA source advances every step   bit5: every range SH slots long, no exit branches
//     bit6: exit test once per 4 slots (ranges rounded up to whole weight registers)
//     bit7: accumulators in AGPRs (asm MFMA)   bit8: operand ring in AGPRs (ds_read -> AGPR, MFMA B from AGPR)
template <int NW, int G, int D, int VAR>
__global__ __launch_bounds__(NW * 64) void res_loop2(float* y, const int* nsl, const float* xsrc, long long xadv,
                                                     int steps, int check, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = lane >> 4, li = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    for (int i = tid; i < UROWS * 64; i += NW * 64) ((float*)lds)[i] = xval(i >> 6, i & 63);
    unsigned addr[2][G][SH];
    float w[2][G][WH];
    int n[2][G];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int s = 0; s < SH; ++s) {
            addr[ph][g][s] = lds0 + (unsigned)urow(wave, g, ph, s, q) * 256u + li * 16;
            asm volatile("" : "+v"(addr[ph][g][s]));
        }
#pragma unroll
        for (int p = 0; p < WH; ++p) {
            const int s = 4 * p + (li >> 2);
            w[ph][g][p] = s < SH ? wval(wave, g, ph, s, q, li & 3) : 0.f;
            asm volatile("" : "+v"(w[ph][g][p]));
        }
        n[ph][g] = (VAR & 32) ? SH : __builtin_amdgcn_readfirstlane(nsl[((blockIdx.x * NW + wave) * G + g) * 2 + ph]);
        if (VAR & 64) n[ph][g] = (n[ph][g] + 3) & ~3;
    }
    // DMA: pieces of 4 staged rows (1 KiB); NP per wave and region
    constexpr int NP = (UROWS / 2 / 4 + NW - 1) / NW;
    unsigned voff[2][NP];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int u = ph * (UROWS / 2) + (p * NW + wave) * 4 + (lane >> 4);
        voff[ph][p] = (unsigned)(((hash3(blockIdx.x, u, 3) % 4096) * 64 + (blockIdx.x % 64) * 4096 * 64) * 4 + li * 16);
    }
    const char* xs = reinterpret_cast<const char*>(xsrc);
    const long long xadv_eff = (VAR & 16) ? xadv : 0;
    auto dma = [&](int ph, const char* src, int p0, int p1) {
        if constexpr (!(VAR & 8)) return;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            if (p >= p0 && p < p1 && (p * NW + wave) * 4 < UROWS / 2)
                dma16_saddr(voff[ph][p], src, __builtin_amdgcn_readfirstlane(
                    lds0 + (unsigned)(ph * (UROWS / 2) + (p * NW + wave) * 4) * 256u));
    };
    float* ybase = y + ((long long)(blockIdx.x * NW + wave) * G * 4 + q) * 64 + li * 4;
    __syncthreads();

    f32x4 ring[2][D];
    f32x4 acc[G][4];
    if constexpr ((VAR & 2) != 0) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int s = 0; s < D; ++s) { ring[r][s] = f32x4{1.f + lane, 2.f, 3.f, 4.f}; asm volatile("" : "+v"(ring[r][s])); }
    }
    auto emit = [&](int g, int step) {
        if constexpr (VAR & 1) {
            asm volatile("" :: "v"(acc[g][0]), "v"(acc[g][1]), "v"(acc[g][2]), "v"(acc[g][3]));
            return;
        }
        f32x4 out;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            auto p01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[g][m].x), __float_as_uint(acc[g][m].y), false, false);
            auto p23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[g][m].z), __float_as_uint(acc[g][m].w), false, false);
            const float r01 = __uint_as_float(p01[0]) + __uint_as_float(p01[1]);
            const float r23 = __uint_as_float(p23[0]) + __uint_as_float(p23[1]);
            auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(r01), __float_as_uint(r23), false, false);
            out[m] = __uint_as_float(h[0]) + __uint_as_float(h[1]);
        }
        float* dst = ybase + (long long)g * 4 * 64;
        if (check) { if (step == 0) *reinterpret_cast<f32x4*>(dst) = out; }
        else __builtin_nontemporal_store(out, reinterpret_cast<f32x4*>(dst));
    };
    // operand reads and their waits are inline asm (hipcc sinks a plain LDS load to its use across
    // the scalar exit branches); LDS operations return in order, so the wait counts are static
#define RD(P_, G_, S_) { if constexpr ((VAR & 256) != 0) asm volatile("ds_read_b128 %0, %1" : "=a"(ring[(G_) & 1][(S_) % D]) : "v"(addr[P_][G_][S_])); \
                         else asm volatile("ds_read_b128 %0, %1" : "=v"(ring[(G_) & 1][(S_) % D]) : "v"(addr[P_][G_][S_])); }
#define WAITN(G_, S_) ((S_) >= D ? ((SH - 1 - (S_)) < (D - 1) ? (SH - 1 - (S_)) : (D - 1)) : (D - 1 + ((G_) + 1 < G ? D : 0)))
#define WT(G_, S_) { if constexpr ((VAR & 256) != 0) asm volatile("s_waitcnt lgkmcnt(%1)" : "+a"(ring[(G_) & 1][(S_) % D]) : "n"(WAITN(G_, S_))); \
                     else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[(G_) & 1][(S_) % D]) : "n"(WAITN(G_, S_))); }
#define MF(ACC_, W_, X_, AB_) __builtin_amdgcn_mfma_f32_4x4x1f32(W_, X_, ACC_, 2, AB_, 0)
#define MFA(ACC_, W_, X_, AB_) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:2 abid:" #AB_ : "+a"(ACC_) : "v"(W_), "v"(X_))
#define MFB(ACC_, W_, X_, AB_) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:2 abid:" #AB_ : "+v"(ACC_) : "v"(W_), "a"(X_))
#define SLOT4(P_, G_, S_, AB_, FIRST_)                                                             \
    {                                                                                              \
        const f32x4 x = ring[(G_) & 1][(S_) % D];                                                  \
        const float wv = w[P_][G_][(S_) >> 2];                                                     \
        if constexpr ((VAR & 4) != 0) { asm volatile("" :: "v"(x), "v"(wv)); }                     \
        else if constexpr ((VAR & 128) != 0) {                                                     \
            if (FIRST_) { const f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[G_][0] = z; acc[G_][1] = z; acc[G_][2] = z; acc[G_][3] = z; } \
            MFA(acc[G_][0], wv, x.x, AB_); MFA(acc[G_][1], wv, x.y, AB_);                          \
            MFA(acc[G_][2], wv, x.z, AB_); MFA(acc[G_][3], wv, x.w, AB_);                          \
        } else if constexpr ((VAR & 256) != 0) {                                                   \
            if (FIRST_) { const f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[G_][0] = z; acc[G_][1] = z; acc[G_][2] = z; acc[G_][3] = z; } \
            MFB(acc[G_][0], wv, x.x, AB_); MFB(acc[G_][1], wv, x.y, AB_);                          \
            MFB(acc[G_][2], wv, x.z, AB_); MFB(acc[G_][3], wv, x.w, AB_);                          \
        } else if (FIRST_) {                                                                       \
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};                                                  \
            acc[G_][0] = MF(z, wv, x.x, AB_); acc[G_][1] = MF(z, wv, x.y, AB_);                    \
            acc[G_][2] = MF(z, wv, x.z, AB_); acc[G_][3] = MF(z, wv, x.w, AB_);                    \
        } else {                                                                                   \
            acc[G_][0] = MF(acc[G_][0], wv, x.x, AB_); acc[G_][1] = MF(acc[G_][1], wv, x.y, AB_);  \
            acc[G_][2] = MF(acc[G_][2], wv, x.z, AB_); acc[G_][3] = MF(acc[G_][3], wv, x.w, AB_);  \
        }                                                                                          \
    }
#define SLOT(P_, G_, S_, FIRST_)                                                                   \
    if (((S_) & 3) == 0) SLOT4(P_, G_, S_, 0, FIRST_) else if (((S_) & 3) == 1) SLOT4(P_, G_, S_, 1, FIRST_) \
    else if (((S_) & 3) == 2) SLOT4(P_, G_, S_, 2, FIRST_) else SLOT4(P_, G_, S_, 3, FIRST_)
    // one phase: region P of the stage; the first D reads of group g + 1 are requested at the start
    // of group g (other ring); the fold + store of a finished group sits under those reads
#define PHASE(P_)                                                                                  \
    if (!(VAR & 2)) { _Pragma("unroll") for (int s = 0; s < D; ++s) RD(P_, 0, s); }                \
    _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                \
        if (g + 1 < G && !(VAR & 2)) { _Pragma("unroll") for (int s = 0; s < D; ++s) RD(P_, g + 1, s); } \
        if ((P_) == 0 && g == 0) { if (first) first = false; else emit(G - 1, step - 1); }         \
        if ((P_) == 1 && g > 0) emit(g - 1, step);                                                 \
        if (G > 1 && g + 1 < G) dma(1 - (P_), (P_) ? xs + xadv_eff : xs, (NP * g) / (G - 1), (NP * (g + 1)) / (G - 1)); \
        _Pragma("unroll") for (int s = 0; s < SH; ++s) {                                           \
            if (!(VAR & 2)) WT(g, s);                                                              \
            SLOT(P_, g, s, (P_) == 0 && s == 0)                                                    \
            if (s + D < SH && !(VAR & 2)) RD(P_, g, s + D);                                        \
            if (!(VAR & 32) && s + 1 == n[P_][g]) break;                                           \
        }                                                                                          \
    }
    bool first = true;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int step = 0; step < steps; ++step) {
        // re-made opaque every step: otherwise hipcc hoists all exit comparisons out of the time
        // loop as 64-bit masks and spills them to VGPR lanes
#pragma unroll
        for (int g = 0; g < G; ++g) asm volatile("" : "+s"(n[0][g]), "+s"(n[1][g]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        PHASE(0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        PHASE(1)
        if (VAR & 16) xs += xadv;
    }
    emit(G - 1, steps - 1);
    if (tid == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
}

template <int NW, int G, int D, int VAR>
double run(const char* name, int steps, bool check = false) {
    const int blocks = 256;
    float* y; int* nsl; float* xsrc; unsigned long long* cyc;
    (void)hipMalloc(&cyc, blocks * 8);
    const size_t ybytes = (size_t)blocks * NW * G * 4 * 64 * 4;
    (void)hipMalloc(&y, ybytes); (void)hipMemset(y, 0, ybytes);
    std::vector<int> hn((size_t)blocks * NW * G * 2);
    for (size_t i = 0; i < hn.size(); ++i) hn[i] = 15 + (int)(hash3((unsigned)i, 1, 2) % 4);   // 15..18
    (void)hipMalloc(&nsl, hn.size() * 4); (void)hipMemcpy(nsl, hn.data(), hn.size() * 4, hipMemcpyHostToDevice);
    const size_t xbytes = (size_t)64 * 4096 * 64 * 4 + (size_t)(steps + 2) * 1024 * 1024;
    (void)hipMalloc(&xsrc, xbytes); (void)hipMemset(xsrc, 0, xbytes);
    auto k = res_loop2<NW, G, D, VAR>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(NW * 64), 160 * 1024, 0, y, nsl, xsrc, (long long)1024 * 1024, check ? 2 : steps, check ? 1 : 0, cyc);
    (void)hipEventRecord(e0);
    if (!check) hipLaunchKernelGGL(k, dim3(blocks), dim3(NW * 64), 160 * 1024, 0, y, nsl, xsrc, (long long)1024 * 1024, steps, 0, cyc);
    (void)hipEventRecord(e1);
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(err)); exit(1); }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double result = 0;
    if (check) {
        std::vector<float> hy(ybytes / 4);
        (void)hipMemcpy(hy.data(), y, ybytes, hipMemcpyDeviceToHost);
        double worst = 0; long bad = 0;
        for (int b = 0; b < 2; ++b) for (int wv = 0; wv < NW; ++wv) for (int g = 0; g < G; ++g)
            for (int i = 0; i < 4; ++i) for (int f = 0; f < 64; ++f) {
                double ref = 0;
                for (int ph = 0; ph < 2; ++ph) {
                    const int ng = hn[(((size_t)b * NW + wv) * G + g) * 2 + ph];
                    for (int s = 0; s < ng; ++s) for (int qq = 0; qq < 4; ++qq)
                        ref += (double)wval(wv, g, ph, s, qq, i) * (double)xval(urow(wv, g, ph, s, qq), f);
                }
                const float got = hy[(((size_t)b * NW + wv) * G * 4 + g * 4 + i) * 64 + f];
                const double e = fabs(got - ref);
                if (e > worst) worst = e;
                if (e > 1e-4) { if (bad < 6) printf("  mismatch b%d w%d g%d row%d f%d got %g want %g\n", b, wv, g, i, f, got, ref); ++bad; }
            }
        printf("%-44s NW=%2d G=%d D=%d check: max |err| = %.3g, mismatches = %ld\n", name, NW, G, D, worst, bad);
        result = (double)bad;
    } else {
        double slots = 0;
        for (size_t i = 0; i < hn.size(); ++i) slots += (VAR & 32) ? SH : (VAR & 64) ? (hn[i] + 3) / 4 * 4 : hn[i];
        std::vector<unsigned long long> hc(blocks);
        (void)hipMemcpy(hc.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double cavg = 0; for (auto c : hc) cavg += (double)c / blocks;
        const double slots_per_simd = slots / blocks / 4.0 * steps;
        const double ns = ms * 1e6 / slots_per_simd;
        printf("%-34s NW=%2d G=%d D=%d  ns/slot/SIMD=%6.2f  cycles/slot/SIMD=%6.1f (pipe %5.1f%% busy)  clock %.2f GHz  %.3f ms\n", name, NW, G, D,
               ns, cavg / slots_per_simd, 100.0 * 32 / (cavg / slots_per_simd), cavg / (ms * 1e6), ms);
        result = ns;
    }
    (void)hipFree(y); (void)hipFree(nsl); (void)hipFree(xsrc);
    return result;
}

int main() {
    const int steps = 400;
    double bad = run<8, 3, 4, 0>("arithmetic check", 2, true);
    bad += run<8, 2, 4, 0>("arithmetic check", 2, true);
    bad += run<16, 1, 3, 0>("arithmetic check", 2, true);
    run<8, 3, 4, 0>("full", steps);
    run<8, 3, 3, 0>("full", steps);
    run<8, 3, 6, 0>("full", steps);
    run<8, 3, 4, 1>("no fold/store", steps);
    run<8, 3, 4, 2>("no operand reads", steps);
    run<8, 3, 4, 3>("MFMA only", steps);
    run<8, 3, 4, 4>("no MFMA", steps);
    run<8, 3, 4, 8>("full + DMA (L2 hits)", steps);
    run<8, 3, 4, 24>("full + DMA (advancing source)", steps);
    run<8, 3, 4, 32>("fixed 18 slots, no branches", steps);
    run<8, 3, 4, 35>("fixed 18, MFMA only", steps);
    run<8, 3, 4, 64>("exit test per 4 slots", steps);
    run<8, 3, 3, 64>("exit test per 4 slots", steps);
    run<8, 3, 4, 67>("exit per 4, MFMA only", steps);
    run<16, 1, 3, 32>("fixed 18 slots, no branches", steps);
    run<16, 1, 3, 35>("fixed 18, MFMA only", steps);
    run<16, 1, 3, 64>("exit test per 4 slots", steps);
    run<16, 1, 3, 3>("MFMA only", steps);
    run<16, 1, 3, 64 + 128>("exit per 4, acc in AGPRs", steps);
    run<16, 1, 3, 64 + 256>("exit per 4, ring in AGPRs", steps);
    run<8, 3, 4, 64 + 128>("exit per 4, acc in AGPRs", steps);
    run<8, 3, 4, 64 + 256>("exit per 4, ring in AGPRs", steps);
    run<8, 3, 4, 64 + 1>("exit per 4, no fold", steps);
    run<8, 3, 4, 64 + 1 + 128>("exit per 4, no fold, acc AGPR", steps);
    run<8, 3, 4, 64 + 1 + 256>("exit per 4, no fold, ring AGPR", steps);
    run<8, 2, 4, 0>("full", steps);
    run<8, 2, 6, 0>("full", steps);
    run<8, 2, 4, 8>("full + DMA (L2 hits)", steps);
    run<12, 2, 4, 0>("full", steps);
    run<12, 2, 4, 8>("full + DMA (L2 hits)", steps);
    run<16, 1, 3, 0>("full", steps);
    run<16, 1, 3, 8>("full + DMA (L2 hits)", steps);
    return bad != 0;
}

// Inner loop of the register-resident row-group SpMM (spmm_res): what one CU sustains when the
// tile's stream -- weights and the per-lane LDS addresses of the staged rows -- lives in VGPRs for
// the whole time loop, so a super-step is ONE ds_read_b128 + 4 v_mfma_f32_4x4x1_16b_f32 and no VALU.
//
//   lane = 8 q + c: class q (0..7) walks its own column, chunk c = 16 B of the 128-byte half row
//   slot  = 8 columns x 4 rows x 32 features = 4 MFMAs; weights of two slots share one VGPR
//           (cbsz = 1, abid = slot & 1: block pair {2q, 2q+1} takes its A values from block 2q + abid)
//   phase = one 32-feature half of a time step (lo / hi: same address register, +1024 immediate)
//   stage = 8-row blocks of 2 KiB: [8 rows x 128 B lo][8 rows x 128 B hi]
//
// The program also checks the arithmetic of one step against the host (cbsz = 1 semantics, the
// 8-class fold, the store layout).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o res_loop res_loop.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) f32x4* lds_f4_t;

constexpr int SMAX = 18;            // slots per group (144 columns)
constexpr int UROWS = 576;          // staged rows

__host__ __device__ inline unsigned hash3(unsigned a, unsigned b, unsigned c) {
    unsigned h = a * 2654435761u ^ (b + 0x9e3779b9u) * 40503u ^ (c + 77u) * 2246822519u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return h;
}
// staged row of (wave, group, slot, class): even rows for classes {0,1,4,5}, odd for {3,2,7,6}
// (conflict-free ds_read_b128: the lane groups of the LDS pair class 0 with 3 and 1 with 2)
__host__ __device__ inline int urow(int wave, int g, int s, int q) {
    const int odd = ((q & 3) == 2 || (q & 3) == 3) ? 1 : 0;
    return 2 * (int)(hash3(wave * 8 + g, s, q) % (UROWS / 2)) + odd;
}
__host__ __device__ inline float wval(int wave, int g, int s, int q, int i) {
    return (float)((int)(hash3(wave * 131 + g, s * 8 + q, i + 9) % 2001) - 1000) * 1e-3f;
}
__host__ __device__ inline float xval(int u, int f) {        // staged row u, feature f (0..63)
    return (float)((int)(hash3(u, f, 5) % 4001) - 2000) * 5e-4f;
}
__host__ __device__ inline unsigned stage_off(int u, int half, int c) {
    return (unsigned)((u >> 3) * 2048 + half * 1024 + (u & 7) * 128 + c * 16);
}

__device__ __forceinline__ void dma16_saddr(unsigned voff, const void* sbase, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}

// VAR bit0: no emit (fold + store)   bit1: no operand reads   bit2: no MFMAs   bit3: DMA staging on
//     bit4: DMA source advances every step (misses)   bit5: unconditional exit only at SMAX (n = SMAX)
template <int NW, int G, int D, int VAR>
__global__ __launch_bounds__(NW * 64) void res_loop(float* y, const int* nsl, const float* xsrc, long long xadv,
                                                    int steps, int check) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = lane >> 3, c = lane & 7;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    for (int i = tid; i < UROWS * 64; i += NW * 64) {
        const int u = i >> 6, f = i & 63;
        *(float*)(lds + stage_off(u, f >> 5, (f & 31) >> 2) + (f & 3) * 4) = xval(u, f);
    }
    unsigned addr[G][SMAX];
    float w[G][SMAX / 2];
    int n[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
            addr[g][s] = lds0 + stage_off(urow(wave, g, s, q), 0, c);
            asm volatile("" : "+v"(addr[g][s]));
        }
#pragma unroll
        for (int p = 0; p < SMAX / 2; ++p) {
            w[g][p] = wval(wave, g, 2 * p + (c >> 2), q, c & 3);
            asm volatile("" : "+v"(w[g][p]));
        }
        n[g] = (VAR & 32) ? SMAX : __builtin_amdgcn_readfirstlane(nsl[(blockIdx.x * NW + wave) * G + g]);
    }
    // DMA: this wave's pieces (8 staged rows x 128 B each)
    constexpr int NP = (UROWS / 8 + NW - 1) / NW;
    unsigned voff[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int blk = p * NW + wave;
        const int u = blk * 8 + (lane >> 3);
        voff[p] = (unsigned)(((hash3(blockIdx.x, u, 3) % 4096) * 64 + (blockIdx.x % 64) * 4096 * 64) * 4 + (lane & 7) * 16);
    }
    const char* xs = reinterpret_cast<const char*>(xsrc);
    const unsigned piece0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 2048u);
    const long long xadv_eff = (VAR & 16) ? xadv : 0;
    auto dma = [&](int half, const char* src, int p0, int p1) {
        if constexpr (!(VAR & 8)) return;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (p >= p0 && p < p1 && (p * NW + wave) * 8 < UROWS)
                dma16_saddr(voff[p], src + half * 128, piece0 + (unsigned)p * (NW * 2048u) + half * 1024u);
        }
    };
    // output row of (group, lane): lane (b5, b4, b3) keeps row 2 b4 + b3, features 4c + 2 b5 + {0, 1}
    const int orow = ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
    const int ofeat = 4 * c + 2 * (lane >> 5);
    float* ybase = y + ((long long)(blockIdx.x * NW + wave) * G * 4 + orow) * 64 + ofeat;
    __syncthreads();

    f32x4 ring[2][D];
    f32x4 acc[4];
    auto zero = [&]() { acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0]; acc[2] = acc[0]; acc[3] = acc[0]; };
    auto emit = [&](int g, int half, int step) {
        if constexpr (VAR & 1) {
            asm volatile("" :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
            return;
        }
        auto sw32 = [](float a, float b) {
            auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
            return __uint_as_float(p[0]) + __uint_as_float(p[1]);
        };
        auto sw16 = [](float a, float b) {
            auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
            return __uint_as_float(p[0]) + __uint_as_float(p[1]);
        };
        float V[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            // bit 5: lanes < 32 keep m = x, lanes >= 32 keep m = 2 + x
            const float s0 = sw32(acc[x][0], acc[2 + x][0]), s1 = sw32(acc[x][1], acc[2 + x][1]);
            const float s2 = sw32(acc[x][2], acc[2 + x][2]), s3 = sw32(acc[x][3], acc[2 + x][3]);
            // bit 4: even 16-lane rows keep rows 0 / 1, odd ones rows 2 / 3
            const float t0 = sw16(s0, s2), t1 = sw16(s1, s3);
            // bit 3: lanes with bit 3 clear keep t0's row, the others t1's
            const bool hi8 = (lane & 8) != 0;
            const float keep = hi8 ? t1 : t0, give = hi8 ? t0 : t1;
            V[x] = keep + __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(give), 0x128, 0xf, 0xf, true));
        }
        float* dst = ybase + (long long)g * 4 * 64 + half * 32;
        if (check) { if (step == 0) *reinterpret_cast<f32x2*>(dst) = f32x2{V[0], V[1]}; }
        else __builtin_nontemporal_store(f32x2{V[0], V[1]}, reinterpret_cast<f32x2*>(dst));
    };
    // operand reads and their waits are inline asm: hipcc sinks a plain LDS load to its use across
    // the (scalar) exit branches, which serialises read -> wait -> 4 MFMAs.  LDS operations return in
    // order, so the counts below are static (see WAITN).
#define RD(G_, S_, H_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[(G_) & 1][(S_) % D]) : "v"(addr[G_][S_]), "n"((H_) * 1024))
    // LDS reads issued after r(g, s) when slot (g, s) starts: its own ring refills, or (for the D
    // slots requested ahead) the rest of that request, the look-ahead request of group g + 1 and the
    // refills of slots 0 .. s-1
#define WAITN(G_, S_) ((S_) >= D ? ((SMAX - 1 - (S_)) < (D - 1) ? (SMAX - 1 - (S_)) : (D - 1)) : (D - 1 + ((G_) + 1 < G ? D : 0)))
#define WT(G_, S_) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[(G_) & 1][(S_) % D]) : "n"(WAITN(G_, S_)))
    auto slot_mfma = [&](int g, int s) {
        const f32x4 x = ring[g & 1][s % D];
        const float wv = w[g][s >> 1];
        if constexpr (VAR & 4) { asm volatile("" :: "v"(x), "v"(wv)); return; }
        if (s & 1) {
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, x.x, acc[0], 1, 1, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, x.y, acc[1], 1, 1, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, x.z, acc[2], 1, 1, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, x.w, acc[3], 1, 1, 0);
        } else {
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, x.x, acc[0], 1, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, x.y, acc[1], 1, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, x.z, acc[2], 1, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, x.w, acc[3], 1, 0, 0);
        }
    };
    // one phase (feature half H of a step): groups one after another; the first D reads of group
    // g + 1 are requested at the start of group g (other ring), so a group never starts cold
#define PHASE(H_)                                                                                  \
    if (!(VAR & 2)) { _Pragma("unroll") for (int s = 0; s < D; ++s) RD(0, s, H_); }                \
    _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                \
        if (g + 1 < G && !(VAR & 2)) { _Pragma("unroll") for (int s = 0; s < D; ++s) RD(g + 1, s, H_); } \
        if (g == 0) { if (first) first = false; else emit(G - 1, 1 - (H_), (H_) ? step : step - 1); } \
        else emit(g - 1, H_, step);                                                                \
        zero();                                                                                    \
        if (G > 1 && g + 1 < G) dma(1 - (H_), (H_) ? xs + xadv_eff : xs, (NP * g) / (G - 1), (NP * (g + 1)) / (G - 1)); \
        _Pragma("unroll") for (int s = 0; s < SMAX; ++s) {                                         \
            if (!(VAR & 2)) WT(g, s);                                                              \
            slot_mfma(g, s);                                                                       \
            if (s + D < SMAX && !(VAR & 2)) RD(g, s + D, H_);                                      \
            if (s + 1 == n[g]) break;                                                              \
        }                                                                                          \
    }
    bool first = true;
    zero();
    for (int step = 0; step < steps; ++step) {
        // the slot counts are re-made opaque every step: otherwise hipcc hoists all G x 18 exit
        // comparisons out of the time loop as 64-bit masks and spills them to VGPR lanes
#pragma unroll
        for (int g = 0; g < G; ++g) asm volatile("" : "+s"(n[g]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        PHASE(0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        PHASE(1)
        if (VAR & 16) xs += xadv;
    }
    emit(G - 1, 1, steps - 1);
}

template <int NW, int G, int D, int VAR>
double run(const char* name, int nslots, int steps, bool check = false) {
    const int blocks = 256;
    float* y; int* nsl; float* xsrc;
    const size_t ybytes = (size_t)blocks * NW * G * 4 * 64 * 4;
    (void)hipMalloc(&y, ybytes); (void)hipMemset(y, 0, ybytes);
    std::vector<int> hn((size_t)blocks * NW * G);
    for (size_t i = 0; i < hn.size(); ++i) hn[i] = nslots > 0 ? nslots : 15 + (int)(hash3((unsigned)i, 1, 2) % 4);   // 15..18
    (void)hipMalloc(&nsl, hn.size() * 4); (void)hipMemcpy(nsl, hn.data(), hn.size() * 4, hipMemcpyHostToDevice);
    const size_t xbytes = (size_t)64 * 4096 * 64 * 4 + (size_t)(steps + 2) * 1024 * 1024;
    (void)hipMalloc(&xsrc, xbytes); (void)hipMemset(xsrc, 0, xbytes);
    auto k = res_loop<NW, G, D, VAR>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(NW * 64), 160 * 1024, 0, y, nsl, xsrc, (long long)1024 * 1024, check ? 2 : steps, check ? 1 : 0);
    (void)hipEventRecord(e0);
    if (!check) hipLaunchKernelGGL(k, dim3(blocks), dim3(NW * 64), 160 * 1024, 0, y, nsl, xsrc, (long long)1024 * 1024, steps, 0);
    (void)hipEventRecord(e1);
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(err)); exit(1); }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double result = 0;
    if (check) {
        std::vector<float> hy(ybytes / 4);
        (void)hipMemcpy(hy.data(), y, ybytes, hipMemcpyDeviceToHost);
        double worst = 0; long bad = 0;
        for (int b = 0; b < 2; ++b) for (int wv = 0; wv < NW; ++wv) for (int g = 0; g < G; ++g) {
            const int ng = hn[((size_t)b * NW + wv) * G + g];
            for (int i = 0; i < 4; ++i) for (int f = 0; f < 64; ++f) {
                double ref = 0;
                for (int s = 0; s < ng; ++s) for (int qq = 0; qq < 8; ++qq)
                    ref += (double)wval(wv, g, s, qq, i) * (double)xval(urow(wv, g, s, qq), f);
                const float got = hy[(((size_t)b * NW + wv) * G * 4 + g * 4 + i) * 64 + f];
                const double e = fabs(got - ref);
                if (e > worst) worst = e;
                if (e > 1e-4) { if (bad < 6) printf("  mismatch b%d w%d g%d row%d f%d got %g want %g\n", b, wv, g, i, f, got, ref); ++bad; }
            }
        }
        printf("%-44s check: max |err| = %.3g, mismatches = %ld\n", name, worst, bad);
        result = (double)bad;
    } else {
        double slots = 0;
        for (size_t i = 0; i < hn.size(); ++i) slots += hn[i];
        const double slots_per_simd = slots / blocks / 4.0 * 2.0 * steps;        // two phases per step
        const double ns = ms * 1e6 / slots_per_simd;
        printf("%-44s NW=%2d G=%d D=%d  ns/slot/SIMD=%6.2f  (%5.1f%% of the matrix pipe at 2.4 GHz)  %.3f ms\n", name, NW, G, D,
               ns, 100.0 * (32 / 2.4) / ns, ms);
        result = ns;
    }
    (void)hipFree(y); (void)hipFree(nsl); (void)hipFree(xsrc);
    return result;
}

int main() {
    const int steps = 400;
    double bad = run<8, 4, 4, 0>("arithmetic check", 0, 2, true);
    bad += run<12, 3, 4, 0>("arithmetic check (12 waves)", 0, 2, true);
    run<8, 4, 4, 0>("full", 0, steps);
    run<8, 4, 3, 0>("full", 0, steps);
    run<8, 4, 6, 0>("full", 0, steps);
    run<8, 4, 4, 1>("no emit", 0, steps);
    run<8, 4, 4, 2>("no operand reads", 0, steps);
    run<8, 4, 4, 3>("MFMA only", 0, steps);
    run<8, 4, 4, 4>("no MFMA", 0, steps);
    run<8, 4, 4, 32>("fixed 18 slots", 0, steps);
    run<8, 4, 4, 8>("full + DMA (L2 hits)", 0, steps);
    run<8, 4, 4, 24>("full + DMA (advancing source)", 0, steps);
    run<12, 3, 4, 0>("full", 0, steps);
    run<12, 3, 3, 0>("full", 0, steps);
    run<12, 3, 4, 8>("full + DMA (L2 hits)", 0, steps);
    run<12, 3, 4, 24>("full + DMA (advancing source)", 0, steps);
    run<16, 2, 3, 0>("full", 0, steps);
    run<16, 2, 3, 8>("full + DMA (L2 hits)", 0, steps);
    return bad != 0;
}

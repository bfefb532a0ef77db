// (1) Semantics: v_mfma_f32_4x4x1_16b_f32 with blgp = 4 + q takes its B operand from lanes 16 q .. 16 q + 15
// for ALL four 16-lane groups (B-matrix lane-group broadcast); blgp = 0 takes each lane's own value.
// (2) Issue rate of ONE wave per SIMD running the two inner bodies of the planned block-wave kernel:
//   dense:  s_waitcnt | 16 x mfma 4x4x1 (4 columns x 4 feature registers, blgp 4..7) | ds_read_b128
//   sparse: s_waitcnt | 4 x mfma 4x4x1 | ds_read_b128 | scalar exit test
// optionally beside helper waves that stream global memory into LDS (LDS-DMA) the whole time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BL> __device__ f32x4 mfb(float a, float b) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, z, 0, 0, BL);
}
__global__ void sem(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 r[8] = {mfb<0>(a[l], b[l]), mfb<1>(a[l], b[l]), mfb<2>(a[l], b[l]), mfb<3>(a[l], b[l]),
                  mfb<4>(a[l], b[l]), mfb<5>(a[l], b[l]), mfb<6>(a[l], b[l]), mfb<7>(a[l], b[l])};
    for (int s = 0; s < 8; ++s) for (int i = 0; i < 4; ++i) d[(s * 4 + i) * 64 + l] = r[s][i];
}

#define MF(ACC, W, X, AB, BL) ACC = __builtin_amdgcn_mfma_f32_4x4x1f32(W, X, ACC, 2, AB, BL)
// MODE 0: dense bodies only, 1: sparse bodies only, 2: alternating (1 dense + 2 sparse)
template <int MODE, int D>
__global__ __launch_bounds__(512) void rate(float* out, const float* gsrc, int iters, int helpers, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    for (int i = tid; i < 112 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (float)(i & 255) * 1e-3f;
    __syncthreads();
    if (wave >= 4) {                                       // helper waves: LDS-DMA stream into the upper 32 KB
        if (wave - 4 >= helpers) return;
        const char* src = reinterpret_cast<const char*>(gsrc) + (size_t)blockIdx.x * (1 << 20) + (wave - 4) * (1 << 18);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + 120 * 1024 + (wave - 4) * 4096);
        for (int it = 0; it < iters * 2; ++it) {
            unsigned voff = (unsigned)(((it * 7 + lane / 16 * 37) & 1023) * 256 + (lane & 15) * 16);
#pragma unroll
            for (int p = 0; p < 4; ++p)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff + p * 4096), "s"(src), "s"(dst + p * 1024) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        return;
    }
    unsigned addr[16];
    float w[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) addr[i] = lds0 + (unsigned)((((lane >> 4) * 97 + i * 29 + wave * 13) % 440) * 256 + (lane & 15) * 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = 1e-3f * (float)(lane + i);
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(addr[i]));
    f32x4 ring[D];
#pragma unroll
    for (int i = 0; i < D; ++i) ring[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc[4];
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0]; acc[2] = acc[0]; acc[3] = acc[0];
#define RD(S) asm volatile("ds_read_b128 %0, %1" : "+v"(ring[(S) % D]) : "v"(addr[(S) % 16]))
#define WT(S) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[(S) % D]) : "n"(D - 1))
#define DENSE(S) { WT(S); const f32x4 x = ring[(S) % D]; const float wv = w[(S) % 8];                 \
        MF(acc[0], wv, x.x, 0, 4); MF(acc[1], wv, x.y, 0, 4); MF(acc[2], wv, x.z, 0, 4); MF(acc[3], wv, x.w, 0, 4); \
        MF(acc[0], wv, x.x, 1, 5); MF(acc[1], wv, x.y, 1, 5); MF(acc[2], wv, x.z, 1, 5); MF(acc[3], wv, x.w, 1, 5); \
        MF(acc[0], wv, x.x, 2, 6); MF(acc[1], wv, x.y, 2, 6); MF(acc[2], wv, x.z, 2, 6); MF(acc[3], wv, x.w, 2, 6); \
        MF(acc[0], wv, x.x, 3, 7); MF(acc[1], wv, x.y, 3, 7); MF(acc[2], wv, x.z, 3, 7); MF(acc[3], wv, x.w, 3, 7); \
        RD((S) + D); }
#define SPARSE(S, AB) { WT(S); const f32x4 x = ring[(S) % D]; const float wv = w[((S) >> 2) % 8];     \
        MF(acc[0], wv, x.x, AB, 0); MF(acc[1], wv, x.y, AB, 0); MF(acc[2], wv, x.z, AB, 0); MF(acc[3], wv, x.w, AB, 0); \
        RD((S) + D); }
#pragma unroll
    for (int s = 0; s < D; ++s) RD(s);
    int n = 1 << 20;                                       // exit count of the unrolled bodies: never reached
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+s"(n));
        if constexpr (MODE == 0) {
#pragma unroll
            for (int s = 0; s < 12; ++s) { DENSE(s) if (s + 1 == n) break; }
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int s = 0; s < 48; ++s) { if ((s & 3) == 0) SPARSE(s, 0) else if ((s & 3) == 1) SPARSE(s, 1) else if ((s & 3) == 2) SPARSE(s, 2) else SPARSE(s, 3)
                                           if (s + 1 == n) break; }
        } else {
#pragma unroll
            for (int s = 0; s < 36; s += 3) { DENSE(s) SPARSE(s + 1, 1) SPARSE(s + 2, 2) if (s + 1 == n) break; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
    out[(blockIdx.x * 4 + wave) * 64 + lane] = acc[0].x + acc[1].y + acc[2].z + acc[3].w;
}

template <int MODE, int D> void run(const char* name, int mfma_per_iter, int helpers, float* out, float* gsrc, unsigned long long* cyc) {
    const int iters = 2000;
    auto k = rate<MODE, D>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 160 * 1024, 0, out, gsrc, 50, helpers, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 160 * 1024, 0, out, gsrc, iters, helpers, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double per_iter = (double)h[0] / iters;
    printf("%-28s D=%d helpers=%d: %.1f cycles per body-set (%d MFMAs = %d pipe cycles) -> %.1f %% of the matrix pipe; %.2f ms\n",
           name, D, helpers, per_iter, mfma_per_iter, mfma_per_iter * 8, 100.0 * mfma_per_iter * 8 / per_iter, ms);
}

int main() {
    float ha[64], hb[64], hd[32 * 64];
    for (int l = 0; l < 64; ++l) { ha[l] = 1.f; hb[l] = 1000.f + l; }
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, sizeof(hd));
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    sem<<<1, 64>>>(a, b, d);
    hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost);
    for (int s = 0; s < 8; ++s) {
        printf("blgp %d: B lane used by output lane 0,5,16,21,32,37,48,53:", s);
        const int ls[8] = {0, 5, 16, 21, 32, 37, 48, 53};
        for (int k = 0; k < 8; ++k) printf(" %d", (int)(hd[(s * 4 + 0) * 64 + ls[k]] - 1000.f));
        printf("\n");
    }
    float *out, *gsrc; unsigned long long* cyc;
    hipMalloc(&out, 256 * 4 * 64 * 4); hipMalloc(&gsrc, (size_t)257 << 20); hipMalloc(&cyc, 64);
    hipMemset(gsrc, 0, (size_t)257 << 20);
    // (iters is the exit count of the unrolled bodies: run() passes a count larger than the body count so that every body runs)
    for (int helpers = 0; helpers <= 4; helpers += 4) {
        run<0, 4>("dense (12 bodies)", 12 * 16, helpers, out, gsrc, cyc);
        run<1, 4>("sparse (48 bodies)", 48 * 4, helpers, out, gsrc, cyc);
        run<1, 6>("sparse (48 bodies)", 48 * 4, helpers, out, gsrc, cyc);
        run<2, 4>("mixed (12 dense + 24 sparse)", 12 * 16 + 24 * 4, helpers, out, gsrc, cyc);
    }
    return 0;
}

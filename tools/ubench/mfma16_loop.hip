// Issue rate of v_mfma_f32_16x16x4_f32 in the reservoir kernel's pattern (gfx950): 4 interleaved
// accumulators, weight fragments from LDS (one ds_read_b128 per 4 MFMAs), optional VALU filler.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma16_loop mfma16_loop.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// VARIANT bit0: weight reads from LDS   bit1: 64 VALU fmas per 64 MFMAs   bit2: transcendental filler
template <int VARIANT>
__global__ __launch_bounds__(256, 4) void k(float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1e-3f * (i % 97);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0, 0, 0, 0};
    f32x4 h[4];
    for (int j = 0; j < 4; ++j) h[j] = f32x4{1e-3f * lane, 0.5f, 0.25f, 0.125f};
    float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    for (int it = 0; it < iters; ++it) {
        int wo = 0;
        asm volatile("" : "+v"(wo));
        const float* w = lds + wo;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 wf[4];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                if (VARIANT & 1) wf[jt] = *reinterpret_cast<const f32x4*>(w + ((jt * 4 + kb) * 64 + lane) * 4);
                else wf[jt] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
                    acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[jt][s], h[kb][s], acc[jt], 0, 0, 0);
        }
        if (VARIANT & 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = fmaf(v[u], 0.999f, 0.001f);
        }
        if (VARIANT & 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[j][r] = __builtin_amdgcn_rcpf(__expf(acc[j][r]) + 1.f);
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j].x + acc[j].y + acc[j].z + acc[j].w;
    for (int u = 0; u < 8; ++u) s += v[u];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VARIANT>
void run(const char* name, int threads) {
    const int blocks = 1024, iters = 2000;     // 4 workgroups of 256 per CU -> 4 waves per SIMD
    float* sink; (void)hipMalloc(&sink, (size_t)blocks * threads * 4);
    (void)hipFuncSetAttribute((const void*)k<VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<VARIANT>, dim3(blocks), dim3(threads), 65536 / 2, 0, sink, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<VARIANT>, dim3(blocks), dim3(threads), 65536 / 2, 0, sink, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)blocks * (threads / 64) * iters * 64 / 1024.0;
    printf("%-44s %.3f ms  %.1f ns per MFMA per SIMD  (%.1f TF/s)\n", name, ms, ms * 1e6 / mfma_per_simd,
           (double)blocks * (threads / 64) * iters * 64 * 2048 / (ms * 1e-3) / 1e12);
    (void)hipFree(sink);
}

int main() {
    run<0>("mfma 16x16x4 only", 256);
    run<1>("+ weight fragments from LDS", 256);
    run<3>("+ LDS + 64 VALU fma per 64 MFMA", 256);
    run<5>("+ LDS + 16 exp/rcp per 64 MFMA", 256);
    run<7>("+ LDS + VALU + exp/rcp", 256);
    return 0;
}

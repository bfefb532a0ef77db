// Does the lane -> address map of a 16-byte-per-lane global load / store matter when the wave covers the same 4 KB?
// Each wave copies tiles of 16 rows x 256 bytes, step after step (the reservoir's access pattern):
//   map 0: lane (n = l & 15, q = l >> 4), instruction k: row n, bytes 64 k + 16 q   (the MFMA accumulator layout)
//   map 1: lane l, instruction k: bytes 1024 k + 16 l of the tile                   (16 consecutive lanes = one row)
//   map 2: as 0 for the loads, as 1 for the stores;  map 3: the reverse
// build: hipcc --offload-arch=gfx950 -O3 -o lane_map_copy lane_map_copy.hip ; run: ./lane_map_copy
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MAP>
__global__ __launch_bounds__(1024) void copy_tiles(const float* __restrict__ x, float* __restrict__ y, int T, long long step_floats, int tiles_per_wave) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int n = lane & 15, q = lane >> 4;
    const bool lrow = MAP == 1 || MAP == 3, srow = MAP == 1 || MAP == 2;
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < tiles_per_wave; ++i) {
            const long long base = (long long)t * step_floats + ((long long)wave * tiles_per_wave + i) * 1024;
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *(const f32x4*)(x + base + (lrow ? 256 * k + 4 * lane : 64 * n + 16 * k + 4 * q));
#pragma unroll
            for (int k = 0; k < 4; ++k) *(f32x4*)(y + base + (srow ? 256 * k + 4 * lane : 64 * n + 16 * k + 4 * q)) = v[k];
        }
}

int main() {
    const int T = 128, tiles = 6144;                 // 6144 tiles of 16 nodes, 64 features
    const long long step = (long long)tiles * 1024;
    float *x, *y;
    hipMalloc(&x, step * T * 4); hipMalloc(&y, step * T * 4);
    hipMemset(x, 0, step * T * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int map = 0; map < 4; ++map)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            // 256 workgroups x 16 waves, waves 0-7 of a SIMD pair... : 4096 waves, 1.5 tiles each -> use 3072 waves x 2 tiles
            if (map == 0) hipLaunchKernelGGL(copy_tiles<0>, dim3(256), dim3(768), 0, 0, x, y, T, step, 2);
            if (map == 1) hipLaunchKernelGGL(copy_tiles<1>, dim3(256), dim3(768), 0, 0, x, y, T, step, 2);
            if (map == 2) hipLaunchKernelGGL(copy_tiles<2>, dim3(256), dim3(768), 0, 0, x, y, T, step, 2);
            if (map == 3) hipLaunchKernelGGL(copy_tiles<3>, dim3(256), dim3(768), 0, 0, x, y, T, step, 2);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("map %d: %.3f ms per %d steps, %.2f TB/s (read + write)\n", map, ms, T, 2.0 * step * T * 4 / ms / 1e9);
        }
    return 0;
}

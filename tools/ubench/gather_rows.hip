// What can 1e7 x T independent 256-byte row gathers reach on this part?  (SURVEY 8d's adversarial graph: 100 uniformly random
// columns per row, N = 100 000, D = 64: every (edge, step) pair reads its own 256-byte row; the step's slab is 25.6 MB -- six
// times an XCD's L2, a tenth of the Infinity Cache.)  A wave owns rows; 16 lanes gather one source row per load (16 B per
// lane), 4 rows per wave instruction, sums them and writes one 256-byte result row -- the generic CSR kernel without its
// weights, i.e. the upper bound of any kernel that gathers per (edge, step) without sharing through LDS.
//   order 0: every wave walks its rows' edge lists as the graph gives them (random columns)
//   order 1: the same edges SORTED by column inside a workgroup's row range -- what a column-blocked schedule buys at best
//   order 2: columns = the row's own neighbourhood (row + 0..99): the L1 / L2 serve nearly everything (upper bound of the path)
// build: hipcc --offload-arch=gfx950 -O3 -o gather_rows gather_rows.hip ; run: ./gather_rows
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave per result row: lanes (g = l >> 4, i = l & 15): edge e + g, bytes 16 i of the source row
__global__ __launch_bounds__(256) void gather(const int* __restrict__ col, const float* __restrict__ x, float* __restrict__ y,
                                              int n, int deg, int T, long long step) {
    const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int* c = col + (long long)row * deg;
    for (int t = 0; t < T; ++t) {
        const float* xs = x + (long long)t * step;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int e = 0; e < deg; e += 16) {                   // 4 wave instructions = 16 rows in flight
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *(const f32x4*)(xs + (long long)c[e + 4 * k + g] * 64 + 4 * i);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += v[k];
        }
        acc[0] += __shfl_xor(acc[0], 16); acc[1] += __shfl_xor(acc[1], 16); acc[2] += __shfl_xor(acc[2], 16); acc[3] += __shfl_xor(acc[3], 16);
        acc[0] += __shfl_xor(acc[0], 32); acc[1] += __shfl_xor(acc[1], 32); acc[2] += __shfl_xor(acc[2], 32); acc[3] += __shfl_xor(acc[3], 32);
        if (g == 0) *(f32x4*)(y + (long long)t * step + (long long)row * 64 + 4 * i) = acc;
    }
}

int main() {
    const int n = 100000, deg = 112, T = 64;                  // 112 = 100 rounded up to whole 16-edge batches
    const long long step = (long long)n * 64;
    std::vector<int> col((size_t)n * deg);
    float *x, *y; int* dcol;
    hipMalloc(&x, step * T * 4); hipMalloc(&y, step * T * 4); hipMalloc(&dcol, col.size() * 4);
    hipMemset(x, 0, step * T * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int order = 0; order < 3; ++order) {
        srand(1);
        for (int r = 0; r < n; ++r)
            for (int e = 0; e < deg; ++e)
                col[(size_t)r * deg + e] = order == 2 ? (r + e) % n : (int)(((long long)rand() * 32768 + rand()) % n);
        if (order == 1)                                        // a workgroup's 4 rows x deg edges sorted by column
            for (int r = 0; r + 4 <= n; r += 4) std::sort(col.begin() + (size_t)r * deg, col.begin() + (size_t)(r + 4) * deg);
        hipMemcpy(dcol, col.data(), col.size() * 4, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(gather, dim3((n + 3) / 4), dim3(256), 0, 0, dcol, x, y, n, deg, T, step);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("order %d: %.2f ms per %d steps = %.2f ms per 256 steps; gathers %.2f TB/s; algorithmic (x + y once) %.3f of 8 TB/s\n",
                            order, ms, T, ms * 256 / T, (double)n * deg * 256 * T / ms / 1e9, 2.0 * step * 4 * T / (ms * 1e-3) / 8e12);
        }
    }
    return 0;
}

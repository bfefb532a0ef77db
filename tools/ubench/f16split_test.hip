// Round-4 feasibility checks for the split-fp16 hop, run on the GPU box:
//   1. ds_read_b64_tr_b16 semantics with per-lane ROW addresses (rows 32 B apart or anywhere);
//   2. operand / result lane maps of v_mfma_f32_16x16x32_f16;
//   3. accuracy of x = hi + lo (two fp16 pieces of the scaled value), three products hi*hi + hi*lo + lo*hi
//      accumulated in fp32, against fp64 and against an fp32 fma chain; fp16 subnormal inputs;
//   4. issue rate of the inner loop shape (4 transpose reads + 6 MFMAs per 32-column chunk, 2 waves per SIMD).
// hipcc --offload-arch=gfx950 -O3 -o f16split_test f16split_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
#define LDSP(p) ((s4 __attribute__((address_space(3)))*)(p))

__global__ void k_tr(const _Float16* in, const int* addr, _Float16* out) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = in[i];
    __syncthreads();
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP((char*)lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = ((_Float16*)&v)[j];
}

__global__ void k_mfma(const h8* a, const h8* b, f4* d) {
    f4 z = {0, 0, 0, 0};
    d[threadIdx.x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[threadIdx.x], b[threadIdx.x], z, 0, 0, 0);
}

// D[16 x 16] = W[16 x K] X[K x 16] through the split products; one wave, K / 32 chunks.
__global__ void k_split(const float* w, const float* x, float* d, int K, float sw, float sx) {
    const int l = threadIdx.x, m = l & 15, g = l >> 4;
    f4 acc = {0, 0, 0, 0};
    for (int c = 0; c < K / 32; ++c) {
        h8 ah, al, bh, bl;
        for (int e = 0; e < 8; ++e) {
            const int kk = c * 32 + g * 8 + e;
            const float wv = w[m * K + kk] * sw, xv = x[kk * 16 + m] * sx;
            const _Float16 wh = (_Float16)wv, xh = (_Float16)xv;
            ah[e] = wh; al[e] = (_Float16)(wv - (float)wh);
            bh[e] = xh; bl[e] = (_Float16)(xv - (float)xh);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
    }
    const float inv = 1.f / (sw * sx);
    for (int r = 0; r < 4; ++r) d[(g * 4 + r) * 16 + m] = acc[r] * inv;
}

// Loop shape of the hop: NCH chunks per unit, per chunk 4 transpose reads (hi / lo x two 4-row sets) and
// 6 MFMAs (two 16-row halves x three products); A fragments resident (16 VGPRs per chunk).
template <int NCH, int MODE>
__global__ __launch_bounds__(512, 2) void k_loop(const h8* afr, const int* addr, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 96 * 1024 / 4; i += 512) ((float*)lds)[i] = 1e-3f * (i & 1023);
    __syncthreads();
    h8 a[NCH][4];
    int ad[NCH][2];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[c][q] = afr[(c * 4 + q) * 512 + threadIdx.x];
        ad[c][0] = addr[(c * 2 + 0) * 512 + threadIdx.x];
        ad[c][1] = addr[(c * 2 + 1) * 512 + threadIdx.x];
    }
    f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            h8 bh, bl;
            if (MODE & 1) {
                s4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(lds + ad[c][0]));
                s4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(lds + ad[c][1]));
                s4 r2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(lds + ad[c][0] + 49152));
                s4 r3 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(lds + ad[c][1] + 49152));
                typedef short s8 __attribute__((ext_vector_type(8)));
                s8 h = __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7);
                s8 lo = __builtin_shufflevector(r2, r3, 0, 1, 2, 3, 4, 5, 6, 7);
                bh = __builtin_bit_cast(h8, h); bl = __builtin_bit_cast(h8, lo);
            } else {
                bh = a[c][1]; bl = a[c][3];
            }
            if (MODE & 2) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c][0], bh, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c][2], bh, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c][0], bl, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c][2], bl, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c][1], bh, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c][3], bh, acc1, 0, 0, 0);
            } else {
                acc0[0] += (float)bh[0] + (float)bl[3]; acc1[1] += (float)bh[5] + (float)bl[7];
            }
        }
        if (MODE & 4) __builtin_amdgcn_s_barrier();
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
}

template <int NCH, int MODE>
static void time_loop(const char* name, const h8* afr, const int* addr, float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k_loop<NCH, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    k_loop<NCH, MODE><<<256, 512, 96 * 1024>>>(afr, addr, out, 16);
    hipEventRecord(e0);
    k_loop<NCH, MODE><<<256, 512, 96 * 1024>>>(afr, addr, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_chunk_ns = ms * 1e6 / ((double)iters * NCH);
    printf("%-34s %8.3f ms  %7.1f ns per chunk per wave-pair  (MFMA floor at 2.4 GHz: %.1f ns for 2 waves x 6 MFMAs x 16 cyc)\n",
           name, ms, per_chunk_ns, 2 * 6 * 16 / 2.4);
}

int main() {
    int bad = 0;
    // ---- 1. transpose read
    {
        std::vector<_Float16> hin(8192); for (int i = 0; i < 8192; ++i) hin[i] = (_Float16)(float)(i % 2048);
        _Float16 *din, *dout; int* daddr; hipMalloc(&din, 16384); hipMalloc(&dout, 512); hipMalloc(&daddr, 256);
        hipMemcpy(din, hin.data(), 16384, hipMemcpyHostToDevice);
        std::vector<int> addr(64); std::vector<_Float16> ho(256);
        // (a) canonical: lane l points at 8 * l bytes
        for (int l = 0; l < 64; ++l) addr[l] = 8 * l;
        hipMemcpy(daddr, addr.data(), 256, hipMemcpyHostToDevice);
        k_tr<<<1, 64>>>(din, daddr, dout); hipMemcpy(ho.data(), dout, 512, hipMemcpyDeviceToHost);
        int b1 = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
            const int want = (l & 15) + j * 16 + (l >> 4) * 64;
            if ((int)(float)ho[l * 4 + j] != want) { if (b1 < 6) printf("tr canonical l=%d j=%d got %d want %d\n", l, j, (int)(float)ho[l * 4 + j], want); ++b1; }
        }
        // (b) rows anywhere: lane i of group g points at rowbase[g][i / 4] + 8 * (i % 4); expect lane c, elem j = row j, column c
        int rowb[4][4]; std::mt19937 rng(5);
        for (int g = 0; g < 4; ++g) for (int r = 0; r < 4; ++r) rowb[g][r] = 32 * (int)(rng() % 120);
        for (int l = 0; l < 64; ++l) addr[l] = rowb[l >> 4][(l & 15) >> 2] + 8 * (l & 3);
        hipMemcpy(daddr, addr.data(), 256, hipMemcpyHostToDevice);
        k_tr<<<1, 64>>>(din, daddr, dout); hipMemcpy(ho.data(), dout, 512, hipMemcpyDeviceToHost);
        int b2 = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
            const int want = (rowb[l >> 4][j] / 2 + (l & 15)) % 2048;
            if ((int)(float)ho[l * 4 + j] != want) { if (b2 < 6) printf("tr rows l=%d j=%d got %d want %d\n", l, j, (int)(float)ho[l * 4 + j], want); ++b2; }
        }
        printf("1. ds_read_b64_tr_b16: canonical mismatches %d, free-row mismatches %d\n", b1, b2);
        bad += b1 + b2;
    }
    // ---- 2. MFMA lane maps
    {
        std::mt19937 rng(7);
        std::vector<float> A(16 * 32), B(32 * 16);
        for (auto& v : A) v = (float)((int)(rng() % 17) - 8);
        for (auto& v : B) v = (float)((int)(rng() % 13) - 6);
        std::vector<_Float16> ha(64 * 8), hb(64 * 8);
        for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
            ha[l * 8 + e] = (_Float16)A[(l & 15) * 32 + 8 * (l >> 4) + e];
            hb[l * 8 + e] = (_Float16)B[(8 * (l >> 4) + e) * 16 + (l & 15)];
        }
        h8 *da, *db; f4* dd; hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 1024);
        hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
        k_mfma<<<1, 64>>>(da, db, dd);
        std::vector<float> hd(256); hipMemcpy(hd.data(), dd, 1024, hipMemcpyDeviceToHost);
        int b = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const int row = 4 * (l >> 4) + r, col = l & 15; float want = 0;
            for (int kk = 0; kk < 32; ++kk) want += A[row * 32 + kk] * B[kk * 16 + col];
            if (hd[l * 4 + r] != want) { if (b < 6) printf("mfma l=%d r=%d got %g want %g\n", l, r, hd[l * 4 + r], want); ++b; }
        }
        printf("2. v_mfma_f32_16x16x32_f16 lane map mismatches %d\n", b);
        bad += b;
    }
    // ---- 3. accuracy of the split products
    {
        const int K = 128; std::mt19937 rng(11); std::uniform_real_distribution<float> U(-1.f, 1.f), P(0.f, 1.f);
        double worst_split = 0, worst_f32 = 0, worst_split_small = 0;
        float *dw, *dx, *dd; hipMalloc(&dw, 16 * K * 4); hipMalloc(&dx, K * 16 * 4); hipMalloc(&dd, 1024);
        for (int rep = 0; rep < 200; ++rep) {
            std::vector<float> w(16 * K), x(K * 16);
            const float xs = rep < 100 ? 1.f : 1e-3f;                  // small-magnitude inputs too
            for (int m = 0; m < 16; ++m) { double s = 0; for (int kk = 0; kk < K; ++kk) { w[m * K + kk] = P(rng) * P(rng); s += w[m * K + kk]; }
                for (int kk = 0; kk < K; ++kk) w[m * K + kk] = (float)(w[m * K + kk] / s); }
            for (auto& v : x) v = std::tanh(2.f * U(rng)) * xs;
            hipMemcpy(dw, w.data(), 16 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), K * 16 * 4, hipMemcpyHostToDevice);
            float wmax = 0; for (auto v : w) wmax = std::fmax(wmax, v);
            const float sw = std::exp2(std::floor(std::log2(16384.f / wmax))), sx = 4096.f;
            k_split<<<1, 64>>>(dw, dx, dd, K, sw, sx);
            std::vector<float> d(256); hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
            for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
                double ex = 0; float f = 0;
                for (int kk = 0; kk < K; ++kk) { ex += (double)w[m * K + kk] * x[kk * 16 + n]; f = std::fmaf(w[m * K + kk], x[kk * 16 + n], f); }
                const double es = std::fabs(d[m * 16 + n] - ex) / xs, ef = std::fabs(f - ex) / xs;
                if (rep < 100) { worst_split = std::fmax(worst_split, es); } else worst_split_small = std::fmax(worst_split_small, es);
                worst_f32 = std::fmax(worst_f32, ef);
            }
        }
        printf("3. |error| vs fp64, relative to the input scale: split-fp16 %.3e (inputs ~1), %.3e (inputs ~1e-3); fp32 fma chain %.3e\n",
               worst_split, worst_split_small, worst_f32);
        // subnormal fp16 operands: 2^-20 * 1
        std::vector<float> w(16 * 32, 0.f), x(32 * 16, 0.f); w[0] = 9.5367431640625e-07f; x[0] = 1.f;
        hipMemcpy(dw, w.data(), 16 * 32 * 4, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), 32 * 16 * 4, hipMemcpyHostToDevice);
        k_split<<<1, 64>>>(dw, dx, dd, 32, 1.f, 1.f);
        float r; hipMemcpy(&r, dd, 4, hipMemcpyDeviceToHost);
        printf("   fp16 subnormal operand 2^-20 x 1 -> %g (%s)\n", r, r != 0.f ? "kept" : "FLUSHED");
        if (worst_split > 2e-6) ++bad;
    }
    // ---- 4. loop rate
    {
        const int NCH = 8;
        std::vector<_Float16> af((size_t)NCH * 4 * 512 * 8); for (size_t i = 0; i < af.size(); ++i) af[i] = (_Float16)(0.001f * (i % 97));
        std::vector<int> adl(NCH * 2 * 512), adr(NCH * 2 * 512); std::mt19937 rng(3);
        for (int c = 0; c < NCH; ++c) for (int s = 0; s < 2; ++s) for (int t = 0; t < 512; ++t) {
            const int l = t & 63, w = t >> 6, g = l >> 4, i = l & 15;
            const int krow = g * 8 + s * 4 + (i >> 2);                  // 0..31 inside the chunk
            adl[(c * 2 + s) * 512 + t] = ((w * 96 + c * 32 + krow) % 1536) * 32 + 8 * (i & 3);     // consecutive staged rows
        }
        for (int c = 0; c < NCH; ++c) for (int w = 0; w < 8; ++w) {
            int rows[32]; for (int r = 0; r < 32; ++r) rows[r] = rng() % 1536;                       // random staged rows
            for (int s = 0; s < 2; ++s) for (int l = 0; l < 64; ++l) {
                const int g = l >> 4, i = l & 15;
                adr[(c * 2 + s) * 512 + w * 64 + l] = rows[g * 8 + s * 4 + (i >> 2)] * 32 + 8 * (i & 3);
            }
        }
        h8* dafr; int *dl, *dr; float* dout;
        hipMalloc(&dafr, af.size() * 2); hipMalloc(&dl, adl.size() * 4); hipMalloc(&dr, adr.size() * 4); hipMalloc(&dout, 256 * 512 * 4);
        hipMemcpy(dafr, af.data(), af.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dl, adl.data(), adl.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dr, adr.data(), adr.size() * 4, hipMemcpyHostToDevice);
        const int iters = 20000;
        printf("4. loop shape, %d chunks, 256 workgroups x 8 waves (2 per SIMD), %d iterations\n", NCH, iters);
        time_loop<NCH, 2>("MFMAs only", dafr, dl, dout, iters);
        time_loop<NCH, 1>("transpose reads only, consecutive", dafr, dl, dout, iters);
        time_loop<NCH, 1>("transpose reads only, random rows", dafr, dr, dout, iters);
        time_loop<NCH, 3>("reads + MFMAs, consecutive rows", dafr, dl, dout, iters);
        time_loop<NCH, 3>("reads + MFMAs, random rows", dafr, dr, dout, iters);
        time_loop<NCH, 7>("reads + MFMAs + barrier, random", dafr, dr, dout, iters);
    }
    printf("%s\n", bad ? "FAILED" : "ok");
    return bad != 0;
}

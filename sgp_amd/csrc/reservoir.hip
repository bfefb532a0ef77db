// C ABI of the reservoir layer + weight packing.  Kernels: reservoir_impl.h, instantiated
// per reservoir width in reservoir_jt*.hip (separate translation units build in parallel).
#include "reservoir_impl.h"

namespace sgp_res {
int launch_jt1(const ResArgs&, int, hipStream_t);
int launch_jt2(const ResArgs&, int, hipStream_t);
int launch_jt4(const ResArgs&, int, hipStream_t);
int launch_jt8(const ResArgs&, int, hipStream_t);
int launch_jt16(const ResArgs&, int, hipStream_t);
}

namespace {
using namespace sgp_res;
__global__ void pack_weights(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                             const float* __restrict__ b, float* __restrict__ out,
                             int F, int R, int JT, int NKX) {
    const long long n_bias = (long long)JT * 16;
    const long long n_wx = (long long)JT * NKX * 64;
    const long long total = packed_floats(JT, NKX);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < n_bias) {
            const int j = (int)i;
            v = j < R ? b[j] : 0.f;
        } else if (i < n_bias + n_wx) {
            const long long o = i - n_bias;
            int l, ks, jt;
            if (NKX % 4 == 0) {                      // [JT][NKX/4][64][4]
                const int s = (int)(o & 3);
                l = (int)((o >> 2) & 63);
                const int k4 = (int)((o >> 8) % (NKX / 4));
                jt = (int)((o >> 8) / (NKX / 4));
                ks = 4 * k4 + s;
            } else {                                 // [JT][NKX][64]
                l = (int)(o & 63);
                ks = (int)((o >> 6) % NKX);
                jt = (int)((o >> 6) / NKX);
            }
            const int j = 16 * jt + (l & 15);
            const int k = (l >> 4) * NKX + ks;
            v = (j < R && k < F) ? w_ih[(long long)j * F + k] : 0.f;
        } else {
            const long long o = i - n_bias - n_wx;
            const int s = (int)(o & 3);
            const int l = (int)((o >> 2) & 63);
            const int kb = (int)((o >> 8) % JT);
            const int jt = (int)((o >> 8) / JT);
            const int j = 16 * jt + (l & 15);
            const int k = 16 * kb + 4 * (l >> 4) + s;
            v = (j < R && k < R) ? w_hh[(long long)j * R + k] : 0.f;
        }
        out[i] = v;
    }
}

int pick_nkx(int F) {
    const int need = (F + 3) / 4;
    const int opts[] = {1, 2, 4, 8, 16, 32, 64};
    for (int o : opts) if (o >= need) return o;
    return 0;
}
int pick_jt(int R) {
    const int need = (R + 15) / 16;
    const int opts[] = {1, 2, 4, 8, 16};
    for (int o : opts) if (o >= need) return o;
    return 0;
}

}  // namespace


extern "C" {

int64_t sgp_reservoir_workspace_bytes(int32_t F, int32_t R) {
    const int jt = pick_jt(R), nkx = pick_nkx(F);
    if (!jt || !nkx) return -1;
    return packed_floats(jt, nkx) * 4;
}

int sgp_reservoir_f32(const float* x, int64_t xrs, int64_t xss,
                      const float* w_ih, const float* w_hh, const float* b,
                      double alpha, int32_t act,
                      float* out, int64_t ors, int64_t oss,
                      float* h_state, void* workspace,
                      int32_t T, int32_t N, int32_t F, int32_t R,
                      sgp_stream_t stream) {
    SGP_REQUIRE(x && w_ih && w_hh && b && out && workspace, "sgp_reservoir_f32: null pointer");
    SGP_REQUIRE(T >= 0 && N >= 0 && F > 0 && R > 0, "sgp_reservoir_f32: bad size");
    SGP_REQUIRE(act >= SGP_ACT_TANH && act <= SGP_ACT_IDENTITY, "sgp_reservoir_f32: unknown activation %d", act);
    SGP_REQUIRE(sgp::aligned16(workspace), "sgp_reservoir_f32: workspace must be 16-byte aligned");
    if (T == 0 || N == 0) return 0;
    const int jt = pick_jt(R), nkx = pick_nkx(F);
    if (!jt) return sgp::fail(SGP_EUNSUP, "sgp_reservoir_f32: reservoir size %d > 256 not supported", R);
    if (!nkx) return sgp::fail(SGP_EUNSUP, "sgp_reservoir_f32: input size %d > 256 not supported", F);
    hipStream_t s = (hipStream_t)stream;
    const long long total = packed_floats(jt, nkx);
    int pg = (int)((total + 255) / 256);
    if (pg > 1024) pg = 1024;
    hipLaunchKernelGGL(pack_weights, dim3(pg), dim3(256), 0, s, w_ih, w_hh, b, (float*)workspace, F, R, jt, nkx);
    int rc = sgp::check_launch("pack_weights");
    if (rc) return rc;

    ResArgs a;
    a.x = x; a.xrs = xrs; a.xss = xss;
    a.wp = (const float*)workspace;
    a.out = out; a.ors = ors; a.oss = oss;
    a.h_state = h_state;
    a.alpha = (float)alpha;                      // scalar operands are rounded to fp32 like
    a.one_minus_alpha = (float)(1.0 - alpha);    // torch does for `(1 - alpha) * h` (reservoir.py:80)
    a.act = act; a.T = T; a.N = N; a.F = F; a.R = R;
    a.tiles_per_wave = 0; a.n_tiles = 0;
    switch (jt) {
        case 1: return launch_jt1(a, nkx, s);
        case 2: return launch_jt2(a, nkx, s);
        case 4: return launch_jt4(a, nkx, s);
        case 8: return launch_jt8(a, nkx, s);
        case 16: return launch_jt16(a, nkx, s);
    }
    return sgp::fail(SGP_EUNSUP, "sgp_reservoir_f32: unreachable");
}

}  // extern "C"

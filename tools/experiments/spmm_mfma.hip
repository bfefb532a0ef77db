// Round-1 row-group SpMM on the fp32 matrix cores with a single stage buffer (sgp_spmm_mfma_f32), retired in
// round 4: superseded by spmm_res / spmm_mix (same arithmetic, two-phase staging, register-resident streams)
// on every measured workload.  Kept as a record: the kernel and its C entry as they stood inside
// sgp_amd/csrc/spmm.hip (they use that file's Src / TiledArgs helpers; not built).

// ---------------------------------------------------------------- MFMA row-group kernel
// Same tile staging as spmm_tiled, different inner product: a wave owns FOUR consecutive output
// rows ("row group") and walks the sorted union of their columns.  v_mfma_f32_4x4x1_16b_f32
// performs 16 independent 4x1 (x) 1x4 outer products per instruction: A = the 4 rows' weights for
// one source column, B = 4 features of that column's staged row.  Exact fp32 FMAs, so numerics
// equal the VALU kernels.
//
// Lane l = (q = l >> 4, li = l & 15).  The group's columns are dealt round-robin to 4 classes q;
// per "super-step" every class fetches 16 bytes (features 4 li .. 4 li + 3) of ITS column with one
// ds_read_b128 -- full LDS rate, 4 different staged rows per wave instruction, conflict-free --
// and issues 4 MFMAs (one per feature m of the 16 bytes): MFMA block b = l >> 2 pairs the weights
// w[row i = l & 3] of class q's column with feature 4 li + m.  Accumulator m, register i, lane l
// = partial y[row i][4 li + m] over class q's columns; once per time step the 4 classes are
// summed across lanes (l ^ 16, l ^ 32) and class q stores row q as one float4 per lane.
// Weights and row indices are an LDS-resident copy of the group's stream, read 4 super-steps at
// a time (one ds_read_b128 + one ds_read_b64 per 16 MFMAs); the VALU only forms addresses.
//   gw   [quad][q][i][4]  float   weight of row i for class q's column in super-steps 4*quad + 0..3
//   gidx [quad][q][4]     int32   LDS byte offset (index in the tile's staged list * 256) of that column
// (0 / weight 0 padding), gptr[16 * tile + g] .. = quad range of group g.
struct MfmaArgs {
    const int* trow; const int* uptr; const int* ucol;
    const int* gptr; const int* gidx; const float* gw; const int* rowmap;
    int n_tiles;
    Src src;
    float* Y; long long yrs, ybs;
    int n_rows, batch, feat;
    int t_chunk, n_tchunks;
};

constexpr int kMfmaPasses = 7;                       // 448 staged rows
constexpr int kMfmaStageBytes = kMfmaPasses * 64 * 256;
constexpr int kMfmaQuadBytes = 256 + 64;             // weights + row offsets of 4 super-steps
constexpr int kMfmaMaxQuads = (160 * 1024 - kMfmaStageBytes) / kMfmaQuadBytes;

template <bool HALO, int ABL = 0>
__global__ __launch_bounds__(1024) void spmm_mfma(MfmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int PASSES = kMfmaPasses;
    constexpr int FT = 64;
    constexpr int RPP = 64;

    const int nwg = a.n_tiles * a.n_tchunks;
    const int orig = blockIdx.x;
    const int qq = nwg >> 3, rr = nwg & 7, xcd = orig & 7;
    const int w = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
    const int tile = w % a.n_tiles;
    const int tchunk = w / a.n_tiles;
    const int f_base = blockIdx.y * FT;

    const int tid = threadIdx.x;
    const int li = tid & 15;
    const int eg = tid >> 4;
    const int u0 = a.uptr[tile];
    const int nU = a.uptr[tile + 1] - u0;

    int soff[PASSES];
    unsigned halo_mask = 0;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int u = p * RPP + eg;
        const int c = (u < nU) ? a.ucol[u0 + u] : 0;
        if (HALO && c >= a.src.n_own) {
            halo_mask |= 1u << p;
            soff[p] = (c - a.src.n_own) * (int)a.src.xhrs;
        } else {
            soff[p] = c * (int)a.src.xrs;
        }
    }
    const int n_pass = (nU + RPP - 1) / RPP;

    const int t_begin = tchunk * a.t_chunk;
    const int t_end = min(a.batch, t_begin + a.t_chunk);
    if (t_begin >= t_end) return;

    // the tile's stream -> LDS (once per workgroup): weights then indices
    const int tile_q0 = a.gptr[tile * 16], tile_q1 = a.gptr[tile * 16 + 16];
    const int tile_quads = tile_q1 - tile_q0;
    char* wlds = lds + kMfmaStageBytes;
    char* ilds = wlds + tile_quads * 256;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.gw) + (long long)tile_q0 * 16;
        f32x4* dst = reinterpret_cast<f32x4*>(wlds);
        for (int i = tid; i < tile_quads * 16; i += 1024) dst[i] = src[i];
        const f32x4* isrc = reinterpret_cast<const f32x4*>(a.gidx) + (long long)tile_q0 * 4;
        f32x4* idst = reinterpret_cast<f32x4*>(ilds);
        for (int i = tid; i < tile_quads * 4; i += 1024) idst[i] = isrc[i];
    }

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // scalar: branches stay uniform
    const int lane = tid & 63;
    const int q = lane >> 4;
    const int grp = tile * 16 + wave;
    const int q_begin = __builtin_amdgcn_readfirstlane(a.gptr[grp]) - tile_q0;
    const int n_quads = __builtin_amdgcn_readfirstlane(a.gptr[grp + 1]) - (q_begin + tile_q0);
    const int my_row = a.rowmap[tile * 64 + wave * 4 + q];   // output row of class q (-1: none)
    const char* wmine = wlds + q_begin * 256 + (q * 4 + (lane & 3)) * 16;
    const char* imine = ilds + q_begin * 64 + q * 16;
    const char* xmine = lds + li * 16;

    f32x4 stage[PASSES];
    // one staged row of step t per call: the prefetch is spread over the quad loop so the 7
    // global loads per thread never queue up in front of the compute (in-order issue)
    auto issue_one = [&](int t, int p) {
        if constexpr (ABL >= 4) return;
        if (p < n_pass) {
            const bool far = HALO && ((halo_mask >> p) & 1u);
            const float* b = far ? a.src.xh + (long long)t * a.src.xhbs
                                 : a.src.x + (long long)t * a.src.xbs;
            stage[p] = ld4(b + f_base + li * 4 + soff[p]);
        }
    };
#pragma unroll
    for (int p = 0; p < PASSES; ++p) issue_one(t_begin, p);

    for (int t = t_begin; t < t_end; ++t) {
        if constexpr (ABL < 4) {
#pragma unroll
            for (int p = 0; p < PASSES; ++p)
                if (p < n_pass)
                    *reinterpret_cast<f32x4*>(lds + ((p * RPP + eg) * FT + li * 4) * 4) = stage[p];
        }
        if constexpr (ABL != 9) __syncthreads();
        const bool more = t + 1 < t_end;

        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
        f32x4 wn = f32x4{0.f, 0.f, 0.f, 0.f};
        int4 in = int4{0, 0, 0, 0};
        if (n_quads > 0) {
            wn = *reinterpret_cast<const f32x4*>(wmine);
            in = *reinterpret_cast<const int4*>(imine);
        }
#define SGP_SUPER(W, IDX)                                                                       \
        {                                                                                       \
            f32x4 xv;                                                                           \
            if constexpr (ABL == 6) { const float f = __int_as_float((int)(IDX)); xv = f32x4{f, f, f, f}; } \
            else xv = *reinterpret_cast<const f32x4*>(xmine + (IDX));                           \
            if constexpr (ABL == 5) {                                                           \
                asm volatile("" :: "v"(xv.x), "v"(xv.y), "v"(xv.z), "v"(xv.w), "v"(W));        \
            } else {                                                                            \
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(W, xv.x, acc0, 0, 0, 0);              \
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(W, xv.y, acc1, 0, 0, 0);              \
                acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(W, xv.z, acc2, 0, 0, 0);              \
                acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(W, xv.w, acc3, 0, 0, 0);              \
            }                                                                                   \
        }
#define SGP_QUAD(C)                                                                             \
        {                                                                                       \
            const f32x4 wv = wn;                       /* the next quad's stream is fetched */  \
            const int4 ix = in;                        /* under this quad's MFMAs */            \
            if ((C) + 1 < n_quads) {                                                            \
                wn = *reinterpret_cast<const f32x4*>(wmine + ((C) + 1) * 256);                  \
                in = *reinterpret_cast<const int4*>(imine + ((C) + 1) * 64);                    \
            }                                                                                   \
            /* padded super-steps carry zero weights: no branch, the MFMA adds 0 */             \
            SGP_SUPER(wv.x, ix.x)                                                               \
            SGP_SUPER(wv.y, ix.y)                                                               \
            SGP_SUPER(wv.z, ix.z)                                                               \
            SGP_SUPER(wv.w, ix.w)                                                               \
        }
        // first PASSES quads carry one prefetch load each; the rest run in a plain loop
        if constexpr (ABL != 8 && ABL != 9) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            if (more) issue_one(t + 1, p);
            if (p < n_quads) SGP_QUAD(p)
        }
        for (int c = PASSES; c < n_quads; ++c) SGP_QUAD(c)
        }
#undef SGP_QUAD
#undef SGP_SUPER
        // sum the 4 column classes; class q keeps row q.  Two VALU-only exchange rounds:
        // v_permlane16_swap trades the odd 16-lane rows of one register with the even rows of
        // another (classes q <-> q ^ 1), v_permlane32_swap trades the wave halves (q <-> q ^ 2);
        // swap + add leaves every lane with the sum it has to keep.
        f32x4 out;
        if constexpr (ABL == 7) {
            out = acc0 + acc1 + acc2 + acc3;
        } else {
#define SGP_FOLD(ACC, DST)                                                                       \
            {                                                                                   \
                /* rows 0/1 of the group: even classes keep .x, odd keep .y; rows 2/3: .z/.w */  \
                auto p01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(ACC.x), __float_as_uint(ACC.y), false, false); \
                auto p23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(ACC.z), __float_as_uint(ACC.w), false, false); \
                const float r01 = __uint_as_float(p01[0]) + __uint_as_float(p01[1]);            \
                const float r23 = __uint_as_float(p23[0]) + __uint_as_float(p23[1]);            \
                auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(r01), __float_as_uint(r23), false, false); \
                DST = __uint_as_float(h[0]) + __uint_as_float(h[1]);                            \
            }
            SGP_FOLD(acc0, out.x) SGP_FOLD(acc1, out.y) SGP_FOLD(acc2, out.z) SGP_FOLD(acc3, out.w)
#undef SGP_FOLD
        }
        if (my_row >= 0)
            st4(a.Y + (long long)t * a.ybs + (long long)my_row * a.yrs + f_base + li * 4, out);
        if constexpr (ABL != 9) __syncthreads();
    }
}

template <bool HALO>
int launch_mfma(const MfmaArgs& a, hipStream_t s) {
    const size_t lds_bytes = 160 * 1024;
#ifdef SGP_ABLATION
#define SGP_ABL(V)                                                                                 \
    if (tiled_variant() == V) {                                                                    \
        auto k4 = spmm_mfma<HALO, V>;                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL(k4, dim3(a.n_tiles * a.n_tchunks, a.feat / 64), dim3(1024), 160 * 1024, s, a); \
        return sgp::check_launch("spmm_mfma");                                                     \
    }
    SGP_ABL(4) SGP_ABL(5) SGP_ABL(6) SGP_ABL(7) SGP_ABL(8) SGP_ABL(9)
#undef SGP_ABL
#endif
    auto kern = spmm_mfma<HALO>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return sgp::fail((int)e, "spmm_mfma: LDS opt-in: %s", hipGetErrorString(e));
    dim3 grid(a.n_tiles * a.n_tchunks, a.feat / 64);
    hipLaunchKernelGGL(kern, grid, dim3(1024), lds_bytes, s, a);
    return sgp::check_launch("spmm_mfma");
}


// ---- C entry (was in the extern "C" block)
int32_t sgp_spmm_mfma_max_union(void) { return kMfmaPasses * 64; }
int32_t sgp_spmm_mfma_max_quads(void) { return kMfmaMaxQuads; }

int sgp_spmm_mfma_f32(const int32_t* trow, const int32_t* uptr, const int32_t* ucol,
                      const int32_t* gptr, const int32_t* gidx, const float* gw,
                      const int32_t* rowmap,
                      int32_t n_tiles, int32_t max_union, int32_t max_tile_quads,
                      const float* X, int64_t xrs, int64_t xbs,
                      const float* Xh, int64_t xhrs, int64_t xhbs, int32_t n_own,
                      float* Y, int64_t yrs, int64_t ybs,
                      int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                      sgp_stream_t stream) {
    SGP_REQUIRE(trow && uptr && ucol && gptr && gidx && gw && rowmap && X && Y,
                "sgp_spmm_mfma_f32: null pointer");
    SGP_REQUIRE(n_tiles >= 0 && n_rows >= 0 && batch >= 0 && max_union >= 0 && max_tile_quads >= 0,
                "sgp_spmm_mfma_f32: bad size");
    {
        const long long own = Xh ? n_own : n_cols, far = Xh ? n_cols - n_own : 0;
        SGP_REQUIRE(n_cols >= 0 && own >= 0 && far >= 0 && own * xrs < (1ll << 31) && far * xhrs < (1ll << 31),
                    "sgp_spmm_mfma_f32: row offsets exceed 32 bits (use sgp_spmm_csr_f32)");
    }
    if (n_rows == 0 || batch == 0 || feat == 0) return 0;
    if (feat % 64 != 0)
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_mfma_f32: feat=%d is not a multiple of 64", feat);
    if (max_union > kMfmaPasses * 64 || max_tile_quads > kMfmaMaxQuads)
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_mfma_f32: tile working set (%d rows, %d quads) exceeds LDS (%d, %d)",
                         max_union, max_tile_quads, kMfmaPasses * 64, kMfmaMaxQuads);
    SGP_REQUIRE(xrs % 4 == 0 && xbs % 4 == 0 && yrs % 4 == 0 && ybs % 4 == 0 && sgp::aligned16(X) &&
                sgp::aligned16(Y) && (!Xh || (xhrs % 4 == 0 && xhbs % 4 == 0 && sgp::aligned16(Xh))) &&
                sgp::aligned16(gidx) && sgp::aligned16(gw),
                "sgp_spmm_mfma_f32: strides/pointers must be 16-byte aligned");
    MfmaArgs a;
    a.trow = trow; a.uptr = uptr; a.ucol = ucol; a.gptr = gptr; a.gidx = gidx; a.gw = gw;
    a.rowmap = rowmap;
    a.n_tiles = n_tiles;
    a.src = Src{X, xrs, xbs, Xh ? Xh : X, xhrs, xhbs, Xh ? n_own : 0x7fffffff};
    a.Y = Y; a.yrs = yrs; a.ybs = ybs;
    a.n_rows = n_rows; a.batch = batch; a.feat = feat;
    // Time chunk per workgroup: long enough to amortise the per-tile setup, short enough that
    // neighbouring tiles (which share source rows through L2 / Infinity Cache) cannot drift
    // far apart in t -- with 431-step chunks rocprofv3 showed 2.4x the algorithmic HBM reads.
    const int nft = feat / 64;
    long long want = (long long)batch * n_tiles * nft / 4096;
    int tc = (int)(want < 16 ? 16 : (want > chunk_cap() ? chunk_cap() : want));
    if (tc > batch) tc = batch;
    a.t_chunk = tc;
    a.n_tchunks = (batch + tc - 1) / tc;
    hipStream_t s = (hipStream_t)stream;
    return Xh ? launch_mfma<true>(a, s) : launch_mfma<false>(a, s);
}


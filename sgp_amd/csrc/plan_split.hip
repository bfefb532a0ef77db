// Host-side planner of sgp_spmm_split_f32 (kernel: spmm_split.hip; array formats: sgp_amd/splitplan.py, whose numpy
// planner this file restates -- tests/test_splitplan.py holds the two to the same bytes).  No device code: the
// functions below take HOST pointers and run on the caller's cores.
//
// Why native: the plan of the target graph (N = 100 000, 100-NN) cost 22 s of single-threaded numpy in front of an
// 80 ms encoder pass (round-5 review); the encoder is called once per experiment (reference: lib/utils.py:27-32), so
// its wall time is what a user sees.  Here: one O(nnz log) dealing pass + a per-tile fill on all cores.
#include "common.h"
#include <math.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

namespace {

// k-slots of the four 8-row sets a chunk's two transpose reads serve together (splitplan._SET_SLOTS)
constexpr int SET_SLOTS[4][8] = {{0, 1, 2, 3, 8, 9, 10, 11}, {4, 5, 6, 7, 12, 13, 14, 15},
                                 {16, 17, 18, 19, 24, 25, 26, 27}, {20, 21, 22, 23, 28, 29, 30, 31}};

inline uint16_t half_bits(_Float16 h) { uint16_t b; memcpy(&b, &h, 2); return b; }

// v (already scaled) -> hi + lo fp16 pieces: hi = v truncated to 10 explicit mantissa bits (rounded in the fp16
// subnormal range), lo = the rounded remainder (splitplan.split_fp16)
inline void split_fp16(float v, uint16_t& hi, uint16_t& lo) {
    uint32_t bits;
    memcpy(&bits, &v, 4);
    bits &= 0xFFFFE000u;
    float hi32;
    memcpy(&hi32, &bits, 4);
    if (fabsf(v) < 6.103515625e-05f) hi32 = (float)(_Float16)v;
    const _Float16 h = (_Float16)hi32;
    const _Float16 l = (_Float16)(v - (float)h);
    hi = half_bits(h);
    lo = half_bits(l);
}

// floor(log2(16384 / rmax)) clipped to [-126, 126] (exact: exponent extraction, not a rounded logarithm)
inline int row_exponent(float rmax) {
    if (!(rmax > 0.0f)) return 0;
    const float q = 16384.0f / rmax;
    int e;
    if (isinf(q)) e = 126;
    else { int ex; (void)frexpf(q, &ex); e = ex - 1; }
    return std::min(126, std::max(-126, e));
}

struct FillArgs {
    const int64_t* rowptr; const int64_t* col; const float* val;
    int64_t n_rows, n_cols;
    const int64_t* wave_of_row; const int64_t* slot_of_row; const int64_t* tile_of_wave; const int64_t* rows_of_wave;
    int64_t n_waves, n_tiles;
    int waves, chunks, max_union;
    int32_t* hdr; int32_t* rowid; int32_t* ucol; uint16_t* afr; int32_t* adr; float* rinv;
};

struct Scratch {
    std::vector<int32_t> stage_of_col, kpos_of_col;
    std::vector<int64_t> cols;
    std::vector<float> dense;
    explicit Scratch(const FillArgs& a) : stage_of_col(a.n_cols, -1), kpos_of_col(a.n_cols, -1), dense((size_t)a.chunks * 512) {}
};

// one tile: staged rows, per-wave k-slots (bank-aware), transpose-read addresses, A fragments.  Returns the number of
// (wave, column) keys, -1 on a plan that breaks the kernel's limits.
int64_t fill_tile(const FillArgs& a, int64_t t, const int64_t* first_wave, const int64_t* row_list, const int64_t* row_first,
                  Scratch& s, int64_t& union_out) {
    const int W = a.waves, C = a.chunks;
    const int64_t w0 = first_wave[t], w1 = first_wave[t + 1];
    // staged rows: the tile's sorted distinct columns
    s.cols.clear();
    for (int64_t w = w0; w < w1; ++w)
        for (int64_t i = row_first[w]; i < row_first[w + 1]; ++i) {
            const int64_t r = row_list[i];
            s.cols.insert(s.cols.end(), a.col + a.rowptr[r], a.col + a.rowptr[r + 1]);
        }
    std::sort(s.cols.begin(), s.cols.end());
    s.cols.erase(std::unique(s.cols.begin(), s.cols.end()), s.cols.end());
    if (s.cols.empty()) s.cols.push_back(0);                 // a tile of empty rows still stages one (finite) row: its padding reads
    const int64_t U = (int64_t)s.cols.size();
    if (U > a.max_union) return -1;
    union_out = U;
    std::vector<int64_t> tile_cols(s.cols);
    for (int64_t i = 0; i < U; ++i) {
        s.stage_of_col[tile_cols[i]] = (int32_t)i;
        a.ucol[t * a.max_union + i] = (int32_t)tile_cols[i];
    }
    a.hdr[t * 64 + 2 * W] = (int32_t)U;
    int64_t keys = 0;
    for (int64_t w = w0; w < w1; ++w) {
        const int win = (int)(w - w0);
        const int nrows = (int)a.rows_of_wave[w];
        a.hdr[t * 64 + W + win] = nrows;
        // the wave's sorted distinct columns; position p -> chunk p / 32
        s.cols.clear();
        for (int64_t i = row_first[w]; i < row_first[w + 1]; ++i) {
            const int64_t r = row_list[i];
            a.rowid[(t * W + win) * 16 + a.slot_of_row[r]] = (int32_t)r;
            s.cols.insert(s.cols.end(), a.col + a.rowptr[r], a.col + a.rowptr[r + 1]);
        }
        std::sort(s.cols.begin(), s.cols.end());
        s.cols.erase(std::unique(s.cols.begin(), s.cols.end()), s.cols.end());
        const int nk = (int)s.cols.size();
        if (nk > 32 * C) return -1;
        keys += nk;
        int32_t* adr = a.adr + ((t * W + win) * C) * 64;
        for (int c = 0; c < C; ++c) {
            // k-slots of the chunk: the j-th row of a bank residue (stage & 7) goes to set j, in residue order inside
            // the set; rows beyond four of a residue take the free slots in order
            int srow[32];
            for (int k = 0; k < 32; ++k) srow[k] = -1;
            const int k0 = 32 * c, k1 = std::min(nk, 32 * c + 32);
            int cnt_res[8] = {0, 0, 0, 0, 0, 0, 0, 0}, in_set[4] = {0, 0, 0, 0};
            int rank[32], slot[32];
            for (int k = k0; k < k1; ++k) rank[k - k0] = cnt_res[s.stage_of_col[s.cols[k]] & 7]++;
            // primaries: for rank r, keys in residue order (at most one per residue)
            for (int r = 0; r < 4; ++r)
                for (int res = 0; res < 8; ++res)
                    for (int k = k0; k < k1; ++k)
                        if (rank[k - k0] == r && (s.stage_of_col[s.cols[k]] & 7) == res) slot[k - k0] = SET_SLOTS[r][in_set[r]++];
            bool occ[32] = {};
            for (int k = k0; k < k1; ++k) if (rank[k - k0] < 4) occ[slot[k - k0]] = true;
            int free_at = 0;
            for (int res = 0; res < 8; ++res)               // extras in (residue, stage) order
                for (int k = k0; k < k1; ++k)
                    if (rank[k - k0] >= 4 && (s.stage_of_col[s.cols[k]] & 7) == res) {
                        while (occ[free_at]) ++free_at;
                        slot[k - k0] = free_at;
                        occ[free_at] = true;
                    }
            for (int k = k0; k < k1; ++k) {
                srow[slot[k - k0]] = s.stage_of_col[s.cols[k]];
                s.kpos_of_col[s.cols[k]] = 32 * c + slot[k - k0];
            }
            // padding slots repeat a row of their own set (same address = broadcast); an empty set reads staged row 0
            for (int q = 0; q < 4; ++q) {
                int rep = -1;
                for (int i = 0; i < 8; ++i) { const int v = srow[SET_SLOTS[q][i]]; if (v >= 0 && (rep < 0 || v < rep)) rep = v; }
                if (rep < 0) rep = 0;
                for (int i = 0; i < 8; ++i) if (srow[SET_SLOTS[q][i]] < 0) srow[SET_SLOTS[q][i]] = rep;
            }
            for (int g = 0; g < 4; ++g)
                for (int li = 0; li < 16; ++li) {
                    unsigned av[2];
                    for (int j = 0; j < 2; ++j) {
                        const int sl = srow[8 * g + 4 * j + (li >> 2)];
                        av[j] = (unsigned)((sl >> 3) * 512 + (sl & 7) * 32 + 8 * (li & 3));
                    }
                    if (av[0] >= 65536u || av[1] >= 65536u) return -1;
                    adr[c * 64 + 16 * g + li] = (int32_t)(av[0] | (av[1] << 16));
                }
        }
        // A fragments: duplicates added in fp32 in edge order, every row scaled by its own power of two, split
        std::fill(s.dense.begin(), s.dense.end(), 0.0f);
        for (int64_t i = row_first[w]; i < row_first[w + 1]; ++i) {
            const int64_t r = row_list[i];
            const int sl = (int)a.slot_of_row[r];
            for (int64_t e = a.rowptr[r]; e < a.rowptr[r + 1]; ++e) {
                const int pos = s.kpos_of_col[a.col[e]];
                const int k = pos & 31;
                s.dense[(size_t)(pos >> 5) * 512 + (sl + 16 * (k >> 3)) * 8 + (k & 7)] += a.val[e];
            }
        }
        uint16_t* afr = a.afr + ((size_t)(t * W + win) * C) * 2 * 512;
        for (int sl = 0; sl < 16; ++sl) {
            float rmax = 0.0f;
            for (int c = 0; c < C; ++c)
                for (int g = 0; g < 4; ++g)
                    for (int e = 0; e < 8; ++e) rmax = std::max(rmax, fabsf(s.dense[(size_t)c * 512 + (sl + 16 * g) * 8 + e]));
            const int er = row_exponent(rmax);
            const float scale = ldexpf(1.0f, er);
            for (int c = 0; c < C; ++c)
                for (int g = 0; g < 4; ++g)
                    for (int e = 0; e < 8; ++e) {
                        const size_t at = (size_t)(sl + 16 * g) * 8 + e;
                        uint16_t hi, lo;
                        split_fp16(s.dense[(size_t)c * 512 + at] * scale, hi, lo);
                        afr[((size_t)c * 2 + 0) * 512 + at] = hi;
                        afr[((size_t)c * 2 + 1) * 512 + at] = lo;
                    }
            a.rinv[(t * W + win) * 16 + sl] = sl < nrows ? ldexpf(1.0f, -er) : 0.0f;
        }
    }
    return keys;
}

}  // namespace

extern "C" int64_t sgp_split_plan_deal(const int64_t* rowptr, const int64_t* col, int64_t n_rows, int64_t n_cols,
                                       const int64_t* order, int64_t n_order,
                                       int32_t waves, int32_t chunks, int32_t max_union, int32_t rows_per_wave,
                                       int64_t* wave_of_row, int64_t* slot_of_row, int64_t* tile_of_wave, int64_t* rows_of_wave) {
    if (!rowptr || (!col && n_rows > 0 && rowptr[n_rows] > 0) || !wave_of_row || !slot_of_row || !tile_of_wave || !rows_of_wave ||
        n_rows < 0 || n_cols < 0 || waves < 1 || chunks < 1 || max_union < 1 || rows_per_wave < 1 || (order && n_order < 0))
        return sgp::fail(SGP_EINVAL, "sgp_split_plan_deal: bad argument");
    const int64_t cap = 32ll * chunks;
    std::vector<int64_t> wmark(n_cols, -1), tmark(n_cols, -1), c;
    for (int64_t r = 0; r < n_rows; ++r) wave_of_row[r] = slot_of_row[r] = -1;
    int64_t wave = -1, tile = -1, w_rows = 0, w_cols = 0, t_cols = 0, t_waves = 0;
    auto open_wave = [&](bool new_tile) {
        ++wave;
        if (new_tile) { ++tile; t_cols = t_waves = 0; }
        ++t_waves;
        w_rows = w_cols = 0;
        tile_of_wave[wave] = tile;
        rows_of_wave[wave] = 0;
    };
    const int64_t n_seq = order ? n_order : n_rows;
    for (int64_t i = 0; i < n_seq; ++i) {
        const int64_t r = order ? order[i] : i;
        if (r < 0 || r >= n_rows) return sgp::fail(SGP_EINVAL, "sgp_split_plan_deal: order entry out of range");
        if (wave_of_row[r] >= 0) return sgp::fail(SGP_EINVAL, "sgp_split_plan_deal: row %lld appears twice in the order", (long long)r);
        c.assign(col + rowptr[r], col + rowptr[r + 1]);
        std::sort(c.begin(), c.end());
        c.erase(std::unique(c.begin(), c.end()), c.end());
        const int64_t nc = (int64_t)c.size();
        if (nc > cap || nc > max_union) return -2;             // a row beyond a wave's column budget: no single-pass plan
        if (nc && (c.front() < 0 || c.back() >= n_cols)) return sgp::fail(SGP_EINVAL, "sgp_split_plan_deal: column out of range");
        if (wave < 0) open_wave(true);
        int64_t new_w = 0, new_t = 0;
        for (int64_t v : c) { new_w += wmark[v] != wave; new_t += tmark[v] != tile; }
        const bool need_wave = w_rows == rows_per_wave || w_cols + new_w > cap;
        if (t_cols + new_t > max_union || (need_wave && t_waves == waves)) { open_wave(true); new_w = new_t = nc; }
        else if (need_wave) { open_wave(false); new_w = nc; }
        for (int64_t v : c) { wmark[v] = wave; tmark[v] = tile; }
        w_cols += new_w;
        t_cols += new_t;
        wave_of_row[r] = wave;
        slot_of_row[r] = w_rows;
        rows_of_wave[wave] = ++w_rows;
    }
    return wave + 1;
}

extern "C" int sgp_split_plan_fill(const int64_t* rowptr, const int64_t* col, const float* val, int64_t n_rows, int64_t n_cols,
                                   const int64_t* wave_of_row, const int64_t* slot_of_row, const int64_t* tile_of_wave,
                                   const int64_t* rows_of_wave, int64_t n_waves, int64_t n_tiles,
                                   int32_t waves, int32_t chunks, int32_t max_union,
                                   int32_t* hdr, int32_t* rowid, int32_t* ucol, void* afr, int32_t* adr, float* rinv,
                                   double* stats, int32_t threads) {
    SGP_REQUIRE(rowptr && col && val && wave_of_row && slot_of_row && tile_of_wave && rows_of_wave && hdr && rowid && ucol &&
                afr && adr && rinv && stats, "sgp_split_plan_fill: null pointer");
    SGP_REQUIRE(n_rows > 0 && n_cols > 0 && n_waves > 0 && n_tiles > 0 && waves >= 1 && 2 * waves < 64 && chunks >= 1 &&
                max_union >= 1 && max_union <= 65536 / 64 * 8, "sgp_split_plan_fill: bad size");
    FillArgs a{rowptr, col, val, n_rows, n_cols, wave_of_row, slot_of_row, tile_of_wave, rows_of_wave, n_waves, n_tiles,
               waves, chunks, max_union, hdr, rowid, ucol, (uint16_t*)afr, adr, rinv};
    memset(hdr, 0, sizeof(int32_t) * 64 * n_tiles);
    std::fill(rowid, rowid + n_tiles * waves * 16, -1);
    std::fill(ucol, ucol + n_tiles * (int64_t)max_union, -1);
    memset(afr, 0, sizeof(uint16_t) * (size_t)n_tiles * waves * chunks * 2 * 512);
    memset(adr, 0, sizeof(int32_t) * (size_t)n_tiles * waves * chunks * 64);
    memset(rinv, 0, sizeof(float) * (size_t)n_tiles * waves * 16);
    // waves of every tile (contiguous), rows of every wave in slot order
    std::vector<int64_t> first_wave(n_tiles + 1, 0), row_first(n_waves + 1, 0);
    for (int64_t w = 0; w < n_waves; ++w) {
        const int64_t t = tile_of_wave[w];
        SGP_REQUIRE(t >= 0 && t < n_tiles && (w == 0 || t == tile_of_wave[w - 1] || t == tile_of_wave[w - 1] + 1),
                    "sgp_split_plan_fill: tile_of_wave is not a non-decreasing run");
        SGP_REQUIRE(rows_of_wave[w] >= 0 && rows_of_wave[w] <= 16, "sgp_split_plan_fill: a wave holds at most 16 rows");
        ++first_wave[t + 1];
        row_first[w + 1] = row_first[w] + rows_of_wave[w];
    }
    for (int64_t t = 0; t < n_tiles; ++t) {
        SGP_REQUIRE(first_wave[t + 1] >= 1 && first_wave[t + 1] <= waves, "sgp_split_plan_fill: a tile holds 1 .. waves waves");
        first_wave[t + 1] += first_wave[t];
    }
    const int64_t n_dealt = row_first[n_waves];
    std::vector<int64_t> row_list(std::max<int64_t>(n_dealt, 1), -1);
    int64_t seen = 0;
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t w = wave_of_row[r];
        if (w < 0) continue;
        SGP_REQUIRE(w < n_waves && slot_of_row[r] >= 0 && slot_of_row[r] < rows_of_wave[w] && row_list[row_first[w] + slot_of_row[r]] < 0,
                    "sgp_split_plan_fill: wave / slot of row %lld inconsistent", (long long)r);
        row_list[row_first[w] + slot_of_row[r]] = r;
        ++seen;
    }
    SGP_REQUIRE(seen == n_dealt, "sgp_split_plan_fill: rows_of_wave does not match wave_of_row");

    int nthr = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    nthr = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(nthr, 64), n_tiles));
    std::atomic<int64_t> next(0), keys(0), usum(0), umax(0);
    std::atomic<int> bad(0);
    auto work = [&]() {
        Scratch s(a);
        int64_t k_local = 0, u_local = 0, u_max = 0;
        for (;;) {
            const int64_t t = next.fetch_add(1);
            if (t >= n_tiles || bad.load()) break;
            int64_t U = 0;
            const int64_t k = fill_tile(a, t, first_wave.data(), row_list.data(), row_first.data(), s, U);
            if (k < 0) { bad.store(1); break; }
            k_local += k; u_local += U; u_max = std::max(u_max, U);
        }
        keys += k_local; usum += u_local;
        int64_t cur = umax.load();
        while (u_max > cur && !umax.compare_exchange_weak(cur, u_max)) {}
    };
    if (nthr == 1) work();
    else {
        std::vector<std::thread> pool;
        for (int i = 0; i < nthr; ++i) pool.emplace_back(work);
        for (auto& th : pool) th.join();
    }
    if (bad.load()) return sgp::fail(SGP_EUNSUP, "sgp_split_plan_fill: the dealt rows break the kernel's limits");
    double norm_inf = 0.0;
    for (int64_t r = 0; r < n_rows; ++r) {
        double sum = 0.0;
        for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e) sum += fabs((double)val[e]);
        norm_inf = std::max(norm_inf, sum);
    }
    const double dealt = (double)std::max<int64_t>(1, n_dealt);
    stats[0] = (double)n_tiles; stats[1] = (double)n_waves; stats[2] = (double)n_dealt / (double)n_waves;
    stats[3] = dealt / (double)n_tiles; stats[4] = (double)usum.load() / dealt;
    stats[5] = (double)keys.load() / ((double)n_waves * chunks * 32); stats[6] = (double)umax.load(); stats[7] = norm_inf;
    return 0;
}

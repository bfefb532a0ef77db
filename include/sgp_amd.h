/*
 * sgp_amd.h -- C ABI of libsgp_amd.so: the MI355X (gfx950) device side of SGP's
 * training-free spatiotemporal encoder.
 *
 * The reference (Graph-Machine-Learning-Group/sgp) is pure Python; the native work on its
 * hot path is done by third-party ops.  Each entry point below replaces one of
 * those call sites (paths relative to the reference repo root):
 *
 *   sgp_spmm_csr_f32 / sgp_spmm_tiled_f32
 *       lib/sgp_preprocessing.py:202   `x = adj @ x`
 *       (torch_sparse SparseTensor.__matmul__ -> spmm_sum(row,rowptr,col,value,colptr,csr2csc,mat))
 *   sgp_reservoir_f32
 *       lib/nn/reservoir/reservoir.py:77-81 (ReservoirLayer.forward: 2x F.linear, act, leak)
 *       driven by the Python time loop at lib/nn/reservoir/reservoir.py:170-183
 *   sgp_node_mean_bcast_f32
 *       lib/nn/encoders/sgp_spatial_encoder.py:32-34 (`ones_like(x) * x.mean(-2, keepdim=True)`)
 *   sgp_copy_rows_f32
 *       lib/sgp_preprocessing.py:200,217 + sgp_spatial_encoder.py:35 (`torch.cat(out, -1)`) --
 *       only needed when a caller hands over a tensor that is not already in the
 *       output slot; the fused path writes slots in place and never concatenates.
 *   sgp_gather_rows_f32
 *       lib/datasets/iid_dataset.py:57-99 (IID (t, n) row gather of the embedding; "next" row f1)
 *   sgp_grouped_linear_f32
 *       lib/nn/models/sgp_model.py:41-52 (decoder input encoder: grouped Conv1d + activation; f4)
 *
 * Conventions
 *   - All pointers are DEVICE pointers owned by the caller (e.g. the PyTorch
 *     allocator) unless stated otherwise; nothing is allocated inside.
 *   - Strides are in ELEMENTS (floats), not bytes.
 *   - Kernels are enqueued asynchronously on `stream` (a hipStream_t passed as
 *     void*; NULL = the default stream).
 *   - Every function returns 0 on success, a negative SGP_E* code on a bad
 *     argument, or a positive hipError_t if the runtime refused the launch.
 *     Nothing throws.  sgp_last_error() returns a thread-local description.
 */
#ifndef SGP_AMD_H
#define SGP_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 6): the launch predicate became the explicit trailing (pred, run_if) pair of every sgp_spmm_*_f32 entry
 * (sgp_launch_predicate removed); host-side planner sgp_split_plan_deal / sgp_split_plan_fill, the time-piece
 * reservoir entry sgp_reservoir_pieces_f32 and the wide split hop sgp_spmm_split_wide_f32 added.
 * 2 (round 5): sgp_spmm_split_f32 takes per-column scale tables and a per-row plan array; sgp_col_stats_f32,
 * sgp_split_prepare_f32, sgp_launch_predicate added; round 4 had already removed sgp_spmm_mfma / pipe / blk_*, widened
 * sgp_spmm_colblock_f32 by the halo arguments and grown sgp_reservoir_workspace_bytes (bf16-piece fragments). */
#define SGP_ABI_VERSION 3

#define SGP_EINVAL   (-1)  /* bad size / null pointer / misaligned stride */
#define SGP_EUNSUP   (-2)  /* shape outside what the kernels are built for */

/* activations of lib/nn/reservoir/reservoir.py:37-41 */
#define SGP_ACT_TANH      0
#define SGP_ACT_RELU      1
#define SGP_ACT_SELF_NORM 2
#define SGP_ACT_IDENTITY  3
/* tanh evaluated with RELATIVE accuracy (an odd polynomial below |x| = 0.25; SGP_ACT_TANH is accurate to 3e-7 absolute,
 * 6 instead of 14 instructions per value): for layers whose bias and input scaling are so small that their states are
 * far below 1 -- sgp_amd's Python layer selects it when max |bias| < 0.25 */
#define SGP_ACT_TANH_REL  4

typedef void* sgp_stream_t;

int sgp_abi_version(void);
const char* sgp_last_error(void);
/* Name of the gfx target the device code was compiled for ("gfx950"). */
const char* sgp_build_arch(void);

/* The library's view of the ONE debug / tuning hook, the environment variable SGP_TUNE = "key=value,key=value"
 * (keys: sgp_amd/tune.py): the integer value of `key`, `dflt` when the variable or the key is absent.  Most keys
 * are read once per process by the kernel launchers; this entry parses the variable anew on every call. */
int64_t sgp_tune_value(const char* key, int64_t dflt);

/* ------------------------------------------------------------------ SpMM ---
 * Y[b, i, 0:feat] = sum_{e in [rowptr[i], rowptr[i+1])} val[e] * X[b, col[e], 0:feat]
 * for b in [0, batch), i in [0, n_rows).  X and Y may alias the same
 * allocation as long as the [feat]-wide column ranges do not overlap
 * (hop k reads slot k-1 and writes slot k of the [T, N, D_out] output).
 *
 * Columns >= n_own (when X_halo != NULL) are read from the halo buffer:
 *   X_halo[b, col - n_own, :]   (rows received from peer GPUs between hops)
 * Pass X_halo = NULL, n_own = n_cols for the single-GPU case.
 */
int sgp_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                     const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                     const float* X_halo, int64_t xh_row_stride, int64_t xh_batch_stride,
                     int32_t n_own,
                     float* Y, int64_t y_row_stride, int64_t y_batch_stride,
                     int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                     const int32_t* pred, int32_t run_if, sgp_stream_t stream);

/* LDS-staged variant for graphs with locality.  Rows are grouped into tiles of at most
 * `tile_rows` consecutive rows (tile k = rows tile_row_ptr[k] .. tile_row_ptr[k+1]); for
 * tile k the host supplies the sorted list of distinct source rows it references
 * (ucol[uptr[k] .. uptr[k+1])) and every edge carries the index of its column inside
 * that list.  Edge lists are padded per row to a multiple of 16 with (lcol = 0, val = 0):
 *   erow[i] .. erow[i+1]  = padded edge range of row i (multiple of 16 long)
 *   ecol[e] (uint16)      = index into the tile's ucol list
 *   eval[e]               = weight
 * max_union = largest per-tile list length, max_row_edges = largest padded per-row
 * edge count (host-side facts about the plan; they select the kernel variant).
 * Tiles of more than 128 rows (up to sgp_spmm_tiled_max_tile_rows() = 384; rows with at most 32
 * edges) take the tall form: the edge records live in LDS behind the stage, which must then hold
 * ceil(max_union / 64) * 16 KiB + 6 bytes per padded edge slot of 256 / 384 rows within 160 KiB
 * (small sparse graphs as ONE tile: every source row is staged once per step).
 * Returns SGP_EUNSUP if the plan exceeds the limits reported below.
 */
int sgp_spmm_tiled_f32(const int32_t* tile_row_ptr, const int32_t* uptr, const int32_t* ucol,
                       const int32_t* erow, const uint16_t* ecol, const float* eval,
                       int32_t tile_rows, int32_t n_tiles,
                       int32_t max_union, int32_t max_row_edges,
                       const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                       const float* X_halo, int64_t xh_row_stride, int64_t xh_batch_stride,
                       int32_t n_own,
                       float* Y, int64_t y_row_stride, int64_t y_batch_stride,
                       int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                       const int32_t* pred, int32_t run_if, sgp_stream_t stream);
/* Row-group kernel on the fp32 matrix cores, exact fp32 (lib/sgp_preprocessing.py:200-203, `x = adj @ x` per hop).
 * Tiles of at most 64 rows and their distinct-column lists as above; every tile is cut into 16 groups of 4
 * rows (slots 4g .. 4g+3 of the tile, see rowmap).  A group's sorted column union is dealt round-robin to 4
 * classes q; super-step s handles the s-th column of every class with one v_mfma_f32_4x4x1_16b_f32 per
 * feature.  The stream is stored 4 super-steps ("quad") at a time.  The tile's distinct-column list is cut
 * at usplit[k] (a multiple of 4) into segment A (list positions < usplit[k]) and segment B; a group's quads
 * are stored A-part first and never mix the segments: gptr[2 (16 k + g)] .. gptr[2 (16 k + g) + 1] = quads
 * on segment A, .. gptr[2 (16 k + g) + 2] = quads on segment B.  Per time step the kernel refills one
 * segment of the LDS stage by LDS-DMA (global_load_lds_dwordx4) while the matrix cores consume the other.
 *   gw  [quad][q][super-step s][row i]  float  weight of row i for class q's column in super-step s of the
 *                                        quad (0 = row lacks the column / padding); the MFMA of super-step s
 *                                        broadcasts block s of its class (cbsz = 2 / abid = s)
 *   gidx[quad][q][4]          int32   256 * index of that column in the tile's ucol list, i.e. the byte
 *                                     offset of its staged row in LDS (0 = padding)
 *   gsup[2 (16 k + g) + s]    int32   ceil(columns of that range / 4): the padding of a range's last quad is
 *                                     skipped in units of one super-step
 *   rowmap[64 * k + 4 g + i]  int32   output row of slot i of group g of tile k, -1 = empty (the host may
 *                                     permute rows inside a tile so that the 4 rows of a group share columns)
 * uptr / ucol list segment A first (padded to a multiple of 4 entries), then segment B.  A group's weights
 * and the per-lane LDS addresses of its staged rows are loaded once per workgroup into VGPRs, so a super-step
 * is one ds_read_b128 + 4 MFMAs with no VALU address arithmetic and no stream reads from LDS (ranges longer
 * than the resident super-steps continue from an LDS copy of the stream).  Limits: sgp_spmm_res_max_union()
 * staged rows and sgp_spmm_res_max_quads() quads per tile.  sgp_spmm_res_tune(cfg): 0 = 16 waves x 1 group
 * per workgroup, 1 = 8 waves x 2 groups (process-wide).
 * (Round 4 retired this kernel's predecessors sgp_spmm_mfma_f32 / sgp_spmm_pipe_f32 and the row-block form
 * sgp_spmm_blk_f32 -- superseded on every measured workload; their sources are kept under tools/experiments/.) */
int sgp_spmm_res_f32(const int32_t* uptr, const int32_t* ucol, const int32_t* usplit,
                     const int32_t* gptr, const int32_t* gsup, const int32_t* gidx, const float* gw,
                     const int32_t* rowmap,
                     int32_t n_tiles, int32_t max_union, int32_t max_tile_quads,
                     const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                     const float* X_halo, int64_t xh_row_stride, int64_t xh_batch_stride,
                     int32_t n_own,
                     float* Y, int64_t y_row_stride, int64_t y_batch_stride,
                     int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                     const int32_t* pred, int32_t run_if, sgp_stream_t stream);
int32_t sgp_spmm_res_max_union(void);
int32_t sgp_spmm_res_max_quads(void);
int sgp_spmm_res_tune(int32_t cfg);

/* Mixed dense / sparse form of the row-group product (lib/sgp_preprocessing.py:200-203, `x = adj @ x`
 * per hop; plan: sgp_amd/mixplan.py).  Tiles, staged rows, two-phase LDS-DMA staging and the
 * 4-row-group stream (uptr .. rowmap, gptr .. gw) are those of sgp_spmm_res_f32, but the 16 groups of
 * a tile form 4 blocks of 16 rows (tile slot = 16 block + 4 group + row), and the columns shared by
 * (nearly) all groups of a block are taken out of the group streams and multiplied by
 * v_mfma_f32_16x16x4_f32 instead (16 rows x 4 columns x 16 features per instruction):
 *   dptr[2 * 4 * n_tiles + 1]   first dense instruction of (tile, block, segment)
 *   didx[n_dense][4]            LDS byte offsets (staged row * 256) of the instruction's 4 columns
 *   dw[n_dense][64]             its A operand in lane order: lane 16 k + i = weight of (row i of the
 *                               block, column k), 0 where the row does not use the column
 * max_dense = longest (block, segment) list, at most sgp_spmm_mix_max_dense(halo != 0) (the lists
 * live in registers).  Wave w of a workgroup owns sparse group w and the dense part of block w / 4
 * for the feature quarter w % 4; dense sums reach the storing wave through a 16 KB LDS slab.
 * Arithmetic: exact fp32 FMAs; a row's sum is (sparse part, k order of its group stream) + (dense
 * part, k order of the block's list) -- another summation order than sgp_spmm_res_f32, same values
 * to rounding.  X / X_halo / Y as in sgp_spmm_tiled_f32. */
int sgp_spmm_mix_f32(const int32_t* uptr, const int32_t* ucol, const int32_t* usplit,
                     const int32_t* gptr, const int32_t* gsup, const int32_t* gidx, const float* gw,
                     const int32_t* rowmap,
                     const int32_t* dptr, const int32_t* didx, const float* dw,
                     int32_t n_tiles, int32_t max_union, int32_t max_dense,
                     const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                     const float* X_halo, int64_t xh_row_stride, int64_t xh_batch_stride,
                     int32_t n_own,
                     float* Y, int64_t y_row_stride, int64_t y_batch_stride,
                     int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                     const int32_t* pred, int32_t run_if, sgp_stream_t stream);
int32_t sgp_spmm_mix_max_union(void);
int32_t sgp_spmm_mix_max_dense(int32_t halo);

/* out[0] = max |X| over a strided [batch, n_rows, feat] view (device scalar; non-finite inputs give a
 * non-finite value).  Serves callers of sgp_spmm_split_f32 that have no analytic bound on their operand
 * (`sgp_spatial_embedding` on arbitrary [B, N, F] batches, lib/nn/models/sgp_model.py:169-181; relu
 * reservoirs).  Widths that are multiples of 4 need 16-byte aligned rows. */
int sgp_abs_max_f32(const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                    int32_t n_rows, int32_t batch, int32_t feat, float* out, sgp_stream_t stream);

/* Launch predicate of the hop entries (the trailing `pred, run_if` pair of every sgp_spmm_*_f32): with pred != NULL (a
 * DEVICE word) the launch runs only if *pred == run_if when its kernel starts on the stream, and is a no-op otherwise
 * (every workgroup exits at its first instruction).  This is how a hop chooses between the split-fp16 kernel and the
 * exact-fp32 kernels ON THE DEVICE: sgp_split_prepare_f32 writes the flag, the caller enqueues sgp_spmm_split_f32 with
 * run_if = 1 and its exact kernel with run_if = 0 behind it (no host round trip, legal under stream capture).
 * pred = NULL: unconditional.  (ABI 2 carried this as thread-local state set by sgp_launch_predicate(); ABI 3 made it
 * an argument so that nothing armed by one call can reach another.) */

/* Per-column statistics of a strided [batch, n_rows, feat] view over the steps 0, t_stride, 2 t_stride, ... and the rows
 * 0, r_stride, 2 r_stride, ...: stats[0 : feat] = max |x[:, c]| (bit pattern of the float; NaN / inf win),
 * stats[feat : 2 feat] = sum of squares.  accumulate = 0 clears stats first; 1 adds a second source (the halo rows of
 * a node partition) to it.  feat % 4 == 0, feat <= 1024, 16-byte aligned rows. */
int sgp_col_stats_f32(const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                      int32_t n_rows, int32_t batch, int32_t feat, int32_t t_stride, int32_t r_stride, int32_t accumulate,
                      float* stats, sgp_stream_t stream);

/* Scales, next bound and admission flag of the split-fp16 hop, on the device.  Per column c the bound B_c >= max
 * |x[:, c]| is bound_in[c] (device) if given, else bound_scalar if > 0, else the measured maximum (needs full = 1:
 * statistics over EVERY row).  An a-priori bound is never replaced by a measurement: the scales of a bounded operand
 * do not depend on how the caller cuts the time axis (bit-identical results per time chunk).  x_tab[c] = 2^e with
 * B_c 2^e in [2^13, 2^14] (B_c = 0: scale 1, the column is identically zero), x_tab[feat + c] = 2^-e,
 * bound_out[c] = B_c * norm_inf * (1 + 1e-6) (the bound of A x for ||A||_inf = norm_inf; may be NULL).
 * flag[0] = 1 iff every bound is finite, covers the sampled data, and -- when statistics are given -- satisfies
 *     B_c * sqrt(s_eff) <= 2^16 * sqrt(ssq_c / n_samples)           for every column with B_c > 0
 * (n_samples = rows behind the sums, s_eff = rows of the operand / rows sampled >= 1: a strided sample overstates a
 * mean square by at most that factor).  Why: the split kernel represents |x| >= 2^-16 B_c to 2^-23 relative and
 * everything smaller to 2^-38 B_c ABSOLUTE; the test keeps that absolute term below 2^-22 of the column's RMS,
 * i.e. at fp32's own level.  stats = NULL: no test, the caller vouches for the bound (flag = bounds finite). */
int sgp_split_prepare_f32(const float* stats, double n_samples, double s_eff, int32_t full,
                          const float* bound_in, float bound_scalar, float norm_inf, int32_t feat,
                          float* x_tab, float* bound_out, int32_t* flag, sgp_stream_t stream);

/* Split-fp16 hop (lib/sgp_preprocessing.py:200-203, `x = adj @ x` per hop; plan: sgp_amd/splitplan.py;
 * kernel: csrc/spmm_split.hip).  Every operand value is carried as two fp16 pieces of its scaled self
 * (v * scale = hi + lo, 22 significant bits) and every product as hi*hi + hi*lo + lo*hi accumulated in fp32
 * by v_mfma_f32_16x16x32_f16, at 16x the fp32 matrix rate, which pays for dense 16 x 32 blocks of A and 256-row
 * tiles (3.2 staged source rows per result row instead of 5.8).
 * ERROR MODEL.  x is scaled per feature column (x_tab, from sgp_split_prepare_f32), A per row (plan), so the
 * result is invariant to rescaling a column of x or a row of A, like fp32.  Within 2^16 of its column's bound a
 * value keeps 22 bits (relative 2^-23); below that the error is absolute, <= 2^-38 x the column's bound.  The
 * products are exact, sums are fp32.  The kernel is therefore fp32-equivalent exactly where
 * sgp_split_prepare_f32 sets its flag, and callers launch it under that predicate with an exact kernel behind it.
 * Arrays:
 *   hdr[n_tiles][64]                         [W : 2 W] rows of each of the W waves, [2 W] staged rows U
 *   rowid[n_tiles][W][16]                    result row of every slot of every wave, -1 = empty
 *   ucol[n_tiles][max_union]                 source row staged at position s (-1 beyond U)
 *   afr[n_tiles][W][chunks][2][64][8] fp16   A fragments in lane order (piece 0 | 1) of a[i, :] * 2^e_i
 *   adr[n_tiles][W][chunks][64]              per-lane byte addresses of the two transpose reads, a0 | a1 << 16
 *   rinv[n_tiles][W][16]                     2^-e_i of every slot's row (0 for empty slots)
 * with W = sgp_spmm_split_waves() waves of sgp_spmm_split_rows_per_wave() = 16 rows, the k-slots of a chunk in any
 * order (the planner picks one that keeps the rows a transpose read fetches together on different LDS banks),
 * chunks = sgp_spmm_split_chunks(), max_union = sgp_spmm_split_max_union().  feat % 16 == 0, feat <=
 * sgp_spmm_split_max_feat().  X / X_halo / n_own as in sgp_spmm_tiled_f32 (columns >= n_own address the halo rows
 * a node partition received).  x_tab: device [2][feat], scale then inverse, |x[:, c]| * x_tab[c] < 65504.
 * NON-FINITE OPERANDS.  The kernel multiplies dense 16 x 32 blocks: padded and zero entries form 0 * x, so a NaN /
 * inf in a source row (or a value that overflows fp16 after scaling: |x| above its column's bound by 4x) reaches EVERY
 * result row of every wave that stages that row, not only the row's graph neighbours as in a sparse fp32 product.
 * sgp_split_prepare_f32's admission test sees non-finite values only in the rows and steps its statistics read: all of
 * them when the bound is measured (full = 1), a sample (~8 steps, every r-th row, plus every row of the last step -- where
 * a value that entered a recurrence earlier still is) when the caller supplies an a-priori bound.  A caller that passes a bound therefore vouches for finiteness and for the bound on the unsampled part
 * (sgp_amd's encoders pass one only for states their own bounded-activation reservoir kernels wrote).
 * accumulate != 0: Y += A X (the later passes of an operator whose rows were cut into column segments).
 * t_chunk = time steps per workgroup (0 = chosen here). */
int sgp_spmm_split_f32(const int32_t* hdr, const int32_t* rowid, const int32_t* ucol, const void* afr,
                       const int32_t* adr, const float* rinv, int32_t n_tiles,
                       const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                       const float* X_halo, int64_t xh_row_stride, int64_t xh_batch_stride, int32_t n_own,
                       float* Y, int64_t y_row_stride, int64_t y_batch_stride,
                       int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                       const float* x_tab, int32_t accumulate, int32_t t_chunk, const int32_t* pred, int32_t run_if, sgp_stream_t stream);
int32_t sgp_spmm_split_chunks(void);
int32_t sgp_spmm_split_max_union(void);
int32_t sgp_spmm_split_waves(void);
int32_t sgp_spmm_split_rows_per_wave(void);
int32_t sgp_spmm_split_max_feat(void);

/* The same kernel in its WIDE form (csrc/spmm_split_wide.hip): 8 waves x 16 rows x 14 chunks -- 448 columns per wave, two
 * waves per SIMD.  For operators whose rows exceed the standard form's 224 columns (the reference's full large-scale
 * graphs, config/largescale/sgp_pv.yaml / sgp_cer.yaml with experiments/run_largescale_sgp.py:167-170: ~740 / ~495
 * entries per row): half as many accumulating passes and less than half the staged rows per result row.  Same arguments,
 * array formats and error model as sgp_spmm_split_f32, with W = sgp_spmm_split_wide_waves(), chunks =
 * sgp_spmm_split_wide_chunks(), max_union = sgp_spmm_split_wide_max_union(). */
int sgp_spmm_split_wide_f32(const int32_t* hdr, const int32_t* rowid, const int32_t* ucol, const void* afr,
                            const int32_t* adr, const float* rinv, int32_t n_tiles,
                            const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                            const float* X_halo, int64_t xh_row_stride, int64_t xh_batch_stride, int32_t n_own,
                            float* Y, int64_t y_row_stride, int64_t y_batch_stride,
                            int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                            const float* x_tab, int32_t accumulate, int32_t t_chunk, const int32_t* pred, int32_t run_if, sgp_stream_t stream);
int32_t sgp_spmm_split_wide_chunks(void);
int32_t sgp_spmm_split_wide_max_union(void);
int32_t sgp_spmm_split_wide_waves(void);
int32_t sgp_spmm_split_wide_rows_per_wave(void);
int32_t sgp_spmm_split_wide_max_feat(void);

/* Host-side planner of sgp_spmm_split_f32 (csrc/plan_split.hip; HOST pointers, no GPU needed; the encoder builds the
 * plan once per graph in front of `x = adj @ x`, lib/sgp_preprocessing.py:188-203).
 * sgp_split_plan_deal: rows -- in `order` (n_order entries) or 0 .. n_rows - 1 when order = NULL -- are dealt greedily to
 * waves of at most rows_per_wave rows touching at most 32 * chunks distinct columns, waves to tiles of at most `waves`
 * waves touching at most max_union distinct columns.  Writes wave_of_row / slot_of_row [n_rows] (-1 for rows outside
 * the order) and tile_of_wave / rows_of_wave (capacity n_rows); returns the number of waves, -2 when a single row
 * exceeds a wave's budget (no one-pass plan), -1 on a bad argument.
 * sgp_split_plan_fill: the kernel's arrays (formats: sgp_spmm_split_f32 above) for that deal, tiles in parallel on
 * `threads` host threads (0 = all); stats[8] = tiles, waves, rows per wave, rows per tile, staged rows per result row,
 * chunk fill, largest staged-row count, ||A||_inf. */
int64_t sgp_split_plan_deal(const int64_t* rowptr, const int64_t* col, int64_t n_rows, int64_t n_cols,
                            const int64_t* order, int64_t n_order,
                            int32_t waves, int32_t chunks, int32_t max_union, int32_t rows_per_wave,
                            int64_t* wave_of_row, int64_t* slot_of_row, int64_t* tile_of_wave, int64_t* rows_of_wave);
int sgp_split_plan_fill(const int64_t* rowptr, const int64_t* col, const float* val, int64_t n_rows, int64_t n_cols,
                        const int64_t* wave_of_row, const int64_t* slot_of_row, const int64_t* tile_of_wave,
                        const int64_t* rows_of_wave, int64_t n_waves, int64_t n_tiles,
                        int32_t waves, int32_t chunks, int32_t max_union,
                        int32_t* hdr, int32_t* rowid, int32_t* ucol, void* afr, int32_t* adr, float* rinv,
                        double* stats, int32_t threads);

/* Column-blocked hop for graphs without locality (lib/sgp_preprocessing.py:202, `x = adj @ x`; plan:
 * sgp_amd/colblock.py).  The columns are cut into n_blocks blocks of consecutive columns whose source
 * rows fit the L2 of an XCD; n_wg persistent workgroups (16 waves) each own a contiguous range of at
 * most sgp_spmm_colblock_rows_cap() rows for all time steps and sweep the blocks in the same order, so
 * every gather of a sweep is served by the L2; partial sums of a step stay in LDS.  Arrays:
 *   plan[n_entries][2]            {column | (row - wg_row0[wg]) << 23, weight bits}: the edges of a
 *                                 (workgroup, block) segment row by row, padded with weight-0 entries
 *                                 to a multiple of 64 * sgp_spmm_colblock_round_pad()
 *   segptr[n_wg * n_blocks + 1]   first ROUND (64 entries) of segment (wg, block)
 *   wg_row0[n_wg + 1]             first row of every workgroup
 * feat must be a multiple of 64 (one launch dimension per 64 features), n_cols < 2^22.  X_halo / n_own as in
 * sgp_spmm_tiled_f32: columns >= n_own address the halo rows a node partition received (round 4).  The sums of a row meet through LDS float
 * atomics: their order, hence the last bits of a result, may differ between runs. */
int sgp_spmm_colblock_f32(const int32_t* plan, const int32_t* segptr, const int32_t* wg_row0,
                          int32_t n_wg, int32_t n_blocks,
                          const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                          const float* X_halo, int64_t xh_row_stride, int64_t xh_batch_stride, int32_t n_own,
                          float* Y, int64_t y_row_stride, int64_t y_batch_stride,
                          int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                          const int32_t* pred, int32_t run_if, sgp_stream_t stream);
int32_t sgp_spmm_colblock_rows_cap(void);
int32_t sgp_spmm_colblock_round_pad(void);

/* Limits of the tiled kernel: largest per-tile distinct-column count it can stage for
 * `feat` (0 = feat unsupported; feat must be a multiple of 64), largest tile height and
 * largest padded per-row edge count. */
int32_t sgp_spmm_tiled_max_union(int32_t feat);
int32_t sgp_spmm_tiled_max_tile_rows(void);
int32_t sgp_spmm_tiled_max_row_edges(void);

/* ------------------------------------------------------------- Reservoir ---
 * One leaky-ESN layer over the whole sequence (time loop on the device):
 *   h[t] = (1 - alpha) * h[t-1] + alpha * act(x[t] W_ih^T + b + h[t-1] W_hh^T)
 * x:   [T, N, F] with strides (x_step_stride, x_row_stride, 1)
 * out: [T, N, R] with strides (out_step_stride, out_row_stride, 1)
 * w_ih [R, F], w_hh [R, R], b [R]: row-major device arrays, exactly the
 *   reference's parameters (lib/nn/reservoir/reservoir.py:43-52).
 * h_state: optional [N, R] contiguous; if non-NULL it is the initial state and
 *   receives the final one (T-chunked streaming / resume); NULL = zeros.
 * workspace: device scratch of sgp_reservoir_workspace_bytes(F, R) bytes
 *   (weights re-laid out in MFMA fragment order); contents need not persist.
 * Multi-layer reservoirs (reservoir.py:174-176) are run layer by layer: layer
 * l > 0 reads layer l-1's slot of the output as its x.
 * Arithmetic: fp32 products on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), fp32 accumulation.  Layers with
 * R = 32 or 64 and F = 16, 32 or 64, and large layers with R = 256 and F = 32, 64 or 128 (16-byte aligned rows)
 * take csrc/reservoir_bf3.h instead: every operand as
 * three bf16 pieces (24 bits, no scale, no bound on the values), six piece products per product on
 * v_mfma_f32_16x16x32_bf16, fp32 accumulation -- as close to the fp64 result as a CPU fp32 evaluation
 * (tests/test_gpu_reservoir_bf3.py); SGP_TUNE=res_bf3=0 keeps the fp32 matrix cores for them too.
 * Small problems (<= 512 tiles of 16 nodes, 32 < R <= 128, F <= 64: csrc/reservoir_splitj_bf3.h) form the same
 * three-piece products, except the RECURRENT ones under act = SGP_ACT_TANH: there the state lies in [-1, 1] (a convex
 * combination of the old state and a tanh) and is cut into two fp16 pieces of 2^14 h, row j of w_hh into two fp16
 * pieces under its own power-of-two scale (largest entry at 2^13 .. 2^14); hi hi + hi lo + lo hi on
 * v_mfma_f32_16x16x32_f16, fp32 accumulation, the row sum scaled back exactly.  Per operand: relative 2^-23 down to
 * 2^-16 of its bound (1 for the state, the row's largest |w|), absolute 2^-38 of the bound below -- far under the
 * 3e-7 absolute accuracy of SGP_ACT_TANH itself.  A workgroup (16 nodes) whose INITIAL h_state has an entry outside
 * [-1, 1] (or NaN) runs the three-piece loop instead, decided on the device; the other activations always do.
 * The large-N forms (R = 32 / 64 with F = 16 / 32 / 64; R = 256 with F = 32 / 64 / 128, >= 2048 node tiles) do the same under SGP_ACT_TANH (pack_weights_bf3h / pack_weights_sbf3h: the row scale
 * 2^(e_j + 14) is folded into the bias and the input fragments, so the accumulator carries it as a whole and is scaled
 * back once, exactly): launched alone when h_state is NULL; with an h_state, a test kernel writes "some entry lies outside
 * [-1, 1] or is NaN" into a device word and the two-piece instance runs under word == 0, the three-piece instance
 * under word == 1 behind it (whole launch; no host round trip).
 * All of this only for 0 <= alpha <= 1 (the leak is convex); any other leaking rate keeps three bf16 pieces.
 * SGP_TUNE=res_h16=0 keeps three bf16 pieces for the bounded state too.
 */
int64_t sgp_reservoir_workspace_bytes(int32_t F, int32_t R);
int sgp_reservoir_f32(const float* x, int64_t x_row_stride, int64_t x_step_stride,
                      const float* w_ih, const float* w_hh, const float* b,
                      double alpha, int32_t act,
                      float* out, int64_t out_row_stride, int64_t out_step_stride,
                      float* h_state, void* workspace,
                      int32_t T, int32_t N, int32_t F, int32_t R,
                      sgp_stream_t stream);

/* The same layer over n_pieces TIME PIECES side by side (small graphs: the sequential chain of
 * lib/nn/reservoir/reservoir.py:170-183 runs on ceil(N / 16) workgroups -- 21 of 256 CUs at N = 325 -- so the time axis
 * is the only parallelism left).  Workgroup (node tile, p) runs t_piece steps (the last piece: t_last <= t_piece) of
 *     x + p * x_piece_stride, out + p * out_piece_stride, h_state + p * N * R        (strides in floats)
 * from the state h_state[p] and leaves its final state there; no_store != 0 writes no output rows (a piece's WARM-UP:
 * started from zero some hundred steps early, a contractive recurrence arrives at the true state).  The recurrence
 * itself is not changed: what makes the pieces a valid evaluation of the sequence is the caller's comparison of every
 * piece's end state with its successor's warmed-up start (sgp_amd/nn/reservoir/reservoir.py::run_time_parallel, on the
 * device) and the sequential launch -- this entry with n_pieces = 1 under `pred` -- that repairs a rejected splice.
 * pred / run_if: launch predicate as on the hop entries.  Served by the split-J bf16-piece kernel only
 * (csrc/reservoir_splitj_bf3.h: 32 < R <= 128, F <= 64, <= 512 node tiles); anything else returns SGP_EUNSUP. */
int sgp_reservoir_pieces_f32(const float* x, int64_t x_row_stride, int64_t x_step_stride,
                             const float* w_ih, const float* w_hh, const float* b,
                             double alpha, int32_t act,
                             float* out, int64_t out_row_stride, int64_t out_step_stride,
                             float* h_state, void* workspace,
                             int32_t t_piece, int32_t t_last, int32_t n_pieces,
                             int64_t x_piece_stride, int64_t out_piece_stride, int32_t no_store,
                             int32_t N, int32_t F, int32_t R,
                             const int32_t* pred, int32_t run_if, sgp_stream_t stream);

/* All L layers of a narrow stacked reservoir in ONE launch (lib/nn/reservoir/reservoir.py:170-180:
 * the reference steps every layer inside one time step, layer l consuming layer l-1's new state).
 * The layers are pipelined as a wavefront over the waves of a workgroup (layer l on step t while
 * layer l+1 is on step t-1, hand-off through LDS), layer inputs never travel through HBM.
 *   w_ih / w_hh / b / alpha: HOST arrays of length L; w_ih[l] ([R, F] for l = 0, else [R, R]),
 *        w_hh[l] ([R, R]), b[l] ([R]) are DEVICE pointers to the reference's parameters
 *        (reservoir.py:42-52), alpha[l] the layer's leaking rate (reservoir.py:109-123)
 *   out: [T, N, >= L*R] strides (out_step_stride, out_row_stride, 1); layer l fills columns
 *        l*R .. (l+1)*R-1 of every step (the layer-major order of reservoir.py:181-183)
 *   h_state: optional [L, N, R] contiguous, initial states in / final states out; NULL = zeros
 *   workspace: sgp_reservoir_fused_workspace_bytes(F, R, L) bytes of device scratch, 16-byte aligned
 * Built for F <= 64, R <= 64, 2 <= L <= 16 with all layers' weights in LDS
 * (sgp_reservoir_fused_supported); other shapes: SGP_EUNSUP, run sgp_reservoir_f32 per layer.
 * Same exact-fp32 MFMA products as sgp_reservoir_f32; deeper layers sum their input part in the
 * k order of the recurrent part, so results agree to rounding, not bitwise. */
int64_t sgp_reservoir_fused_workspace_bytes(int32_t F, int32_t R, int32_t L);
int32_t sgp_reservoir_fused_supported(int32_t F, int32_t R, int32_t L);
int sgp_reservoir_fused_f32(const float* x, int64_t x_row_stride, int64_t x_step_stride,
                            const float* const* w_ih, const float* const* w_hh, const float* const* b,
                            const double* alpha, int32_t act,
                            float* out, int64_t out_row_stride, int64_t out_step_stride,
                            float* h_state, void* workspace,
                            int32_t T, int32_t N, int32_t F, int32_t R, int32_t L,
                            sgp_stream_t stream);
/* Same, and the column sums of the produced states per node tile of 16:
 *   tile_sums[ceil(N / 16)][T][L*R] (contiguous, 16-byte aligned), entry (k, t, :) = sum over the
 *   nodes 16 k .. 16 k + 15 (< N) of out[t, node, :L*R].
 * The global_attr block of lib/nn/encoders/sgp_spatial_encoder.py:32-34 is the mean over nodes of
 * exactly this tensor: summing the tiles (sgp_node_mean_bcast_f32 with Y = NULL over the [T, tiles, D]
 * view) replaces a second pass over the whole block.  NULL = sgp_reservoir_fused_f32. */
int sgp_reservoir_fused_sums_f32(const float* x, int64_t x_row_stride, int64_t x_step_stride,
                                 const float* const* w_ih, const float* const* w_hh, const float* const* b,
                                 const double* alpha, int32_t act,
                                 float* out, int64_t out_row_stride, int64_t out_step_stride,
                                 float* h_state, void* workspace, float* tile_sums,
                                 int32_t T, int32_t N, int32_t F, int32_t R, int32_t L,
                                 sgp_stream_t stream);

/* --------------------------------------------------------------- DynGESN ---
 * The graph echo-state baseline (lib/nn/reservoir/graph_reservoir.py:85-93, stepped by
 * tsl/nn/blocks/encoders/gcrnn.py:67-93):
 *     h' = (1 - alpha) h + alpha * act( x W_ih^T + b + A_hat (h W_hh^T) )
 * sgp_gesn_f32 runs a whole sequence through all L layers (the _GraphRNN loop of gcrnn.py:67-93
 * with _cat_states_layers): two launches per (time step, layer), issued from C:
 *   x:   [T, N, F] strides (x_step_stride, x_row_stride, 1)
 *   out: [T, N, L*R] strides (out_step_stride, out_row_stride, 1); layer i fills columns
 *        i*R .. (i+1)*R-1 of every step
 *   w_ih / w_hh / b / alpha: HOST arrays of length L; w_ih[i] ([R, F] for i = 0, else [R, R]),
 *        w_hh[i] ([R, R]) and b[i] ([R]) are DEVICE pointers to the reference's parameters
 *        (graph_reservoir.py:44-52), alpha[i] the layer's leaking rate
 *   h_state: [L, N, R] contiguous, initial states in, final states out (zeros = cold start)
 *   workspace: sgp_gesn_workspace_bytes(N, R, L) bytes of device scratch, 16-byte aligned
 *   rowptr / col / val: the normalised operator in CSR, rows = targets (N rows)
 * Building blocks, also exported (one cell step = graph_reservoir.py:85-93):
 *   sgp_gemm_nt_f32      C[m, n] = sum_k A[m, k] W[n, k] (+ bias[n])   (F.linear; fp32 MFMA)
 *   sgp_gesn_update_f32  h_out[i, :] = (1 - alpha) h_in[i, :]
 *                                     + alpha * act(p[i, :] + sum_e val[e] z[col[e], :])
 *                        (torch_sparse matmul of :91 fused with activation and leak; also
 *                        writes the row into out_row[i * out_stride + 0:R], the step's slot of
 *                        the [T, N, L*R] embedding).  z, p, h_in, h_out: contiguous [N, R].
 */
int64_t sgp_gesn_workspace_bytes(int32_t N, int32_t R, int32_t L);
/* sgp_gesn_tune(persistent): 1 (default, also SGP_TUNE=gesn_persistent=1) lets sgp_gesn_f32 run the
 * sequence in ONE cooperative launch per 256 steps when the shape allows it (R % 16 == 0, R <= 384,
 * L <= 8, the (layer, column group, row tile) items fit one workgroup per CU; csrc/gesn_persist.hip:
 * layers as a wavefront, weights in registers, one grid barrier per step); 0 = always two launches
 * per (step, layer); < 0 = query.  Returns the setting. */
int sgp_gesn_tune(int32_t persistent);
int sgp_gesn_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                 const float* x, int64_t x_row_stride, int64_t x_step_stride,
                 const float* const* w_ih, const float* const* w_hh, const float* const* b,
                 const double* alpha, int32_t act,
                 float* out, int64_t out_row_stride, int64_t out_step_stride,
                 float* h_state, void* workspace,
                 int32_t T, int32_t N, int32_t F, int32_t R, int32_t L, sgp_stream_t stream);
int sgp_gemm_nt_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                    float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, sgp_stream_t stream);
int sgp_gesn_update_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                        const float* z, const float* p, const float* h_in,
                        double alpha, int32_t act,
                        float* h_out, float* out_row, int64_t out_stride,
                        int32_t n_nodes, int32_t R, sgp_stream_t stream);

/* ------------------------------------------------------------ Node mean ----
 * Y[b, i, 0:feat] = (1 / n_rows) * sum_j X[b, j, 0:feat]   for every i.
 * With partial != NULL the kernel instead writes the un-normalised column sums
 * to partial[b, 0:feat] (contiguous) and leaves Y alone (multi-GPU: all-reduce
 * the partial sums, then call sgp_bcast_rows_f32).
 */
int sgp_node_mean_bcast_f32(const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                            float* Y, int64_t y_row_stride, int64_t y_batch_stride,
                            float* partial,
                            int32_t n_rows, int32_t batch, int32_t feat,
                            sgp_stream_t stream);
/* Y[b, i, 0:feat] = scale * src[b, 0:feat]  for i in [0, n_rows). */
int sgp_bcast_rows_f32(const float* src, float scale,
                       float* Y, int64_t y_row_stride, int64_t y_batch_stride,
                       int32_t n_rows, int32_t batch, int32_t feat,
                       sgp_stream_t stream);

/* --------------------------------------------------------- Copy / gather ---
 * Y[b, i, 0:feat] = X[b, i, 0:feat] (strided block copy into an output slot). */
int sgp_copy_rows_f32(const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                      float* Y, int64_t y_row_stride, int64_t y_batch_stride,
                      int32_t n_rows, int32_t batch, int32_t feat,
                      sgp_stream_t stream);
/* out[k, 0:feat] = X[step[k], node[k], 0:feat]   (IID sampling of the embedding).
 * Also used to pack halo rows: step = NULL gathers node[k] for every b:
 *   out[b, k, :] = X[b, node[k], :]. */
int sgp_gather_rows_f32(const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                        const int32_t* step, const int32_t* node, int32_t n_index,
                        float* out, int64_t out_row_stride, int64_t out_batch_stride,
                        int32_t batch, int32_t feat, sgp_stream_t stream);

/* ------------------------------------------------- Decoder first layer ("next" row f4) ---
 * Grouped 1x1 convolution of lib/nn/models/sgp_model.py:41-52 (nn.Conv1d(input_size, out_channels,
 * kernel_size=1, groups) between two Rearranges, then the activation): for every row
 *   out[row, g*oc + o] = act(bias[g*oc + o] + sum_i W[g*oc + o, i] * X_row[g*ic + i]),  g < groups.
 * Row k is X[k, :] (x_row_stride) or, when step / node are given, X[step[k], node[k], :] -- the IID
 * gather of lib/datasets/iid_dataset.py:66-69 fused in, so the sampled batch never exists in HBM.
 * W is Conv1d's weight [groups*oc, ic] (trailing kernel axis of size 1 dropped), handed over in the
 * fragment order produced by sgp_grouped_linear_pack_f32 (sgp_grouped_linear_packed_floats floats).
 * act: 0 = none, 1 = relu, 2 = silu.  fp32 MFMA (v_mfma_f32_16x16x4_f32), exact products. */
int64_t sgp_grouped_linear_packed_floats(int32_t groups, int32_t ic, int32_t oc);
int sgp_grouped_linear_pack_f32(const float* w, float* packed, int32_t groups, int32_t ic, int32_t oc,
                                sgp_stream_t stream);
int sgp_grouped_linear_f32(const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                           const int32_t* step, const int32_t* node,
                           const float* w_packed, const float* bias, int32_t act,
                           float* out, int64_t out_row_stride,
                           int32_t n_rows, int32_t groups, int32_t ic, int32_t oc,
                           sgp_stream_t stream);
/* Training the decoder (sgp_model.py:41-52 is a trained layer): the same forward that also writes the
 * pre-activation values `pre[n_rows, groups*oc]` (NULL = sgp_grouped_linear_f32), and the three pieces
 * of its backward pass.  With z = W x + b, y = act(z), incoming gradient dy:
 *   sgp_grouped_linear_dact_f32       dz = dy * act'(z)            (dz, pre contiguous [n_rows, width])
 *   sgp_grouped_linear_transpose_f32  W [groups*oc, ic] -> W^T laid out as the weight [groups*ic, oc] of
 *                                     the layer with ic and oc exchanged: dx = that layer (packed with
 *                                     sgp_grouped_linear_pack_f32, zero bias, act 0) applied to dz
 *   sgp_grouped_linear_wgrad_f32      dW[g*oc + o, i] = sum_row dz[row, g*oc + o] * X_row[g*ic + i]
 *                                     (rows as in the forward, IID gather included; fp32 MFMA with the
 *                                     rows as contraction index, row slices meet through float atomics)
 *   db = column sums of dz (sgp_node_mean_bcast_f32 with Y = NULL).
 * Dropout(p) behind the activation (sgp_model.py:50; training mode only): `dropout_p` in [0, 1) and a
 * 64-bit `seed` per call; element (row, column) is kept and scaled by 1 / (1 - p) when word 0 of
 * Philox4x32-10(key = seed, counter = row * width + column) >= p * 2^32.  The forward applies the factor
 * to `out` (not to `pre`), sgp_grouped_linear_dact_f32 with the same (p, seed) to dz. */
int sgp_grouped_linear_fwd_f32(const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                               const int32_t* step, const int32_t* node,
                               const float* w_packed, const float* bias, int32_t act,
                               float* out, int64_t out_row_stride, float* pre,
                               double dropout_p, uint64_t seed,
                               int32_t n_rows, int32_t groups, int32_t ic, int32_t oc,
                               sgp_stream_t stream);
int sgp_grouped_linear_dact_f32(const float* dy, int64_t dy_row_stride, const float* pre, int32_t act,
                                double dropout_p, uint64_t seed,
                                float* dz, int64_t n_rows, int32_t width, sgp_stream_t stream);
int sgp_grouped_linear_transpose_f32(const float* w, float* wt, int32_t groups, int32_t ic, int32_t oc,
                                     sgp_stream_t stream);
int sgp_grouped_linear_wgrad_f32(const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                                 const int32_t* step, const int32_t* node,
                                 const float* dz, float* dw,
                                 int32_t n_rows, int32_t groups, int32_t ic, int32_t oc,
                                 sgp_stream_t stream);


/* -------------------------------------------------------------- Timing -----
 * HIP-event helpers so that Python can time kernels on the stream they were
 * launched on without importing a HIP binding. */
int sgp_event_create(void** ev);
int sgp_event_destroy(void* ev);
int sgp_event_record(void* ev, sgp_stream_t stream);
int sgp_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on stop */

#ifdef __cplusplus
}
#endif
#endif /* SGP_AMD_H */

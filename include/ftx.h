/* libftx — C ABI of the MI355X (gfx950) hot path of FusionTransformer.
 *
 * The reference (aliabdelkader/FusionTransformer) has no FFI layer of its own:
 * its native work is reached through torchsparse v1.1.0 (`spf.*`, `spnn.*`),
 * timm 0.4.9 and torch.  Each entry point below replaces one of those native
 * ops at the call site cited next to it (paths relative to the reference
 * root).  The host side (fusiontransformer_amd/) binds these with ctypes; the
 * binding a reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *     row-major, contiguous; the caller owns every buffer (inputs, outputs and
 *     workspaces) and the library keeps nothing beyond the call;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it,
 *     nothing synchronises the host (safe for hipGraph capture) unless stated;
 *   - return 0 on success, a negative FTX_E* code otherwise; the message is in
 *     the thread-local ftx_last_error(); nothing throws, nothing aborts;
 *   - index dtype on the device is int32 (-1 = absent); hashes are int64.
 */
#ifndef FTX_H
#define FTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FTX_OK 0
#define FTX_EINVAL -1  /* bad argument (null pointer, negative size, unsupported shape) */
#define FTX_ELAUNCH -2 /* HIP runtime reported an error at launch */
#define FTX_EWORKSPACE -3 /* workspace too small */

/* library version (major*10000 + minor*100 + patch) and last error text */
int ftx_version(void);
const char *ftx_last_error(void);

/* ---- per-(device, stream) ticket buffer of the statistics kernels (ftx_spconv_reduce_stats, ftx_bn_train_fwd / _bwd) ----
 * Those kernels hand their column totals to "the last block to finish", which needs a few ticket counters and group rows that persist
 * across the blocks of a launch (csrc/ftx_lastblock.h).  The caller owns them: allocate ftx_stream_scratch_bytes() bytes of device
 * memory (256-byte aligned), attach them to the stream (the library zeroes the tickets on that stream) and keep them alive until
 * ftx_stream_scratch_release(stream) or the end of the process.  A stream nobody attached a buffer to gets a library allocation at its
 * first use (outside stream capture), freed by ftx_stream_scratch_release.  ftx_stream_scratch_reset zeroes the tickets on the stream:
 * call it after a kernel of the library died mid-flight (a dirty ticket otherwise trips a device-side assert in the next launch).
 * Keyed by (current device, stream).  No other state of the library outlives a call. */
size_t ftx_stream_scratch_bytes(void);
int ftx_stream_scratch_attach(void *stream, void *buffer, size_t bytes);
int ftx_stream_scratch_reset(void *stream);
int ftx_stream_scratch_release(void *stream);

/* ---- coordinate hashing ------------------------------------------------ */

/* spf.sphash(C): models/utils.py:19,49,79.  coords (n,4) int32 [x,y,z,b] -> out (n) int64. */
int ftx_hash(const int32_t *coords, int64_t n, int64_t *out, void *stream);

/* spf.sphash(C, off): models/utils.py:74-78.  offsets (k,3) int32 -> out (k,n) int64. */
int ftx_hash_kernel(const int32_t *coords, int64_t n, const int32_t *offsets, int32_t k, int64_t *out, void *stream);

/* torch.floor(z.C[:, :3] / s).int() * s ++ z.C[:, -1].int(): models/utils.py:44-48,75-78.
 * pc (n,4) float32 -> out (n,4) int32. */
int ftx_floor_coords(const float *pc, int64_t n, int32_t stride, int32_t *out, void *stream);

/* ---- hash table (spf.sphashquery): models/utils.py:21,50,80 -------------- */

/* Smallest legal capacity (a power of two >= 2n) for n keys. */
int64_t ftx_hashtable_capacity(int64_t n);
/* Build: table_keys (capacity) int64 and table_vals (capacity) int32 are fully
 * (re)initialised here.  Duplicate keys keep the smallest row index. */
int ftx_hashtable_build(const int64_t *keys, int64_t n, int64_t *table_keys, int32_t *table_vals, int64_t capacity, void *stream);
/* Query: out[i] = row of queries[i] in the keys the table was built from, or -1. */
int ftx_hashtable_query(const int64_t *queries, int64_t nq, const int64_t *table_keys, const int32_t *table_vals, int64_t capacity, int32_t *out, void *stream);

/* spf.spcount(idx, m): models/utils.py:22,51.  counts (m) int32 is zeroed here. */
int ftx_count(const int32_t *idx, int64_t n, int32_t *counts, int64_t m, void *stream);

/* torch.unique(hash) (sorted): models/utils.py:20 and torchsparse spdownsample.
 * uniq (n) int64 receives the sorted unique keys, first_index (n) int32 the row of
 * the first occurrence of each, n_unique (1) int32 (device) their number.
 * Rows past n_unique are unspecified. */
size_t ftx_unique_workspace_bytes(int64_t n);
int ftx_unique_sorted(const int64_t *keys, int64_t n, int64_t *uniq, int32_t *first_index, int32_t *n_unique, void *workspace, size_t workspace_bytes, void *stream);
/* rank[i] = position of queries[i] in sorted[0 .. min(*n_sorted, capacity)) (ascending, unique; the count is read ON THE DEVICE), -1 when
 * absent.  On the output of ftx_unique_sorted this is numpy.unique's return_inverse, i.e. the inverse map of torchsparse
 * sparse_quantize(..., return_invs=True) (data/semantic_kitti/semantic_kitti_dataloader.py:231). */
/* out[i,:] = points[i,:] . R (R: 9 floats, row-major, on the HOST) with the rounding of numpy's float32 `points.dot(rot_matrix)` --
 * one fused multiply-add per step of K = 3 --: the rotation / flip of data/utils/augmentation_3d.py:22-41 on the device */
int ftx_rotate_points(const float *points, int64_t n, const float *rot_host, float *out, void *stream);
int ftx_sorted_rank(const int64_t *sorted, const int32_t *n_sorted, int64_t capacity, const int64_t *queries, int64_t nq, int32_t *rank, void *stream);

/* torchsparse spdownsample coordinate rule: floor(c / ratio) * ratio on x,y,z, b kept.
 * coords (n,4) int32 -> out (n,4) int32.  (inside spnn.Conv3d(stride=2): models/spvcnn.py:105,111,117,123) */
int ftx_downsample_coords(const int32_t *coords, int64_t n, int32_t ratio, int32_t *out, void *stream);

/* out[i,:] = src[index[i],:] for int32 rows of width 4 (coordinate gather). */
int ftx_gather_coords(const int32_t *src, const int32_t *index, int64_t n, int32_t *out, void *stream);

/* Every level of the U-Net in ONE pass (models/spvcnn.py:104-126 visits strides 1, 2, 4, 8, 16; torchsparse's spdownsample derives
 * level l+1 from level l, one hash + torch.unique + size read per level): points (n,4) int32 = the floored point coordinates;
 * for each of the n_levels strides (HOST array, each >= 1, at most 8) the set unique(floor_div(p, s) * s) in ascending hash order.
 * uniq (n_levels*n) int64: the levels' sorted unique hashes back to back; first_index (n_levels*n): the point row of the first
 * occurrence of each; level_off (n_levels+1) int32 DEVICE: where each level's run starts (level_off[n_levels] = total).  The caller
 * reads level_off once (the only host read of the whole coordinate build besides the pair counts) and takes the level's coordinates
 * with ftx_level_coords(points, first_index + level_off[l], n_l, stride, out).  Same sets, order and coordinates as the chained form. */
size_t ftx_levels_workspace_bytes(int64_t n, int32_t n_levels);
int ftx_levels_unique(const int32_t *points, int64_t n, const int32_t *strides, int32_t n_levels, int64_t *uniq, int32_t *first_index, int32_t *level_off, int64_t *sorted_keys, int32_t *order, void *workspace, size_t workspace_bytes, void *stream);
/* sorted_keys / order (n_levels*n each, may be NULL): the (level tag << 60 | hash) keys in sorted order and the point row of each; level l
 * owns [l*n, (l+1)*n): its points sorted by voxel, stable.  ftx_level_segments turns that slice into the sorted segments of spvoxelize at
 * the level's stride (what ftx_segment_build(idx_query, n, m) would return) without another sort: seg_off (m+1) from the level's unique
 * hashes (`uniq + level_off[l]`, m = n_l of them). */
int ftx_level_segments(const int64_t *sorted_keys, int64_t n, const int64_t *uniq, int64_t m, int32_t level, int32_t *seg_off, void *stream);
int ftx_level_coords(const int32_t *points, const int32_t *first_index, int64_t n, int32_t stride, int32_t *out, void *stream);

/* ---- kernel maps (inside spnn.Conv3d): models/spvcnn.py:26-30,42-46,57-72,99-101 */

/* nbr[k, o] = row in the table's key set of (out_coords[o] + offsets[k]), or -1.
 * Fuses sphash(out_coords, offsets) + sphashquery.  nbr is (k, n_out) int32. */
int ftx_kernel_map_build(const int32_t *out_coords, int64_t n_out, const int32_t *offsets, int32_t k, const int64_t *table_keys, const int32_t *table_vals, int64_t capacity, int32_t *nbr, void *stream);
/* Pair list of a kernel map (what torchsparse's convert_neighbor_map builds): the valid
 * (k, o) entries of nbr compacted in (k, o) order.
 *   step 1  ftx_kernel_map_count: pos (k, n_out) receives the exclusive scan of the validity
 *           flags, koff (k+1, device) the first pair of every offset; koff[k] = number of pairs
 *           (the caller reads it back to size the pair arrays).
 *   step 2  ftx_kernel_map_pairs: pair_in[p] / pair_out[p] = input / output row of pair p;
 *           pos[k,o] = p or -1; pos_t (k, n_in): pos_t[k, i] = p of the pair (k, i) or -1. */
size_t ftx_kernel_map_count_workspace_bytes(int64_t n_out, int32_t k);
int ftx_kernel_map_count(const int32_t *nbr, int64_t n_out, int32_t k, int32_t *pos, int32_t *koff, void *workspace, size_t workspace_bytes, void *stream);
int ftx_kernel_map_pairs(const int32_t *nbr, int64_t n_out, int64_t n_in, int32_t k, int32_t *pos, int32_t *pos_t, int32_t *pair_in, int32_t *pair_out, int64_t n_pairs, void *stream);

/* spf.calc_ti_weights: models/utils.py:81-82.  pc (n,4) float32 (integer-valued or not),
 * idx (n,8) int32 (point-major, as after the transpose at utils.py:83), weights (n,8) float32. */
int ftx_trilinear_weights(const float *pc, const int32_t *idx, int64_t n, int32_t scale, float *weights, void *stream);

/* ---- point <-> voxel feature movement ----------------------------------- */

/* spf.spvoxelize fwd: models/utils.py:24-27,58.  out (m,c) is zeroed here;
 * out[idx[i]] += feats[i] / counts[idx[i]]. */
int ftx_voxelize_fwd(const float *feats, const int32_t *idx, const int32_t *counts, int64_t n, int32_t c, int64_t m, float *out, void *stream);
/* bwd: grad_feats[i] = grad_out[idx[i]] / counts[idx[i]] (0 where idx<0). */
int ftx_voxelize_bwd(const float *grad_out, const int32_t *idx, const int32_t *counts, int64_t n, int32_t c, int64_t m, float *grad_feats, void *stream);

/* spf.spdevoxelize fwd: models/utils.py:87,99.  out[i] = sum_k w[i,k] * feats[idx[i,k]]. */
int ftx_devoxelize_fwd(const float *feats, const int32_t *idx, const float *weights, int64_t n, int32_t c, int64_t m, float *out, void *stream);
/* bwd: grad_feats (m,c) zeroed here; grad_feats[idx[i,k]] += w[i,k] * grad_out[i]. */
int ftx_devoxelize_bwd(const float *grad_out, const int32_t *idx, const float *weights, int64_t n, int32_t c, int64_t m, float *grad_feats, void *stream);

/* Sorted-segment forms of the two scatter sides (no float atomics, bit-reproducible).
 * ftx_segment_build: keys (n) int32 = destination row of each entry (<0 or >=m: dropped);
 *   order (n) int32 receives the entry ids sorted by destination (ties in ascending entry id),
 *   seg_off (m+1) int32 the first position of every destination's run.
 * voxelize:   keys = idx (point -> voxel), entries = points.
 * devoxelize: keys = idx (n,8) flattened with zero-weight corners set to -1, entries = (point, corner). */
size_t ftx_segment_workspace_bytes(int64_t n, int64_t m);
int ftx_segment_build(const int32_t *keys, int64_t n, int64_t m, int32_t *order, int32_t *seg_off, void *workspace, size_t workspace_bytes, void *stream);
/* out[v] = mean of feats[p] over the points p of voxel v (0 for empty voxels); out (m,c) fully written. */
int ftx_voxelize_fwd_sorted(const float *feats, const int32_t *order, const int32_t *seg_off, int64_t n, int32_t c, int64_t m, float *out, void *stream);
/* grad_feats[v] = sum over entries e=(p,k) of voxel v of weights[e] * grad_out[p]; weights (n,8) flattened, n = points. */
int ftx_devoxelize_bwd_sorted(const float *grad_out, const float *weights, const int32_t *order, const int32_t *seg_off, int64_t n, int32_t c, int64_t m, float *grad_feats, void *stream);

/* out[v] = sum of src[e] over the entries of segment v, in entry order (generic atomic-free scatter-add); out (m,c) fully written. */
int ftx_segment_sum(const float *src, const int32_t *order, const int32_t *seg_off, int64_t n, int32_t c, int64_t m, float *out, void *stream);

/* ---- 2D -> 3D lift ------------------------------------------------------- */

/* Fused `nn.Upsample((H,W))` (nearest) + per-point gather of
 * Net2DBillinear.get_img_feats: models/image_models_billinear.py:113,117-124.
 * grid (b, gh, gw, c) float32 channels-last (= the (B,576,96) token layout);
 * img_idx (n,2) int64 (row,col) in the (H,W) lift image; point_batch (n) int32;
 * out (n,c).  The (H,W,c) map is never materialised. */
int ftx_lift_gather_fwd(const float *grid, const int64_t *img_idx, const int32_t *point_batch, int64_t n, int32_t b, int32_t gh, int32_t gw, int32_t c, int32_t H, int32_t W, float *out, void *stream);
/* bwd: grad_grid (b,gh,gw,c) zeroed here, scatter-add of grad_out rows. */
int ftx_lift_gather_bwd(const float *grad_out, const int64_t *img_idx, const int32_t *point_batch, int64_t n, int32_t b, int32_t gh, int32_t gw, int32_t c, int32_t H, int32_t W, float *grad_grid, void *stream);

/* cells[i] = flat index (frame, source row, source col) of point i's cell in the (b, gh, gw) grid (-1 if out of range):
 * keys for ftx_segment_build so that the lift backward is ftx_segment_sum over the grid cells (no float atomics). */
int ftx_lift_cells(const int64_t *img_idx, const int32_t *point_batch, int64_t n, int32_t b, int32_t gh, int32_t gw, int32_t H, int32_t W, int32_t *cells, void *stream);

/* `nn.Upsample((oh,ow))` nearest on NCHW: models/image_models_billinear.py:17,41.
 * in (b,c,ih,iw) -> out (b,c,oh,ow). bwd zeroes grad_in and scatter-adds. */
int ftx_resample_nearest_fwd(const float *in, int32_t b, int32_t c, int32_t ih, int32_t iw, int32_t oh, int32_t ow, float *out, void *stream);
int ftx_resample_nearest_bwd(const float *grad_out, int32_t b, int32_t c, int32_t ih, int32_t iw, int32_t oh, int32_t ow, float *grad_in, void *stream);

/* Fused Net2DBillinear.sample_down: Conv1x1(3->3) + ReLU + BatchNorm2d(3) on the full-resolution
 * image + nearest pick to (oh, ow) (models/image_models_billinear.py:8-24,41,131).
 * img (b,3,h,w); conv_w (3,3) row-major [out][in]; out (b,3,oh,ow); saved (33) float64 device scratch
 * kept for the backward (training statistics of the full-resolution map).  training != 0: batch
 * statistics (running_* updated when non-NULL); training == 0: running statistics.
 * bwd (training mode) writes the four parameter gradients; the image gets none (it is an input). */
size_t ftx_sample_down_workspace_bytes(void);
int ftx_sample_down_fwd(const float *img, int32_t b, int32_t h, int32_t w, int32_t oh, int32_t ow, const float *conv_w, const float *conv_b, const float *gamma, const float *beta, float *running_mean, float *running_var, float momentum, float eps, int32_t training, float *out, double *saved, void *workspace, size_t workspace_bytes, void *stream);
int ftx_sample_down_bwd(const float *img, const float *grad_out, int32_t b, int32_t h, int32_t w, int32_t oh, int32_t ow, const float *conv_w, const float *conv_b, const float *gamma, const double *saved, float *grad_conv_w, float *grad_conv_b, float *grad_gamma, float *grad_beta, void *workspace, size_t workspace_bytes, void *stream);

/* ---- sparse convolution (spnn.Conv3d fwd/bwd) ----------------------------
 * Pair-list gather-GEMM + ordered reduce (exact-fp32 MFMA, no float atomics, bit-reproducible):
 *   forward       tmp = pairs_gemm(A=in,   gather=pair_in,  W, 0);  out = reduce(tmp, pos,   n_out)
 *   data grad     tmp = pairs_gemm(A=gout, gather=pair_out, W, 1);  gin = reduce(tmp, pos_t, n_in)
 *   weight grad   dW  = pairs_wgrad(A=in, pair_in, G=gout, pair_out)
 *   transposed conv (models/spvcnn.py:42-46): the same three calls with in/out roles swapped. */

/* tmp[p,:] = A[gather[p],:] @ Wk(p), k(p) from koff.  A (rows_a, ca); gather (n_pairs) int32;
 * W (kvol, ca, co) row-major when w_transposed == 0, (kvol, co, ca) used as W[k]^T when 1;
 * koff (kvol+1) int32 DEVICE; tmp (n_pairs, co) fully written. */
int ftx_spconv_pairs_gemm(const float *A, int64_t rows_a, const int32_t *gather, const float *W, int32_t w_transposed, const int32_t *koff, int64_t n_pairs, int32_t ca, int32_t co, int32_t kvol, float *tmp, void *stream);

/* One-launch convolution when every destination row receives exactly one pair (the strided 2^3 map seen from its
 * fine side: data gradient of the strided conv, forward of the transposed conv, models/spvcnn.py:38-50):
 * out[scatter[p],:] = A[gather[p],:] @ Wk(p).  scatter (n_pairs) int32 must be injective; out (rows_out, co); rows not
 * named by scatter are left untouched.  Replaces pairs_gemm + reduce (no tmp round trip). */
int ftx_spconv_pairs_gemm_scatter(const float *A, int64_t rows_a, const int32_t *gather, const int32_t *scatter, const float *W, int32_t w_transposed, const int32_t *koff, int64_t n_pairs, int32_t ca, int32_t co, int32_t kvol, float *out, int64_t rows_out, void *stream);

/* Output-stationary convolution for thin layers, ONE launch (csrc/ftx_spconv_ostat.hip): out[o,:] = sum over k (ascending) of
 * A[nbr[k,o],:] @ W[k], with nbr (kvol, n_out) int32 the neighbour table of the map (-1 = absent; what ftx_kernel_map_build returns).
 * No pair-row scratch and no reduce pass; bit-identical to ftx_spconv_pairs_gemm + ftx_spconv_reduce on the pair list of the same
 * table.  Replaces the Conv3d of models/spvcnn.py:22-35,98-126 where ftx_spconv_ostat_supported(ca, co, kvol, w_transposed) != 0
 * (ca in {4, 32, 64}, co in {32, 64}).  W (kvol, ca, co), or (kvol, co, ca) when w_transposed.
 * flip != 0: the data gradient of a submanifold convolution on the SAME table (symmetric map, odd kvol): out[i,:] = sum over k of
 * A[nbr[kvol-1-k, i],:] @ W[k] -- pass the output gradient as A, w_transposed = 1 and the forward kernel as W.
 * part != NULL: also the BatchNorm statistics of `out` as ftx_spconv_reduce_stats leaves them -- nb = ftx_spconv_ostat_blocks(n_out)
 * partial rows [nb][2][co] float64 followed by the totals row [2][co] (read by ftx_bn_train_fwd_totals); uses the stream's ticket buffer. */
int32_t ftx_spconv_ostat_supported(int32_t ca, int32_t co, int32_t kvol, int32_t w_transposed);
int32_t ftx_spconv_ostat_blocks(int64_t n_out);
int ftx_spconv_ostat(const float *A, int64_t rows_a, const int32_t *nbr, int64_t n_out, const float *W, int32_t w_transposed, int32_t flip, int32_t ca, int32_t co, int32_t kvol, float *out, double *part, int32_t nb, void *stream);

/* Dense rows on the same tile code: out[r,:] = A[r,:] @ W (+ bias), r < n.  W as above with kvol = 1;
 * bias (co) may be NULL.  Replaces the point-branch nn.Linear layers (models/spvcnn.py:164-180,
 * models/middle_fusion.py:18-29) and the kernel_size=1 spnn.Conv3d (spvcnn.py:71-75). */
int ftx_rows_gemm(const float *A, int64_t n, const float *W, int32_t w_transposed, const float *bias, int32_t ca, int32_t co, float *out, void *stream);

/* out[r,:] = sum over k (ascending) of tmp[pos[k,r],:] for pos >= 0; out (n, co) fully written. */
int ftx_spconv_reduce(const float *tmp, const int32_t *pos, int64_t n, int32_t co, int32_t kvol, float *out, void *stream);

/* The reduce pass that also produces the BatchNorm statistics of its output: part (nb + 1, 2, co) float64 -- nb per-block
 * partial (sum, sum of squares) rows, nb = ftx_spconv_reduce_stats_blocks(n, co), then ONE row of column totals, summed in block
 * order by whichever block finishes last (no second launch; bit-reproducible).  Feed `part + nb*2*co` to
 * ftx_bn_train_fwd_totals: the Conv3d -> BatchNorm pair of every SPVCNN block (models/spvcnn.py:22-35,53-79) then reads the
 * convolution output once.  Uses the stream's ticket buffer (ftx_stream_scratch_* below). */
int32_t ftx_spconv_reduce_stats_blocks(int64_t n, int32_t co);
int ftx_spconv_reduce_stats(const float *tmp, const int32_t *pos, int64_t n, int32_t co, int32_t kvol, float *out, double *part, int32_t nb, void *stream);

/* dW[k] = sum_{p in offset k} A[idx_a[p],:]^T @ G[idx_g[p],:]  -> dW (kvol, ca, cg), fully written.
 * idx_a = idx_g = koff = NULL with kvol = 1: dense rows, dW = A[:n_pairs]^T @ G[:n_pairs]. */
size_t ftx_spconv_pairs_wgrad_workspace_bytes(int64_t n_pairs, int32_t ca, int32_t cg, int32_t kvol);
int ftx_spconv_pairs_wgrad(const float *A, int64_t rows_a, const int32_t *idx_a, const float *G, int64_t rows_g, const int32_t *idx_g, const int32_t *koff, int64_t n_pairs, int32_t ca, int32_t cg, int32_t kvol, float *dW, void *workspace, size_t workspace_bytes, void *stream);
/* Resident blocks per CU the weight-gradient tiling assumes for a (ca, cg) layer / for a kernel instantiation (mi, wmg, ni, wng): a
 * constant table (csrc/ftx_spconv.hip), never a device query, so workspace size, tile length and summation order are functions of the
 * arguments alone.  Exposed so that the build can check the table against the code object (tests/test_cabi.py). */
int32_t ftx_spconv_wgrad_resident_blocks(int32_t ca, int32_t cg);
int32_t ftx_spconv_wgrad_table_blocks(int32_t mi, int32_t wmg, int32_t ni, int32_t wng);

/* ---- BatchNorm1d over rows (+residual, +ReLU): spnn.BatchNorm / nn.BatchNorm1d
 *      models/spvcnn.py:30-31,71-79,100-102,164-180; models/middle_fusion.py:18-22 */

/* Training forward: batch statistics over the n rows (biased var for
 * normalisation, unbiased for running_var, momentum as torch), then
 * y = relu?( (x-mean)*invstd*gamma + beta (+ residual) ).
 * save_mean / save_invstd (c) are outputs for the backward.
 * running_mean / running_var may be NULL (no update).  residual may be NULL. */
size_t ftx_bn_workspace_bytes(int64_t n, int32_t c);
int ftx_bn_train_fwd(const float *x, const float *residual, const float *gamma, const float *beta, float *running_mean, float *running_var, float momentum, float eps, int64_t n, int32_t c, int32_t relu, float *y, float *save_mean, float *save_invstd, void *workspace, size_t workspace_bytes, void *stream);
/* Training-mode BatchNorm forward from the column totals (2, c) float64 that ftx_spconv_reduce_stats left: one launch that
 * derives mean / invstd, updates the running statistics, stores save_mean / save_invstd and applies. */
int ftx_bn_train_fwd_totals(const float *x, const float *residual, const float *gamma, const float *beta, float *running_mean, float *running_var, float momentum, float eps, int64_t n, int32_t c, int32_t relu, float *y, float *save_mean, float *save_invstd, const double *totals, void *stream);

/* Eval forward with running statistics. */
int ftx_bn_eval_fwd(const float *x, const float *residual, const float *gamma, const float *beta, const float *running_mean, const float *running_var, float eps, int64_t n, int32_t c, int32_t relu, float *y, void *stream);
/* Training backward.  ReLU mask (relu != 0): when the forward had NO residual (grad_residual == NULL) and `beta` is given, the forward
 * output is RECOMPUTED from x -- bit for bit the value the forward wrote -- and y is not read (it may be NULL): a third of this pass's
 * traffic; otherwise y, the forward output, supplies the mask.
 * grad_x (n,c), grad_residual (n,c, may be NULL), grad_gamma (c), grad_beta (c). */
int ftx_bn_train_bwd(const float *grad_y, const float *x, const float *y, const float *gamma, const float *beta, const float *save_mean, const float *save_invstd, int64_t n, int32_t c, int32_t relu, float *grad_x, float *grad_residual, float *grad_gamma, float *grad_beta, void *workspace, size_t workspace_bytes, void *stream);

/* ---- optimizer step: torch.optim.Adam (L2 weight decay, no amsgrad) over every parameter tensor in one launch ----
 * (common/solver/build.py:7-20 builds the optimizer, modules/SemanticTrainer.py:141-209 steps it once per batch.)
 * table: n_tensors records of ftx_adam_tensor_bytes() bytes in DEVICE memory, little-endian, in this order:
 *   float *param; float *exp_avg; float *exp_avg_sq; const float *grad (NULL: tensor skipped this step); int64 numel;
 *   float step_size = lr / (1 - beta1^t); float inv_bc2_sqrt = 1 / sqrt(1 - beta2^t).
 * chunk_tensor / chunk_offset (n_chunks, device): chunk c covers elements [offset, offset + ftx_adam_chunk_elements()) of tensor
 * chunk_tensor[c].  Update per element: g += wd*p; m += (1-beta1)*(g-m); v = beta2*v + (1-beta2)*g*g;
 * p -= step_size * m / (sqrt(v) * inv_bc2_sqrt + eps).  beta1 / beta2 are doubles: 1 - beta is formed in double and then rounded, as torch does. */
int32_t ftx_adam_chunk_elements(void);
int32_t ftx_adam_tensor_bytes(void);
int ftx_adam_step(const void *table, const int32_t *chunk_tensor, const int64_t *chunk_offset, int32_t n_chunks, double beta1, double beta2, float eps, float weight_decay, void *stream);

/* ---- LayerNorm of the ViT blocks, fused with the residual add in front of it (timm Block.forward: models/transformers.py:16-45) ----
 * forward: s = x + y (y may be NULL: then s_out is not written and s = x), h = (s - mean) * rstd * gamma + beta over rows of c floats,
 * c in {256, 512, 768, 1024}; mean / rstd (rows) are outputs for the backward.  y_bias (c, may be NULL): y is the output of a Linear
 * computed WITHOUT its bias, s = x + (y + y_bias) -- the rounding order of the GEMM's own bias epilogue. */
int ftx_add_layernorm_fwd(const float *x, const float *y, const float *y_bias, const float *gamma, const float *beta, float eps, int64_t rows, int32_t c, float *s_out, float *h_out, float *mean, float *rstd, void *stream);
/* backward: grad_x (rows, c) = grad_s (may be NULL) + d h / d s applied to grad_h -- the gradient of BOTH x and y;
 * grad_params (2 or 3, c) = d gamma, d beta and, with_y_bias != 0, d y_bias = the column sums of grad_x (float64 accumulation, fixed order). */
size_t ftx_layernorm_bwd_workspace_bytes(int64_t rows, int32_t c);
int ftx_add_layernorm_bwd(const float *grad_h, const float *grad_s, const float *s, const float *gamma, const float *mean, const float *rstd, int64_t rows, int32_t c, int32_t with_y_bias, float *grad_x, float *grad_params, void *workspace, size_t workspace_bytes, void *stream);

/* Column sums of a row-major (rows, cols) float32 matrix, float64 accumulation in a fixed order: out (cols).  The bias gradient of
 * the Linear layers (the reference leaves it to autograd's sum_to: models/transformers.py:16-45, spvcnn.py:164-180).  cols % 4 == 0. */
size_t ftx_colsum_workspace_bytes(int64_t rows, int32_t cols);
int ftx_colsum(const float *x, int64_t rows, int32_t cols, float *out, void *workspace, size_t workspace_bytes, void *stream);

/* ---- ViT self-attention (timm Attention.forward): models/transformers.py:36-37 ----
 * Fused softmax(Q K^T * scale) V on exact-fp32 MFMA; the (t, t) score matrix is never stored. */

/* qkv (b, t, 3, h, d) float32 exactly as the fused qkv Linear produces it; out (b, t, h*d);
 * lse (b, h, t) float32 = ln sum_k exp(scale * q.k), saved for the backward.  d must be 64. */
int ftx_attn_fwd(const float *qkv, int32_t b, int32_t t, int32_t h, int32_t d, float scale, float *out, float *lse, void *stream);
/* grad_qkv (b, t, 3, h, d) fully written.  workspace: ftx_attn_bwd_workspace_bytes(b, t, h). */
size_t ftx_attn_bwd_workspace_bytes(int32_t b, int32_t t, int32_t h);
int ftx_attn_bwd(const float *qkv, const float *out, const float *grad_out, const float *lse, int32_t b, int32_t t, int32_t h, int32_t d, float scale, float *grad_qkv, void *workspace, size_t workspace_bytes, void *stream);
/* The same two calls with an explicit tiling of the three attention kernels: qw waves of 32 queries (keys) per block x split key (query)
 * groups.  (0, 0) = chosen per launch from b*h*ceil(t/32) (what ftx_attn_fwd / ftx_attn_bwd do); built: (4,2) (2,2) (2,4) (1,2) (1,4) (1,8);
 * anything else is refused.  A per-call argument, not a process-wide switch.  A measurement / test aid: results of different tilings
 * differ in the last bits (the key range is summed in a different grouping), never with timing. */
int ftx_attn_fwd_tiled(const float *qkv, int32_t b, int32_t t, int32_t h, int32_t d, float scale, float *out, float *lse, int32_t qw, int32_t split, void *stream);
int ftx_attn_bwd_tiled(const float *qkv, const float *out, const float *grad_out, const float *lse, int32_t b, int32_t t, int32_t h, int32_t d, float scale, float *grad_qkv, void *workspace, size_t workspace_bytes, int32_t qw, int32_t split, void *stream);

/* ---- fused train-step losses + metric: modules/SemanticTrainer.py:158-194, models/metric.py:37-58 ----
 * losses[0] = loss_2d = CE_w(img_logit) + lambda * KL(softmax(lidar_logit) || softmax(img_logit2))
 * losses[1] = loss_3d = CE_w(lidar_logit) + lambda * KL(softmax(img_logit) || softmax(lidar_logit2))
 * (weighted-mean CE with class_weights (c) or NULL = ones; KL summed over classes, mean over points;
 * second heads NULL = single head, the KL terms then use the main heads, SemanticTrainer.py:164-165).
 * grad_* (n,c) receive d(loss_2d + loss_3d)/d(logits), fully written.  conf3d / conf2d (c,c) int64 are
 * ACCUMULATED: conf[label, argmax] += 1 for label != ignore_index (NULL = no metric).  c % 4 == 0, c <= 32. */
size_t ftx_fusion_loss_workspace_bytes(void);
/* The same with an explicit mix: loss = ce_scale * CE_w + lambda * KL.  ce_scale = 1 is the additive form of SemanticTrainer.py:158-178
 * (ftx_fusion_loss); ce_scale = 1 - lambda is the torchpack / DDP trainer's mix (modules/SemanticTorchpackTrainer.py:70-106). */
int ftx_fusion_loss_mix(const float *lidar_logit, const float *img_logit, const float *lidar_logit2, const float *img_logit2, const int64_t *label, const float *class_weights, float ce_scale, float lambda_xm, int64_t n, int32_t c, int32_t ignore_index, float *losses, float *grad_lidar, float *grad_img, float *grad_lidar2, float *grad_img2, int64_t *conf3d, int64_t *conf2d, void *workspace, size_t workspace_bytes, void *stream);
int ftx_fusion_loss(const float *lidar_logit, const float *img_logit, const float *lidar_logit2, const float *img_logit2, const int64_t *label, const float *class_weights, float lambda_xm, int64_t n, int32_t c, int32_t ignore_index, float *losses, float *grad_lidar, float *grad_img, float *grad_lidar2, float *grad_img2, int64_t *conf3d, int64_t *conf2d, void *workspace, size_t workspace_bytes, void *stream);

/* ---- evaluation scatter-back: data/utils/validate.py:62-120 + data/utils/evaluate.py:12-26 ----
 * For every ORIGINAL point i (m of them, all frames of the batch concatenated): r = inverse[i] is the row of the
 * model point (voxel) it was quantised into (inverse_map of its frame + the frame's row offset, map_sparse_to_org);
 * pred_3d = argmax(logits3d[r]), pred_2d = argmax(logits2d[r]), pred_ens = argmax(softmax(logits2d[r]) +
 * softmax(logits3d[r])) (first maximum wins); predictions are written as ORIGINAL label ids class_labels[pred]
 * (map_inverse_label).  gt (m) holds learning ids; as in Evaluator.update an original id 0 is replaced by
 * num_classes, and a point only counts if that id occurs in class_labels.  conf_* (c,c) int64 are ACCUMULATED:
 * conf[index of gt id][index of pred id] += 1.  Either logits pointer, any pred_* / conf_* pointer may be NULL.
 * *bad_flag is set to 1 if an inverse index or a gt id is out of range (those points are skipped). c <= 32. */
int ftx_eval_scatter_back(const float *logits3d, const float *logits2d, int64_t n_rows, int32_t num_classes, const int64_t *inverse, const int32_t *gt, int64_t m, const int32_t *class_labels, int32_t *pred3d, int32_t *pred2d, int32_t *pred_ens, int64_t *conf3d, int64_t *conf2d, int64_t *conf_ens, int32_t *bad_flag, void *stream);

/* ---- offline LiDAR -> image projection: data/semantic_kitti/preprocess.py:108-116 ----
 * points (n,3) float32 in the LiDAR frame, proj_matrix (3,4) float32 = P2 * Tr (preprocess.py:32-33).
 * keep[i] = 1 iff x > 0 and 0 < u < width and 0 < v < height (select_points_in_frustum, preprocess.py:75-91);
 * rowcol (n,2) = (v, u) for EVERY point (the caller compacts with keep), i.e. the reference's fliplr(img_points). */
int ftx_project_points(const float *points, int64_t n, const float *proj_matrix, int32_t width, int32_t height, uint8_t *keep, float *rowcol, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FTX_H */

/* lidal_amd.h -- C-ABI of liblidal_amd.so, the MI355X (gfx950) backend of the LiDAL sparse-voxel
 * hot path.
 *
 * Every entry point replaces one function of the pybind module `torchsparse.backend`
 * (torchsparse==1.4.0, pinned at /root/reference/docs/requirements.txt:191; not vendored in the
 * reference) or one block of reference Python that runs on the hot path.  The "replaces" lines
 * cite the reference call site (relative to /root/reference) that reaches it.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in `_host`;
 *   - the caller owns every buffer (inputs, outputs, workspace).  The library allocates device memory in exactly
 *     ONE place: at the first fused BatchNorm launch on a device it takes 2 MiB of device memory (64 publication
 *     buffers, two per stream: up to 32 streams per device) and 64 pinned host bytes (an error word), with one
 *     hipDeviceSynchronize, and keeps them for the life of the process -- see lidal_bn_set_fused below;
 *     LIDAL_BN_FUSED=0 in the environment (or lidal_bn_set_fused(0) before the first BatchNorm call) avoids it;
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*) and returns;
 *     counts that the host needs are written to device memory (`*_dev`) for the caller to read;
 *   - return value: 0 = OK, non-zero = error, message via lidal_last_error() (thread local);
 *   - no exceptions cross the boundary, no torch types in any signature;
 *   - dtype codes: 0 = float32, 1 = bfloat16 (features and weights; accumulation is always f32).
 */
#ifndef LIDAL_AMD_H
#define LIDAL_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIDAL_F32 0
#define LIDAL_BF16 1
/* f32 features and f32 results with the products on the bf16 matrix cores: every operand cut into three bf16 pieces
 * (exactly: 8 + 8 + 8 significand bits), six partial products, f32 accumulation -- the error of an f32 dot product whose
 * products were rounded once more (<= 3 * 2^-24 relative per product), at 2.7x the f32 MFMA rate.  Accepted by
 * lidal_conv_weight_image[_bytes|_tiling] (the image it builds is the only one the code reads) and lidal_conv_apply_image
 * [_ws]; the reduction dimension must be a multiple of 32.  csrc/conv_img.hip: conv_split_kernel. */
#define LIDAL_F32_SPLIT 2

const char* lidal_last_error(void);
int lidal_version(void);

/* ---- hashing ------------------------------------------------------------------------------ */
/* replaces backend.hash_cuda  (F.sphash(coords): network/utils.py:17,42-47,75).
 * coords i32 [n,4] = (x,y,z,batch) -> out i64 [n]: FNV-1a-64 over the four 32-bit words folded to
 * 60 bits. */
int lidal_hash(const int32_t* coords, int64_t n, int64_t* out, void* stream);
/* network/utils.py:44-47,72-75: float rows (x, y, z, b) [n,4] -> int32 rows (floor(x/s) s, floor(y/s) s,
 * floor(z/s) s, (int)b), the voxel a point falls into at tensor stride s, in one pass. */
int lidal_floor_coords(const float* coords, int64_t n, int stride, int32_t* out, void* stream);
/* network/utils.py:14-17 (initial_voxelize): float rows (x, y, z, b) [n,4] -> out_float = ((x * init_res) / after_res,
 * ..., b) and out_floor = (int)floor(out_float), both [n,4]; IEEE f32 multiply and divide, as torch computes
 * `(C[:, :3] * init_res) / after_res` -- one pass instead of mul, div, cat, floor, int. */
int lidal_revoxelize_coords(const float* coords, int64_t n, float init_res, float after_res, float* out_float,
                            int32_t* out_floor, void* stream);
/* replaces backend.kernel_hash_cuda  (F.sphash(coords, offsets): network/utils.py:70-74).
 * offsets i32 [k,3]; out i64 [k,n], hash of (xyz + offset_k, batch). */
int lidal_kernel_hash(const int32_t* coords, int64_t n, const int32_t* offsets, int k,
                      int64_t* out, void* stream);

/* replaces backend.hash_query_cuda  (F.sphashquery: network/utils.py:19,48,76).
 * Open-addressing table of 64-bit keys in HBM; value = index of the FIRST occurrence of the key
 * (the CPU dense_hash_map::insert semantics).  Keys must be < 2^63 (sphash output is 60 bit).
 * The buffer holds cap = 2^k >= 2 n slots {key u64 | value i32} and an occupancy bitmap of 8 bits per slot that the
 * kernel-map probes test before they touch a slot (most probed neighbours do not exist), a second, SPATIAL bitmap of the
 * same size (x-contiguous, direct mapped: the three x-neighbours of a voxel in one word) and a 64-byte header. */
int64_t lidal_hash_table_bytes(int64_t n_keys);
int lidal_hash_table_build(const int64_t* keys, int64_t n, void* table, int64_t table_bytes,
                           void* stream);
/* The table of lidal_hash(coords) built from the coordinates themselves (one launch; `stride` = the tensor stride of the
 * level, a power of two): same slots and values, and the spatial bitmap is filled -- the symmetric kernel-map probes
 * (lidal_kmap_build[_batch] with symmetric = 1) then answer the 13 probed offsets of a 3x3x3 map from 5 word reads.
 * A table built by lidal_hash_table_build (bare keys) marks its spatial bitmap invalid and is probed through the
 * hashed one. */
int lidal_hash_table_build_coords(const int32_t* coords, int64_t n, int stride, void* table, int64_t table_bytes,
                                  void* stream);
/* out[i] = position of q[i] in the keys the table was built from, or -1. */
int lidal_hash_table_query(const void* table, int64_t table_bytes, const int64_t* q, int64_t nq,
                           int64_t* out, void* stream);

/* ---- sorted unique / downsample ------------------------------------------------------------- */
/* replaces torch.unique(pc_hash) in network/utils.py:18 (sorted unique of i64 keys).
 * out [n] capacity; n_out_dev i64[1]. */
int64_t lidal_unique_workspace_bytes(int64_t n);
int lidal_unique_sorted_i64(const int64_t* keys, int64_t n, int64_t* out, int64_t* n_out_dev,
                            void* ws, int64_t ws_bytes, void* stream);
/* replaces F.spdownsample (torchsparse/nn/functional/downsample.py; reached from the four
 * stride-2 convs, network/spvcnn.py:28,34,40,46): xyz floored to multiples of `sample_stride`
 * (= conv stride x tensor stride), unique rows sorted by (batch,x,y,z).
 * Requires 0 <= x,y,z < 65536 and 0 <= batch < 32768.  out i32 [n,4] capacity. */
int64_t lidal_downsample_workspace_bytes(int64_t n);
int lidal_downsample(const int32_t* coords, int64_t n, int sx, int sy, int sz, int32_t* out,
                     int64_t* n_out_dev, void* ws, int64_t ws_bytes, void* stream);

/* Every coarser level of a stride-2 encoder at once (the four F.spdownsample calls of network/spvcnn.py:28,34,
 * 40,46 chained): level l = 1..levels is the sorted unique of floor(c / (2^l s)) * (2^l s) over the rows of
 * `coords` (tensor stride s) -- the same rows in the same (batch, x, y, z) order as l chained lidal_downsample
 * calls, from ONE sort and with ONE set of row counts to read back.  out i32 [levels * n, 4] capacity; level l
 * occupies rows [starts[l-1], starts[l]); starts_dev i64 [levels + 1].  0 <= x, y, z < 65536, 0 <= batch < 8192,
 * levels <= 4; a row outside these ranges is reported as starts[levels] = -1 (no other output is valid then). */
int64_t lidal_downsample_pyramid_workspace_bytes(int64_t n, int levels);
int lidal_downsample_pyramid(const int32_t* coords, int64_t n, int sx, int sy, int sz, int levels, int32_t* out,
                             int64_t* starts_dev, void* ws, int64_t ws_bytes, void* stream);

/* ---- input voxelisation ("next" row 8f-1) ---------------------------------------------------- */
/* replaces dataset/sk_dataset.py:143-171 for one scan: affine augmentation p*M (f64), feats =
 * (transformed metres, intensity), x`scale`, random translation into [0, full_scale)^3 from the
 * global min/max and the host's six uniform draws rnd[6] (:156), astype(int), and
 * np.unique(axis=0, return_index=True, return_inverse=True).
 *   points f32 [p,3], intensity f32 [p], m_dev f64 [9] (row-major trans_m), rnd_dev f64 [6]
 *   feats_p f32 [p,4] (per point; caller gathers rows unique_idx), coords_v i32 [p,3] capacity
 *   (unique voxels, lexicographic x,y,z), unique_idx i64 [p] capacity (first occurrence),
 *   inverse i64 [p], n_out_dev i64 [1], n_invalid_dev i32 [1] (points outside the grid: the
 *   reference asserts there are none, :161). */
int64_t lidal_voxelize_points_workspace_bytes(int64_t p);
int lidal_voxelize_points(const float* points, const float* intensity, int64_t p,
                          const double* m_dev, const double* rnd_dev, double scale, int full_scale,
                          float* feats_p, int32_t* coords_v, int64_t* unique_idx, int64_t* inverse,
                          int64_t* n_out_dev, int32_t* n_invalid_dev, void* ws, int64_t ws_bytes,
                          void* stream);

/* ---- kernel map (rule) building -------------------------------------------------------------- */
/* replaces the cache-miss branch of F.conv3d (torchsparse/nn/functional/conv.py): kernel_hash +
 * hash_query + nonzero.  `table` was built from sphash(in_coords).
 *   nbr_out  i32 [k, n_out]     input row feeding output row j through offset k, or -1
 *                               (in_coord = out_coord + offset_k)
 *   nbmaps   i32 [k*n_out, 2]   capacity; rows (in_idx, out_idx) grouped by k, ascending out_idx
 *                               -- bit-exact torchsparse order
 *   nbsizes  i32 [k],  koff i64 [k+1] exclusive prefix of nbsizes (koff[k] = total rules).
 * symmetric != 0 declares that out_coords are the coordinates the table was built from and that
 * offsets[k-1-j] == -offsets[j] (odd centred kernel at stride 1): half of the probes are then
 * replaced by mirroring (rule (i,j,k) <=> rule (j,i,k-1-j)); the result is identical.
 * mode 0 builds everything; mode 1 only nbr_out (all the forward / inference path reads; nbmaps,
 * nbsizes, koff may be NULL); mode 2 derives nbmaps / nbsizes / koff from an nbr_out built earlier
 * (the weight gradient reads them; table / out_coords / offsets are then ignored). */
int64_t lidal_kmap_workspace_bytes(int64_t n_out, int k);
int lidal_kmap_build(const void* table, int64_t table_bytes, const int32_t* out_coords,
                     int64_t n_out, const int32_t* offsets, int k, int symmetric, int32_t* nbr_out,
                     int32_t* nbmaps, int32_t* nbsizes, int64_t* koff, int mode, void* ws,
                     int64_t ws_bytes, void* stream);
/* From torchsparse-order rule lists (what backend.convolution_forward_cuda(in, out, W, nbmaps, nbsizes,
 * transposed) receives: nbmaps i32 [n_rules, 2] = (in, out) grouped by offset, nbsizes i32 [k], both on
 * the device; n_rules = capacity of nbmaps >= sum(nbsizes)) to the neighbour table lidal_kmap_order /
 * lidal_conv_apply_image consume: nbr_out i32 [k, n_out], -1 where an offset has no rule for a row.
 * n_bad_dev i32 [1] counts rules with an index out of range (0 = fine). */
int lidal_kmap_from_rules(const int32_t* nbmaps, const int32_t* nbsizes, int k, int64_t n_rules,
                          int64_t n_in, int64_t n_out, int32_t* nbr_out, int32_t* n_bad_dev, void* stream);
/* All kernel maps of a network in one chain of launches (n_jobs <= 12; HOST arrays of length n_jobs holding,
 * per map, the arguments of lidal_kmap_build; nbmaps[j] == NULL: neighbour table only): every stage (fill,
 * probe, count, scan, compact, sizes) is one launch over all maps.  Results per map are those of
 * lidal_kmap_build.  ws >= lidal_kmap_build_batch_workspace_bytes(n_out, k, n_jobs). */
int64_t lidal_kmap_build_batch_workspace_bytes(const int64_t* n_out, const int32_t* k, int n_jobs);
int lidal_kmap_build_batch(const void* const* tables, const int64_t* table_bytes,
                           const int32_t* const* out_coords, const int64_t* n_out,
                           const int32_t* const* offsets, const int32_t* k, const int32_t* symmetric,
                           int32_t* const* nbr_out, int32_t* const* nbmaps, int32_t* const* nbsizes,
                           int64_t* const* koff, int n_jobs, void* ws, int64_t ws_bytes, void* stream);
/* nbr_in i32 [k, n_in]: output row fed by input row i through offset k, or -1 (inverse table, used
 * by data-gradient and transposed convolution). */
int lidal_kmap_invert(const int32_t* nbr_out, int64_t n_out, int k, int32_t* nbr_in, int64_t n_in,
                      void* stream);

/* Row order for lidal_conv_apply: rows of nbr [k, n_rows] sorted by their occupancy pattern (bit j set
 * iff nbr[j][row] >= 0; stable, so ties keep row order).  perm i32 [n_rows] (sorted position ->
 * row), nbr_perm i32 [k, n_rows] = nbr[:, perm], tile_masks u32 [ceil(n_rows/128)] = OR of the row
 * patterns of each 128-row tile (may be NULL).  With them a tile only touches the offsets of its
 * own patterns. */
/* the library's own stable LSD radix sort (csrc/sort.hip: Onesweep, 8 bits per pass, 1 + ceil(bits/8)
 * launches) of (u32 or u64 key, i32 value) pairs by the low `bits` key bits -- what every sorted order
 * of this library comes from (row orders, torch.unique of hashes / packed coordinates, contributor
 * lists, the scorer's cell keys); exported for tests.  keys_in / vals_in are not written;
 * ws >= lidal_sort_pairs_workspace_bytes(n). */
int64_t lidal_sort_pairs_workspace_bytes(int64_t n);
int lidal_sort_pairs(const uint32_t* keys_in, const int32_t* vals_in, uint32_t* keys_out, int32_t* vals_out,
                     int64_t n, int bits, void* ws, int64_t ws_bytes, void* stream);
int lidal_sort_pairs_u64(const uint64_t* keys_in, const int32_t* vals_in, uint64_t* keys_out, int32_t* vals_out,
                         int64_t n, int bits, void* ws, int64_t ws_bytes, void* stream);
int64_t lidal_kmap_order_workspace_bytes(int64_t n_rows);
/* The same for n_jobs (<= 16) tables of ONE kernel volume k in one go (host arrays of device pointers /
 * row counts; ws >= lidal_kmap_order_workspace_bytes(sum of the row counts)): one key array, one sort.
 * Results are identical to n_jobs calls of lidal_kmap_order. */
int lidal_kmap_order_batch(const int32_t* const* nbr, const int64_t* n_rows, int n_jobs, int k,
                           int32_t* const* perm, int32_t* const* nbr_perm, uint32_t* const* tile_masks,
                           void* ws, int64_t ws_bytes, void* stream);
int lidal_kmap_order(const int32_t* nbr, int64_t n_rows, int k, int32_t* perm, int32_t* nbr_perm,
                     uint32_t* tile_masks, void* ws, int64_t ws_bytes, void* stream);

/* ---- point <-> voxel ------------------------------------------------------------------------- */
/* replaces backend.count_cuda (F.spcount: network/utils.py:20,49). out i32 [m] (zeroed here). */
int lidal_count(const int32_t* idx, int64_t n, int32_t* out, int64_t m, void* stream);
/* replaces backend.voxelize_forward_cuda / voxelize_backward_cuda (F.spvoxelize:
 * network/utils.py:22,25,56).  out[idx[i]] += feat[i] / counts[idx[i]]; f32 only. */
int lidal_voxelize_fwd(const float* feat, const int32_t* idx, const int32_t* counts, float* out,
                       int64_t n, int64_t m, int c, void* stream);
/* The same when every voxel holds exactly one point (idx a permutation of 0..n-1; LiDAL's scans arrive
 * voxelised, network/spvcnn.py:114): out[idx[i]] = feat[i], bit-equal to the mean form, f32 or bf16 rows. */
int lidal_voxelize_fwd_1to1(const void* feat, const int32_t* idx, void* out, int64_t n, int c, int dtype,
                            void* stream);
/* backward: gin[i] = gout[idx[i]] / counts[idx[i]] (+ residual[i], same dtype [n, c], may be NULL:
 * the gradient reaching the same point rows through a second consumer of the features). */
int lidal_voxelize_bwd(const void* gout, const int32_t* idx, const int32_t* counts,
                       const void* residual, void* gin, int64_t n, int64_t m, int c, int dtype,
                       void* stream);
/* replaces backend.devoxelize_forward_cuda / devoxelize_backward_cuda (F.spdevoxelize:
 * network/utils.py:83,95).  idx i32 [n,8], w f32 [n,8]; out[i] = sum_k w[i,k] feat[idx[i,k]]. */
int lidal_devoxelize_fwd(const void* feat, const int32_t* idx, const float* w, void* out,
                         int64_t n, int64_t m, int c, int dtype, void* stream);
int lidal_devoxelize_bwd(const float* gout, const int32_t* idx, const float* w, float* gin,
                         int64_t n, int64_t m, int c, void* stream);
/* (n_entries = length of `order`; lists averaging >= 32 contributors per voxel are split over
 * several waves / workgroups per voxel, ws from lidal_segment_workspace_bytes, may be NULL if 0) */
/* Atomic-free, bitwise reproducible forms of the two scatter sums above.  A point->voxel index
 * idx i32 [n_entries] (n_entries = n for voxelize, 8n for devoxelize: entry = point*8 + corner) is
 * transposed once into per-voxel contributor lists: order i32 [n_entries] (entries sorted by voxel,
 * ascending inside a voxel; entries with idx < 0 or weight 0 at the end), seg_ptr i64 [m+1]. */
int64_t lidal_invlist_workspace_bytes(int64_t n_entries);
int lidal_invlist_build(const int32_t* idx, const float* w, int64_t n_entries, int64_t m,
                        int32_t* order, int64_t* seg_ptr, void* ws, int64_t ws_bytes, void* stream);
int64_t lidal_segment_workspace_bytes(int64_t n_entries, int64_t m, int c);
int lidal_voxelize_fwd_sorted(const void* feat, const int32_t* order, const int64_t* seg_ptr,
                              const int32_t* counts, void* out, int64_t m, int c, int dtype,
                              int64_t n_entries, void* ws, int64_t ws_bytes, void* stream);
int lidal_devoxelize_bwd_sorted(const void* gout, const int32_t* order, const int64_t* seg_ptr,
                                const float* w, void* gin, int64_t m, int c, int dtype,
                                int64_t n_entries, void* ws, int64_t ws_bytes, void* stream);
/* The same gradient through the CELLS, for levels where a voxel has many contributors (stride 16): every point of a cell --
 * the voxel its own coordinates floor to, idx8[:, 0] -- interpolates from the same eight corners (network/utils.py:67-79),
 * so  cs[cell][j] = sum over the cell's points of w8[p][j] * gout[p]  (vorder / vseg: the points of every cell, the inverse
 * lists of idx8[:, 0])  and  gin[v] = sum of the cs[cell][j] whose corner j is v  (corder / cseg: the inverse lists of the
 * cells' [m, 8] corner indices, entries cell * 8 + j).  Reads every gradient row once instead of eight times.  Fixed
 * orders, no atomics; differs from lidal_devoxelize_bwd_sorted by the association of the f32 sums.  c in {32, 64, 128, 256}
 * (any power of two of 16-byte steps, 4 .. 64); ws >= lidal_devoxelize_bwd_cells_workspace_bytes(m, c) = m * 8 * c * 4. */
int64_t lidal_devoxelize_bwd_cells_workspace_bytes(int64_t m, int c);
int lidal_devoxelize_bwd_cells(const void* gout, const int32_t* vorder, const int64_t* vseg, const float* w8,
                               const int32_t* corder, const int64_t* cseg, void* gin, int64_t m, int c, int dtype,
                               void* ws, int64_t ws_bytes, void* stream);
/* replaces F.calc_ti_weights (torchsparse/nn/functional/devoxelize.py; network/utils.py:77):
 * coords f32 [n, cstride>=3], idx i64 [8,n] -> w f32 [n,8] and idx32 i32 [n,8] (both already
 * transposed as network/utils.py:78-79 does). */
int lidal_ti_weights(const float* coords, int cstride, const int64_t* idx, int64_t n, float scale,
                     float* w, int32_t* idx32, void* stream);

/* ---- sparse convolution ---------------------------------------------------------------------- */
/* replaces backend.convolution_forward_cuda and the data-gradient half of
 * convolution_backward_cuda (every spnn.Conv3d.forward/backward, 49 per model pass).
 * Output-stationary fused gather-GEMM with register accumulators:
 *     out[row(j), :] = sum_k  in[ nbr[kk][j], : ] * W[k],   kk = kflip ? K-1-k : k
 * with nbr i32 [k, n_out] (-1 = no rule) and row(j) = perm ? perm[j] : j  (pass lidal_kmap_order's perm together
 * with its permuted table and its tile_masks, which spare the kernel the offsets a tile has no rule for).
 * n_in = rows of `in` (every nbr entry is < n_in; in and the weights are addressed with 32-bit byte offsets, so
 * each must stay below 2 GiB).  No atomics: each output row is written exactly once => bitwise reproducible.
 * Optional epilogue (inference: the eval-mode spnn.BatchNorm and ReLU that follow the conv in
 * network/utils.py:110-114,147-155): ep_scale / ep_shift f32 [co] (both or neither; see
 * lidal_bn_fold) give out = act(acc * scale + shift), act = ReLU iff (ep_relu & 1); ep_residual
 * (NULL or a [n_out, co] matrix of the output's dtype) is added to that, row for row (the
 * point-branch sum network/spvcnn.py:136,143,151; the shortcut of a residual block,
 * network/utils.py:171, whose ReLU comes AFTER the sum: ep_relu & 2).  With k = 1 and nbr = NULL
 * (the identity rule list) the same kernel is the dense per-row product of the 1x1x1 convolutions
 * and nn.Linear layers.
 * (Rounds 1-3 also exported a first generation, lidal_conv_apply + lidal_conv_weight_pack, with plain [k][co][ci]
 * weights staged through registers; it left the library in round 4 and lives on as a test-only object,
 * tests/native/conv_gen1.hip, that the kernels below are compared with bit for bit.)
 * The weights are supplied as LDS IMAGES (csrc/conv_img.hip) -- every slab (offset, column block, reduction slice)
 * laid out in global memory exactly as the MFMA B-fragment reads want it in LDS, so the kernel stages a slab by
 * LDS-DMA (no vector registers, no ds_write) and reads it bank-conflict free.
 *   lidal_conv_weight_image_bytes  size of the image of a [k][n_red][n_col] weight for a
 *                                  convolution producing n_out rows (the tiling depends on it)
 *   lidal_conv_weight_image        role 0: w is [k][n_red][n_col] (forward: n_red = ci, n_col = co)
 *                                  role 1: w is [k][n_col][n_red] (data gradient of the same
 *                                  parameter: n_red = co, n_col = ci); casts to `dtype`
 *   lidal_conv_apply_image         the contraction above; wimg = the image built for the
 *                                  SAME (ci = n_red, co = n_col, k, dtype, n_out); tile_masks are
 *                                  required with a table (nbr == NULL: identity rule list, k = 1);
 *                                  ci must be a multiple of 4 (f32) / 8 (bf16).
 * tile_stats (NULL or f32 [co][ceil(n_out / 128)][3] -- column-major over the tiles, so that the merge of a channel
 * reads one contiguous run): per output column and 128-row tile of the kernel's row order, (count, mean, M2) of the values as stored -- the batch statistics of a train-mode
 * BatchNorm that follows (network/utils.py:115), taken in the epilogue instead of by a pass over the
 * stored matrix; merged by lidal_bn_train_fwd_tiles. */
/* Rows per tile of the per-tile BatchNorm statistics lidal_conv_apply_image can leave (tile_stats is
 * f32 [ceil(n_out / rows), co, 3]): size the buffer from this, not from a constant. */
int lidal_conv_stats_tile_rows(void);
int64_t lidal_conv_weight_image_bytes(int k, int ci, int co, int dtype, int64_t n_out);
/* identifies the tiling an image is built for (two n_out with the same value share images: a
 * host-side cache key for weights that do not change between calls, i.e. inference) */
int lidal_conv_weight_image_tiling(int ci, int co, int dtype, int64_t n_out);
int lidal_conv_weight_image(const void* w, int w_dtype, int role, void* img, int dtype, int k,
                            int n_red, int n_col, int64_t n_out, void* stream);
/* both images of one [k][ci][co] parameter in ONE launch: img_fwd (role 0, for a convolution
 * producing n_out_fwd rows) and img_bwd (role 1, for its data gradient producing n_out_bwd rows) */
int lidal_conv_weight_image_pair(const void* w, int w_dtype, void* img_fwd, int64_t n_out_fwd,
                                 void* img_bwd, int64_t n_out_bwd, int dtype, int k, int ci, int co,
                                 void* stream);
/* the image pairs of MANY parameters in one launch (training: every weight changes every step).
 * lidal_conv_weight_image_job fills one host record of lidal_conv_weight_image_job_bytes() bytes
 * (role 0: w is [k][ci][co], role 1: [k][co][ci] = nn.Linear's layout; `first` = segments of the
 * records before it; returns this record's segment count, < 0 on error);
 * lidal_conv_weight_image_batch takes the records as a DEVICE array. */
int lidal_conv_weight_image_job_bytes(void);
int64_t lidal_conv_weight_image_job(void* job, const void* w, int role, void* img_fwd,
                                    int64_t n_out_fwd, void* img_bwd, int64_t n_out_bwd, int dtype,
                                    int k, int ci, int co, int64_t first);
int lidal_conv_weight_image_batch(const void* jobs, int n_jobs, int64_t total_segments, int w_dtype,
                                  int dtype, void* stream);
int lidal_conv_apply_image(const void* in, const void* wimg, const int32_t* nbr, const int32_t* perm,
                           const uint32_t* tile_masks, void* out, int64_t n_in, int64_t n_out,
                           int ci, int co, int k, int kflip, int dtype, const float* ep_scale,
                           const float* ep_shift, int ep_relu, const void* ep_residual,
                           float* tile_stats, void* stream);
/* lidal_conv_apply_image with a workspace (ws >= lidal_conv_apply_workspace_bytes(n_out, co), may be NULL / 0): on the
 * coarse levels -- few 128-row tiles, each a long chain of (offset, reduction slice) phases while most of the chip
 * idles -- the launch then SPLITS every tile's active offsets over 2-4 workgroups, which leave f32 partial tiles in
 * ws; a second kernel adds them in a fixed order and runs the epilogue (permutation, affine map / ReLU / residual,
 * BatchNorm tile statistics).  bf16 only.  Which launches split is a function of (n_out, ci, co, k, dtype) alone; results differ
 * from the unsplit kernel's only by the association of the f32 sum over the offsets (~1e-7 relative before the
 * rounding to the output dtype). */
int64_t lidal_conv_apply_workspace_bytes(int64_t n_out, int co);
int lidal_conv_apply_image_ws(const void* in, const void* wimg, const int32_t* nbr, const int32_t* perm,
                              const uint32_t* tile_masks, void* out, int64_t n_in, int64_t n_out,
                              int ci, int co, int k, int kflip, int dtype, const float* ep_scale,
                              const float* ep_shift, int ep_relu, const void* ep_residual,
                              float* tile_stats, void* ws, int64_t ws_bytes, void* stream);
/* The data gradient of a convolution whose INPUT was y = act(bn(x)) (torchsparse: convolution_backward_cuda's
 * grad_input half, followed by the BatchNorm backward of network/utils.py:115): lidal_conv_apply_image on
 * (gout, data-gradient image) -> gin, and in the same launch the backward sums of that BatchNorm per 128-row
 * tile of gin's rows: bn_sums f32 [c_gin, ceil(n_gin / tile rows), 2] = (sum dy', sum dy' xhat), dy' = gin where
 * the fused ReLU (bn_relu) let the value through, xhat = (bn_x - mean) * invstd; bn_x [n_gin, c_gin] in `dtype`.
 * Feed bn_sums to lidal_bn_bwd_tiles.  c_gin must be whole 16-byte vectors. */
int lidal_conv_dgrad_bn_sums(const void* gout, const void* wimg, const int32_t* nbr, const int32_t* perm,
                             const uint32_t* tile_masks, void* gin, int64_t n_gout, int64_t n_gin, int c_gout,
                             int c_gin, int k, int kflip, int dtype, const void* bn_x, const float* bn_mean,
                             const float* bn_invstd, const float* bn_gamma, const float* bn_beta, int bn_relu,
                             float* bn_sums, void* stream);
/* (the same with the workspace of lidal_conv_apply_image_ws) */
int lidal_conv_dgrad_bn_sums_ws(const void* gout, const void* wimg, const int32_t* nbr, const int32_t* perm,
                                const uint32_t* tile_masks, void* gin, int64_t n_gout, int64_t n_gin, int c_gout,
                                int c_gin, int k, int kflip, int dtype, const void* bn_x, const float* bn_mean,
                                const float* bn_invstd, const float* bn_gamma, const float* bn_beta, int bn_relu,
                                float* bn_sums, void* ws, int64_t ws_bytes, void* stream);
/* replaces the weight-gradient half of backend.convolution_backward_cuda:
 *     gw[k] = a[ pairs[:, a_col] ]^T  *  b[ pairs[:, 1 - a_col] ]      (f32 [k][ca][cb])
 * pairs = nbmaps i32 [M,2], koff i64 [k+1] (device); n_a, n_b = rows of a and b.
 * pairs == NULL means the identity rule list (row p with row p): the weight gradient of a dense
 * [n, ca]^T x [n, cb] product (1x1x1 convolutions and the point-branch Linear layers), where a
 * library GEMM would run its whole n-long reduction in a handful of workgroups.
 * The reduction over the rules is split between workgroups; each leaves an f32 [ca][cb] slab in
 * `partial` (n_slabs slabs, at least lidal_conv_wgrad_slabs(...) of them) and a second kernel adds
 * the slabs of every offset in a fixed order => bitwise reproducible, no atomics.  bf16 with
 * ca, cb multiples of 8: W equally long runs of 64-rule stages, one per resident workgroup,
 * gathered by LDS-DMA (csrc/wgrad_dma.hip); otherwise split-K slabs per offset (csrc/conv.hip).
 * dtype LIDAL_F32_SPLIT (round 6): a and b are F32; every product runs as six bf16 MFMAs on the operands' three exact bf16
 * pieces (as LIDAL_F32_SPLIT of lidal_conv_apply_image).  The pieces (bf16 [n_a, 3 ca] and [n_b, 3 cb]) are cut by the
 * call itself into the tail of `partial`: lidal_conv_wgrad_slabs(.., LIDAL_F32_SPLIT) counts the room for them in slabs
 * (-1: the shape is not served -- ca, cb multiples of 8, split operands below 4 GiB). */
int64_t lidal_conv_wgrad_slabs(int64_t n_a, int64_t n_b, int k, int ca, int cb, int dtype);
int lidal_conv_wgrad(const void* a, const void* b, int64_t n_a, int64_t n_b, const int32_t* pairs,
                     const int64_t* koff, int a_col, float* gw, float* partial, int64_t n_slabs,
                     int k, int ca, int cb, int dtype, void* stream);

/* The weight gradient on rule lists re-ordered into ONE STREAM PER WORKGROUP (round 6; csrc/wgrad_streams.hip,
 * csrc/wgrad_dma.hip wgrad_stream_kernel).  lidal_conv_wgrad reads both rows of every rule from the fabric: the ~5 rules
 * that touch a row lie in different offsets, worked on far apart in time (96 -> 96 on 397 k rows: 851 MB per launch for
 * 152 MB of operands).  lidal_wgrad_streams_build re-orders the rules of a stride-1 map (pairs / koff of
 * lidal_kmap_build, k <= 32 offsets, n_rows rows on both sides) so that the rules of a row meet in one XCD's L2: rows are
 * cut into blocks of consecutive spatial keys -- key_tab == NULL: the row index (rows numbered in coordinate order:
 * every level lidal_downsample made), key_range = n_rows; otherwise key(row) = max over key_tab i32 [key_k, n_rows] in
 * [0, key_range): the strided map's inverse neighbour table (lidal_kmap_invert) names a row's parent on the next level,
 * for levels numbered otherwise (SPVCNN's level 0: by coordinate hash) -- block b belongs to the workgroups w with
 * w % 8 == b % 8 (one XCD under round-robin dispatch), which share the offsets in proportion to their rule counts and
 * walk the blocks in order.  Outputs: spairs i32 [spairs_rules >= lidal_wgrad_streams_rules(..), 2] (bit 31 of a
 * rule's first index: the accumulator set; out-of-range rules as padding) and sdesc i32
 * [lidal_wgrad_streams_desc_words(k, n_wg)] = {n_wg, k, stages, 0, soff[n_wg + 1], offsets of workgroup w [n_wg][2],
 * the reducer's table [k][8][3]}.  n_wg = lidal_wgrad_streams_workgroups() (512).  Nothing is read back to the host;
 * the result is a pure function of (pairs, koff, keys): integer arithmetic, stable sorts. */
int lidal_wgrad_streams_workgroups(void);
int64_t lidal_wgrad_streams_rules(int64_t n_rows, int k, int n_wg);
int64_t lidal_wgrad_streams_desc_words(int k, int n_wg);
int64_t lidal_wgrad_streams_workspace_bytes(int64_t n_rows, int k);
int lidal_wgrad_streams_build(const int32_t* pairs, const int64_t* koff, int k, int64_t n_rows, const int32_t* key_tab,
                              int key_k, int64_t key_range, int n_wg, int32_t* spairs, int64_t spairs_rules,
                              int32_t* sdesc, void* ws, int64_t ws_bytes, void* stream);
/* gw[k] = a[spairs[:, a_col]]^T b[spairs[:, 1 - a_col]] over those streams: the same rules as lidal_conv_wgrad in
 * another order -- equal to it within f32 rounding, bitwise reproducible (fixed slab order, no atomics).  bf16, ca and cb
 * multiples of 8 within ONE channel tile (<= 128); partial >= 2 n_wg slabs of [ca][cb] f32. */
int lidal_conv_wgrad_streams_serves(int64_t n_a, int64_t n_b, int k, int ca, int cb);     /* 1: the shape is served */
int lidal_conv_wgrad_streams(const void* a, const void* b, int64_t n_a, int64_t n_b, const int32_t* spairs,
                             const int32_t* sdesc, int n_wg, int a_col, float* gw, float* partial, int64_t n_slabs,
                             int k, int ca, int cb, int dtype, void* stream);

/* ---- batch normalisation over rows ------------------------------------------------------------ */
/* replaces the torch.nn.BatchNorm1d kernels behind spnn.BatchNorm (network/utils.py:115; 49 per
 * model) for x [n, c] row-major, dtype f32 or bf16 (statistics always f32).  Training forward:
 * biased batch variance for normalisation, running stats updated with `momentum` and the unbiased
 * variance (running_* may be NULL); save_mean / save_invstd f32 [c] feed the backward.
 * c must be a multiple of 4 (f32) / 8 (bf16). */
int64_t lidal_bn_workspace_bytes(int64_t n, int c);
/* `relu` != 0 fuses the ReLU that follows the normalisation in the model (forward: max(y, 0);
 * backward: dy is taken where y > 0, y recomputed from x). */
/* num_batches_tracked (nn.BatchNorm1d's i64 scalar buffer, may be NULL) is incremented by one. */
/* relu: bit 1 = ReLU on the normalised value, bit 2 (with a residual) = ReLU after the residual sum
 * (the end of a residual block, network/utils.py:171; the backward of that form takes dy already
 * masked by y > 0, e.g. from lidal_add_relu_bwd, and relu = 0).
 * residual (same dtype [n, c], may be NULL): y = act(bn(x)) + residual -- the point-branch sum of
 * network/spvcnn.py:104,111,118 (`z1.F = z1.F + point_transforms(z.F)`) inside the normalising pass;
 * its gradient is grad_y unchanged. */
int lidal_bn_train_fwd(const void* x, int dtype, int64_t n, int c, const float* gamma,
                       const float* beta, float eps, float momentum, float* running_mean,
                       float* running_var, int64_t* num_batches_tracked, int relu,
                       const void* residual, void* y, float* save_mean, float* save_invstd, void* ws,
                       int64_t ws_bytes, void* stream);
/* lidal_bn_train_fwd with the statistics pass replaced by the per-tile (count, mean, M2) triples
 * lidal_conv_apply_image wrote (tile_stats f32 [c][n_tiles][3]). */
int lidal_bn_train_fwd_tiles(const void* x, int dtype, int64_t n, int c, const float* gamma,
                             const float* beta, float eps, float momentum, float* running_mean,
                             float* running_var, int64_t* num_batches_tracked, int relu,
                             const void* residual, void* y, float* save_mean, float* save_invstd,
                             const float* tile_stats, int64_t n_tiles, void* stream);
int lidal_bn_eval_fwd(const void* x, int dtype, int64_t n, int c, const float* gamma,
                      const float* beta, const float* running_mean, const float* running_var,
                      float eps, int relu, void* y, void* stream);
/* dx may be NULL (only parameter gradients wanted).  dy_stride = elements between the rows of dy
 * (>= c, a multiple of the 16-byte vector): the gradient of a channel slice of a concatenation
 * (network/utils.py:cat in the up stages) is read in place, without a contiguous copy. */
int lidal_bn_bwd(const void* x, const void* dy, int64_t dy_stride, int dtype, int64_t n, int c, const float* gamma,
                 const float* beta, int relu, const float* save_mean, const float* save_invstd,
                 void* dx, float* grad_gamma, float* grad_beta, void* ws, int64_t ws_bytes,
                 void* stream);
/* Column sums of x [n, c] -> out f32 [c] (bias gradients of the 1x1 / Linear layers);
 * ws >= lidal_bn_workspace_bytes(n, c) + 12*c bytes. */
/* BatchNorm backward whose per-column sums came with dy: tile_sums f32 [c, n_tiles, 2] = (sum dy', sum dy' xhat)
 * per 128-row tile (lidal_conv_stats_tile_rows), left by lidal_conv_dgrad_bn_sums.  Merges the tiles in f64
 * (-> grad_beta, grad_gamma) and writes dx; no pass over (x, dy) for the sums. */
int lidal_bn_bwd_tiles(const void* x, const void* dy, int64_t dy_stride, int dtype, int64_t n, int c,
                       const float* gamma, const float* beta, int relu, const float* save_mean,
                       const float* save_invstd, void* dx, float* grad_gamma, float* grad_beta,
                       const float* tile_sums, int64_t n_tiles, void* stream);
/* The tail of a residual block backwards (network/utils.py:142-172: out = relu(bn2(x_a) + shortcut), the shortcut the
 * identity or bn_s(x_b)): gm = g where out > 0 (lidal_add_relu_bwd) and, in the same pass, the partial sums of the
 * BatchNorm backward of bn2 over (x_a, gm) -- and of the shortcut's BatchNorm over (x_b, gm) when x_b != NULL -- that
 * lidal_bn_bwd would take in its first pass over the same operands (bit for bit).  part_a / part_b: >=
 * lidal_bn_workspace_bytes(n, c) each (part_bytes), handed to lidal_bn_bwd_from_sums; relu = 0 there (the ReLU of the
 * tail follows the sum).  Rows are contiguous ([n, c], c whole 16-byte vectors). */
int lidal_add_relu_bwd_bn_sums(const void* out, const void* g, void* gm, int dtype, int64_t n, int c,
                               const void* x_a, const float* mean_a, const float* invstd_a, void* part_a,
                               const void* x_b, const float* mean_b, const float* invstd_b, void* part_b,
                               int64_t part_bytes, void* stream);
/* The same tail for the levels with many rows: gm = g where out > 0, and per part (a slab of rows; there are
 * lidal_bn_tail_parts(n, c, dtype) of them) and channel the f32 pairs (sum gm, sum gm * xhat) of the BatchNorm over x_a --
 * and over x_b, if given -- as [channel][part][2] (sums_a / sums_b: c * n_parts * 2 floats each): the tile sums
 * lidal_bn_bwd_tiles takes, with n_tiles = n_parts and relu = 0.  One element-wise pass instead of lidal_add_relu_bwd and
 * the first pass of one or two lidal_bn_bwd. */
int64_t lidal_bn_tail_parts(int64_t n, int c, int dtype);
int lidal_add_relu_bwd_bn_tile_sums(const void* out, const void* g, void* gm, int dtype, int64_t n, int c,
                                    const void* x_a, const float* mean_a, const float* invstd_a, float* sums_a,
                                    const void* x_b, const float* mean_b, const float* invstd_b, float* sums_b,
                                    int64_t n_parts, void* stream);
/* lidal_bn_bwd without its first pass: the partial sums are in `part` (lidal_add_relu_bwd_bn_sums). */
int lidal_bn_bwd_from_sums(const void* x, const void* dy, int64_t dy_stride, int dtype, int64_t n, int c,
                           const float* gamma, const float* beta, int relu, const float* save_mean,
                           const float* save_invstd, void* dx, float* grad_gamma, float* grad_beta,
                           const void* part, int64_t part_bytes, void* stream);
/* The merge steps of a BatchNorm layer (tile statistics -> mean / invstd; backward partial sums -> parameter
 * gradients) run INSIDE the launch that consumes them (the first workgroups merge and publish, all workgroups fetch;
 * csrc/bn.hip) -- same arithmetic, same results bit for bit as the separate merge launches, which on = 0 brings back
 * (also: environment LIDAL_BN_FUSED=0).
 *   Memory: the publication buffers are the library's own (the one allocation named under "Conventions"): two per
 *   STREAM, so a buffer is re-used only by a later launch of the same stream, i.e. after its last reader has finished
 *   -- any number of host threads and up to 32 streams per device may run BatchNorm launches at the same time; a 33rd
 *   stream silently takes the separate merge launches (same results).
 *   Forward progress: the waiting workgroups rely on the merging ones -- the LOWEST workgroup ids of the launch --
 *   having been dispatched before them (in-order workgroup dispatch, the assumption sort.hip's decoupled look-back
 *   makes too).  A workgroup that waits longer than ~30 s gives up: it writes NaN to its channel AND raises a
 *   per-device error word, and every later BatchNorm entry point and lidal_plan_run on that device fails with a message
 *   naming the launch (lidal_bn_check_device) until lidal_bn_set_fused() is called again: never a silent NaN.
 * lidal_bn_set_fused also acknowledges such an error. */
int lidal_bn_set_fused(int on);
/* bf16 lidal_bn_bwd takes its two sums per channel as f32 sums over slabs of rows (512 slabs on the large levels, merged in
 * f64 like the convolutions' tile sums: an element-wise-shaped first pass at ~5 TB/s) -- 1, the default -- or as the f64
 * partial sums the f32 mode uses (0; LIDAL_BN_SLAB_SUMS=0 in the environment).  Returns the previous setting. */
int lidal_bn_set_slab_sums(int on);
/* 0, or 1 with lidal_last_error() set if a fused BatchNorm launch on the current device timed out (above). */
int lidal_bn_check_device(void);
/* eval-mode BatchNorm as a per-channel affine map (scale = gamma / sqrt(var + eps),
 * shift = beta - mean * scale), the operands of lidal_conv_apply's epilogue. */
int lidal_bn_fold(const float* gamma, const float* beta, const float* running_mean,
                  const float* running_var, float eps, int c, float* scale, float* shift,
                  void* stream);
int lidal_colsum(const void* x, int dtype, int64_t n, int c, float* out, void* ws,
                 int64_t ws_bytes, void* stream);

/* ---- small fused row-wise ops of the training step --------------------------------------------- */
/* relu(a + b) of the residual blocks (network/utils.py:171), forward and backward
 * (gin = g where y > 0; both summands receive it).  numel a multiple of 4 (f32) / 8 (bf16). */
int lidal_add_relu_fwd(const void* a, const void* b, void* y, int64_t numel, int dtype,
                       void* stream);
int lidal_add_relu_bwd(const void* y, const void* g, void* gin, int64_t numel, int dtype,
                       void* stream);
/* replaces torch.nn.functional.cross_entropy(logits, labels, ignore_index, reduction='mean')
 * (train.py:136): out2 f32 [2] = {mean loss over non-ignored rows, number of such rows};
 * backward: dlogits = (softmax - onehot) * grad_scale[0] / out2[1], zero on ignored rows. */
int64_t lidal_ce_workspace_bytes(int64_t n);
int lidal_ce_fwd(const void* logits, int dtype, const int64_t* labels, int64_t n, int c,
                 int64_t ignore_index, float* out2, void* ws, int64_t ws_bytes, void* stream);
int lidal_ce_bwd(const void* logits, int dtype, const int64_t* labels, int64_t n, int c,
                 int64_t ignore_index, const float* fwd_out2, const float* grad_scale,
                 void* dlogits, void* stream);

/* ---- probability inference post-processing ------------------------------------------------- */
/* replaces score/prob_inference.py:100-113: logits f32 [nv, c] of `reps` collated views,
 * inverse i64 [reps*p] (voxel row of every point in every view) -> prob f32 [p,c] = mean over
 * views of softmax(logits[inverse]), pred i64 [p] = argmax. */
int lidal_view_mean_softmax(const float* logits, const int64_t* inverse, int reps, int64_t p,
                            int c, float* prob, int64_t* pred, void* stream);

/* replaces evaluate.py:100-109 + utils/iou_sk.py:14-19 ("next" row 8f-4): conf i32 [c*c] +=
 * bincount(argmax(logits[inverse]) * c + labels) over labelled points (label < 100); conf is
 * accumulated across calls (the caller zeroes it once and all-reduces it across ranks). */
int lidal_confusion_accumulate(const float* logits, const int64_t* inverse, const int64_t* labels,
                               int64_t p, int c, int32_t* conf, void* stream);

/* ---- inter-frame divergence / entropy scoring ---------------------------------------------- */
/* replaces dataset/prepare_kdtree_sk.py:76-80 ("next" row 8f-2): sensor-frame points f32 [p,3] ->
 * world-frame f64 [p,3] = (hcoords * pose^T)[:, :3] with pose f64 [16] row-major 4x4 (device),
 * same operation order as the reference's broadcast-multiply + sum. */
int lidal_register_points(const float* points, int64_t p, const double* pose_dev, double* world,
                          void* stream);
/* Uniform-grid nearest-neighbour structure over one frame's world-frame points (replaces the
 * pickled sklearn KDTree of dataset/prepare_kdtree_sk.py:83 as used by
 * score/sv_level/LiDAL.py:52-66).  cell: any size > 0 (kept in the grid; a query visits the cells that meet the cube of
 * its match radius: 27 for cell = radius, at most 8 for cell = 2 x radius -- the answers are the same).
 *   pts f64 [p,3];  grid bytes from lidal_nn_grid_bytes(p). */
int64_t lidal_nn_grid_bytes(int64_t p);
int64_t lidal_nn_grid_workspace_bytes(int64_t p);
int lidal_nn_grid_build(const double* pts, int64_t p, double cell, void* grid, int64_t grid_bytes,
                        void* ws, int64_t ws_bytes, void* stream);
/* replaces score/sv_level/LiDAL.py:59-81 for one query frame against `n_nei` neighbour frames
 * (host arrays of device pointers, in the reference's neighbour order):
 *   interd f64 [p] (mean KL over matched neighbours), intere f32 [p] (entropy of the mean
 *   probability), map_count i32 [p] (matches). */
int64_t lidal_interframe_workspace_bytes(int64_t p, int n_nei);
int lidal_interframe_score(const double* q_pts, const float* q_prob, int64_t p, int c,
                           const void* const* nei_grids_host, const double* const* nei_pts_host,
                           const float* const* nei_prob_host, const int64_t* nei_p_host,
                           int n_nei, double dis_thresh, double* interd, float* intere,
                           int32_t* map_count, void* ws, int64_t ws_bytes, void* stream);
/* lidal_interframe_score with the query points taken in CELL order (round 6): q_grid = the grid lidal_nn_grid_build made of
 * q_pts themselves (every frame of a sequence has one: it is the neighbour of others), or NULL = the call above.  The
 * queries of a wave then share their cells' cache lines.  Same outputs bit for bit, in the points' own order. */
int lidal_interframe_score_ordered(const double* q_pts, const float* q_prob, int64_t p, int c,
                                   const void* const* nei_grids_host, const double* const* nei_pts_host,
                                   const float* const* nei_prob_host, const int64_t* nei_p_host,
                                   int n_nei, double dis_thresh, double* interd, float* intere,
                                   int32_t* map_count, void* ws, int64_t ws_bytes, const void* q_grid,
                                   void* stream);
/* replaces score/sv_level/LiDAL.py:91-98: per-supervoxel means over point lists given as CSR
 * (sv_ptr i64 [s+1], sv_idx i64 [sv_ptr[s]]).  sv_interd f32 [s], sv_intere f32 [s],
 * sv_center f32 [s,3]. */
int lidal_supervoxel_reduce(const double* interd, const float* intere, const double* pts,
                            const int64_t* sv_ptr, const int64_t* sv_idx, int s, float* sv_interd,
                            float* sv_intere, float* sv_center, void* stream);

/* ---- row-wise helpers of a planned step (what torch glue did between the operators) ------------- */
/* dst[r][0 : row_bytes) = src[r][0 : row_bytes), dst[r][row_bytes : row_bytes + zero_bytes) = 0 for r < rows; rows
 * `src_pitch` / `dst_pitch` bytes apart.  One call per summand is torchsparse.cat (operators.py; network/spvcnn.py:
 * 133,137,145,149) into the channel slices of one buffer; with zero_bytes > 0 it is the channel padding of the
 * 19-class logit gradient; with src_pitch > row_bytes the contiguous copy of a channel slice. */
int lidal_copy2d(const void* src, int64_t src_pitch, void* dst, int64_t dst_pitch, int64_t rows, int64_t row_bytes,
                 int64_t zero_bytes, void* stream);
/* out[r][:c] = a[r][:c] + b[r][:c] (f32 sum rounded once to `dtype`): the sum autograd forms where a tensor has
 * two consumers (an encoder level feeds the next stage AND a decoder concatenation: network/spvcnn.py:120-128);
 * row strides in elements (a channel slice of a concatenation's gradient is read in place); c and the strides
 * whole 16-byte vectors. */
int lidal_add2d(const void* a, int64_t a_stride, const void* b, int64_t b_stride, void* out, int64_t out_stride,
                int64_t rows, int c, int dtype, void* stream);
/* dst f32 [cols][rows] = transpose of src f32 [rows][cols] (rows `src_stride` floats apart): nn.Linear keeps its
 * weight as [Cout][Cin], the weight-gradient kernel produces x^T g = [Cin][Cout]. */
int lidal_transpose_f32(const float* src, int64_t src_stride, float* dst, int rows, int cols, void* stream);
/* dst bf16 [n][c_dst] = src f32 [n][c_src] rounded to nearest even, channels c_src.. zero-filled (the 4-channel
 * input of the stem under bf16: 8-byte rows padded to one 16-byte vector, nn/functional/conv.py _pad_channels). */
int lidal_cast_rows_bf16(const float* src, int c_src, void* dst, int c_dst, int64_t n, void* stream);

/* ---- launch plans ----------------------------------------------------------------------------------- */
/* A whole forward or backward pass as ONE call: `words` is a stream of n_ops operations,
 *     kind | flags << 16,  arg 0, arg 1, ...
 * one 64-bit word each; an operation of kind LIDAL_OP_X calls lidal_x with the words as its arguments IN THE ORDER
 * OF ITS PROTOTYPE ABOVE, stream excluded: pointers as addresses, integers as such, float arguments as the bit
 * pattern of a double (lidal_plan_op_args(kind) words).  The operations are queued on `stream` in order --
 * exactly the launches the same calls made one by one would queue (train.py:127-140 / prob_inference.py:91-113
 * issue theirs one Python call at a time); nothing is captured or cached, every call walks the words it is given.
 * The flags of an operation name the stream it is queued on: 0 = `stream`, i = side stream i (1, LIDAL_OP_FLAG_SIDE, is
 * `side_stream`, which may be NULL if unused; lidal_plan_run_streams takes streams[0] = the main stream and up to 7 side
 * streams), between a LIDAL_OP_FORK_SIDE with the same flags (side stream i waits for what the main stream holds so
 * far) and a LIDAL_OP_JOIN_SIDE (the main stream waits for side stream i).  Stops at the first failing operation;
 * lidal_last_error() names its index and kind. */
enum {
  LIDAL_OP_CONV_WEIGHT_IMAGE_BATCH = 1, LIDAL_OP_CONV_APPLY_IMAGE = 2, LIDAL_OP_CONV_DGRAD_BN_SUMS = 3,
  LIDAL_OP_CONV_WGRAD = 4, LIDAL_OP_BN_TRAIN_FWD = 5, LIDAL_OP_BN_TRAIN_FWD_TILES = 6, LIDAL_OP_BN_BWD = 7,
  LIDAL_OP_BN_BWD_TILES = 8, LIDAL_OP_BN_EVAL_FWD = 9, LIDAL_OP_BN_FOLD = 10, LIDAL_OP_COLSUM = 11,
  LIDAL_OP_ADD_RELU_FWD = 12, LIDAL_OP_ADD_RELU_BWD = 13, LIDAL_OP_VOXELIZE_FWD_1TO1 = 14,
  LIDAL_OP_VOXELIZE_FWD_SORTED = 15, LIDAL_OP_VOXELIZE_BWD = 16, LIDAL_OP_DEVOXELIZE_FWD = 17,
  LIDAL_OP_DEVOXELIZE_BWD_SORTED = 18, LIDAL_OP_CE_FWD = 19, LIDAL_OP_CE_BWD = 20, LIDAL_OP_COPY2D = 21,
  LIDAL_OP_ADD2D = 22, LIDAL_OP_TRANSPOSE_F32 = 23, LIDAL_OP_CAST_ROWS_BF16 = 24, LIDAL_OP_VIEW_MEAN_SOFTMAX = 25,
  LIDAL_OP_FORK_SIDE = 26, LIDAL_OP_JOIN_SIDE = 27, LIDAL_OP_CONV_APPLY_IMAGE_WS = 28,
  LIDAL_OP_CONV_DGRAD_BN_SUMS_WS = 29, LIDAL_OP_ADD_RELU_BWD_BN_SUMS = 30, LIDAL_OP_BN_BWD_FROM_SUMS = 31,
  LIDAL_OP_ADD_RELU_BWD_BN_TILE_SUMS = 32, LIDAL_OP_DEVOXELIZE_BWD_CELLS = 33,
  LIDAL_OP_CONV_WGRAD_STREAMS = 34
};
#define LIDAL_OP_FLAG_SIDE 1
int lidal_plan_op_args(int kind);
/* test / debug aid: synchronous copy of `nbytes` of device memory at address `dev` to `host` (a plan names its
 * buffers by address; tests that replay its operations against the oracle read the operands back with this) */
int lidal_debug_read(const void* dev, void* host, int64_t nbytes);
int lidal_plan_run(const int64_t* words, int64_t n_words, int64_t n_ops, void* stream, void* side_stream);
int lidal_plan_run_streams(const int64_t* words, int64_t n_words, int64_t n_ops, void* const* streams, int n_streams);

#ifdef __cplusplus
}
#endif
#endif /* LIDAL_AMD_H */

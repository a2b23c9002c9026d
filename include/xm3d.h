/* xm3d.h - C ABI of libxm3d_hip.so: the MI355X (gfx950) hot path of XMask3D.
 *
 * Plain pointers and sizes only; no torch types.  Every pointer is a DEVICE
 * pointer unless the comment says HOST.  `stream` is a hipStream_t passed as
 * void* (NULL = the null stream).  All entry points return 0 on success or a
 * negative XM3D_E* code; xm3d_last_error() gives the message (thread local).
 * Kernels are enqueued on `stream`; only the entry points documented as
 * "syncs" wait for the device (they return a count the caller needs on the
 * host to size the next allocation - the reference returns arrays of that
 * length at the same point, so the sync is inherent in its interface too).
 *
 * Each entry point names the reference interface it replaces (file:line under
 * the upstream XMask3D tree).
 */
#ifndef XM3D_H
#define XM3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XM3D_OK 0
#define XM3D_EINVAL -1   /* bad argument (shape, alignment, range)          */
#define XM3D_EHIP -2     /* HIP runtime error                              */
#define XM3D_ENOSPC -3   /* workspace too small / hash table overflow      */
#define XM3D_ERANGE -4   /* coordinate outside the packable range          */

const char* xm3d_last_error(void);
int xm3d_version(void);
/* Fills HOST ints: number of CUs, wavefront size, and gcnArchName (<=63 chars) of device `dev`. */
int xm3d_device_info(int dev, int* n_cu, int* wave, char* arch64);

/* ---------------------------------------------------------------------------
 * Voxelisation  (replaces dataset/voxelizer.py:81-132 Voxelizer.voxelize and
 * dataset/voxelization_utils.py:6-18,38-102 fnv_hash_vec / sparse_quantize)
 *
 * xyz      (n,3) f64 metres
 * T16      HOST 16 doubles, row-major 4x4 rigid transform (M_r @ M_v)
 * grid     (n,3) i32 out: first n_unique rows = integer voxel coords (min-shifted),
 *                 in ascending-FNV-key order (== reference `locs`)
 * inds     (n)   i64 out: first n_unique = index of first point of each voxel
 * inverse  (n)   i64 out: voxel row of every point (== inds_reconstruct)
 * n_unique HOST  i64 out.                                      SYNCS `stream`.
 * ws/ws_bytes    scratch; query the size with xm3d_voxelize_ws_bytes(n).
 * ------------------------------------------------------------------------- */
int xm3d_voxelize_ws_bytes(int64_t n, size_t* bytes);
int xm3d_voxelize(const double* xyz, int64_t n, const double* T16, int32_t* grid, int64_t* inds,
                  int64_t* inverse, int64_t* n_unique, void* ws, size_t ws_bytes, void* stream);
/* FNV keys only (known-answer tests): grid (n,3) i32 -> keys (n) u64. */
int xm3d_fnv_keys(const int32_t* grid, int64_t n, uint64_t* keys, void* stream);

/* ---------------------------------------------------------------------------
 * Coordinate manager primitives (replace MinkowskiEngine's coordinate manager,
 * reached from run/train.py:483, run/infer.py:462 `SparseTensor(feats, coords)`
 * and implicitly from every ME.MinkowskiConvolution in
 * models/modeling/meta_arch/mink_unet.py:47-109).
 *
 * Coordinates are (n,4) i32 rows [batch, x, y, z]; packable range is
 * 0 <= batch < 32768, -32768 <= x,y,z < 32768 (XM3D_ERANGE otherwise; checked
 * on device, reported by the next syncing call or xm3d_check_flag()).
 * ------------------------------------------------------------------------- */
/* Strided coordinate set: out = unique(floor(c/ts_out)*ts_out), ascending (b,x,y,z).
 * out_coords must hold n rows; *n_out HOST.  SYNCS.  ws from xm3d_stride_ws_bytes. */
int xm3d_stride_ws_bytes(int64_t n, size_t* bytes);
int xm3d_coords_stride(const int32_t* coords, int64_t n, int32_t ts_out, int32_t* out_coords,
                       int64_t* n_out, void* ws, size_t ws_bytes, void* stream);
/* Spatially sorted processing order of a coordinate set: order[i] = row index of the
 * i-th coordinate in ascending (b,x,y,z).  Also reports duplicates: *n_unique HOST (SYNCS). */
int xm3d_coords_order(const int32_t* coords, int64_t n, int32_t* order, int64_t* n_unique,
                      void* ws, size_t ws_bytes, void* stream);
/* Open-addressing hash coords->row.  cap must be a power of two >= 2n.
 * table_keys (cap) u64, table_vals (cap) i32: filled by this call. */
int xm3d_hash_build(const int32_t* coords, int64_t n, uint64_t* table_keys, int32_t* table_vals,
                    int64_t cap, void* stream);
/* Neighbour table ("rulebook", output-stationary form): for every output row o and
 * kernel offset k (x fastest; odd k centred, even k in {0..k-1}; scaled by `ts`):
 *   nbr[k*n_out + o] = row of the hashed set at out_coords[o] + sign*offset_k, or -1.
 * sign=+1: stride-1 and strided convs (hash = input set);
 * sign=-1: transposed conv (hash = coarse input set, out_coords = fine set). */
int xm3d_kernel_map(const int32_t* out_coords, int64_t n_out, const uint64_t* table_keys,
                    const int32_t* table_vals, int64_t cap, int32_t ksize, int32_t ts, int32_t sign,
                    int32_t* nbr, void* stream);
/* Reads and clears the device error flag (SYNCS the device). 0 or XM3D_ERANGE/XM3D_ENOSPC. */
int xm3d_check_flag(void);

/* ---------------------------------------------------------------------------
 * Sparse convolution (replaces ME.MinkowskiConvolution / ConvolutionTranspose
 * forward+backward, mink_unet.py:47-109, resnet_base.py:68, and the fused
 * MinkowskiBatchNorm(eval)+MinkowskiReLU+residual epilogue of BasicBlock).
 *
 *   out[o,:] = epi( sum_k  in[nbr[k,o],:] @ W[k] )           W: (K,Cin,Cout) f32
 *   epi(v)   = relu?( v*scale + shift + residual[o,:] )      (each part optional)
 *
 * order (n_out) i32 or NULL: processing order of output rows (locality only,
 * never changes results).  algo: 0 = auto, 1 = scalar reference kernel (W as
 * given), 2 = MFMA f32 kernel: W must then be the fragment-ordered buffer made
 * by xm3d_spconv_pack_weight and Cin, Cout multiples of 32 (the 3-channel stem
 * runs on algo 1).  nbr may be NULL for K=1 (identity map, plain GEMM).
 * ------------------------------------------------------------------------- */
int xm3d_spconv_fwd(const float* in, int64_t n_in, int32_t cin, const float* W, int32_t K, int32_t cout,
                    const int32_t* nbr, const int32_t* order, int64_t n_out, const float* scale,
                    const float* shift, const float* residual, int32_t relu, float* out, int32_t algo,
                    void* stream);
/* Tiled rulebook (algo 3): compacts a neighbour table once per kernel map into per-workgroup
 * pair lists shared by every conv on that map.  Row tile b = 256 consecutive rows of `order`;
 * tsrc/tdst[(b*K + k)*256 + j] = input row / local output row of the j-th valid pair of
 * (b, k), tcnt[b*K + k] = number of pairs.  Buffers: tsrc i32 and tdst u8 of ntiles*K*256
 * entries, tcnt i32 of ntiles*K, ntiles = ceil(n_out/256).  nbr NULL = K=1 identity map.
 * xm3d_spconv_fwd_tiles then computes the same result as xm3d_spconv_fwd(algo 2) from those
 * lists (Wp = packed weights; Cin, Cout multiples of 32).  ksplit > 1 spreads the kernel offsets
 * of one row tile over ksplit workgroups (small grids): partial sums go to `slab`
 * (ksplit*n_out*cout f32) and are reduced in fixed order with the epilogue fused (deterministic). */
int xm3d_rulebook_tiles(const int32_t* nbr, const int32_t* order, int64_t n_out, int32_t K, int32_t* tsrc,
                        uint8_t* tdst, int32_t* tcnt, void* stream);
/* Output channels one workgroup of xm3d_spconv_fwd_tiles owns for a (cin, cout) layer (32; 48 only under the experiment
 * switch XM3D_SPCONV_CT=48): callers size the split-K factor from the resulting workgroup count. */
int xm3d_spconv_tile_channels(int32_t cin, int32_t cout);
int xm3d_spconv_fwd_tiles(const float* in, int64_t n_in, int32_t cin, const float* Wp, int32_t K, int32_t cout,
                          const int32_t* tsrc, const uint8_t* tdst, const int32_t* tcnt, const int32_t* order,
                          int64_t n_out, const float* scale, const float* shift, const float* residual,
                          int32_t relu, float* out, int32_t ksplit, float* slab, void* stream);
/* Split-operand form (algo 4, the default on the MinkUNets): same tiled rulebook, same result contract and epilogue as
 * xm3d_spconv_fwd_tiles, computed on the bf16 matrix cores with every f32 operand split into two bf16 terms (three
 * products per f32 product, f32 accumulation: error <= ~2^-16 per term, inside the 2e-5 per-conv bound of the f32 kernel;
 * summation order fixed -> bitwise reproducible).  Wq = xm3d_spconv_pack_weight_split(W): K*cin*cout*4 bytes (bf16 hi and
 * lo parts in MFMA-fragment order).  Cin, Cout multiples of 32.  xm3d_spconv_split_channels(cout) = output channels one
 * workgroup owns (96 / 64 / 32): callers size the split-K factor from the resulting workgroup count. */
/* xm3d_spconv_fwd_split2: the same with the activations ALSO kept pre-split between layers: in_split (NULL or (2, n_in, cin)
 * bf16 = hi plane then lo plane of `in`, as written by a previous call) spares the gather path the conversion; out_split
 * (NULL or (2, n_out, cout) bf16) receives the split copy of `out` from the epilogue.  `in` (f32) is still required. */
int xm3d_spconv_fwd_split2(const float* in, const void* in_split, int64_t n_in, int32_t cin, const void* Wq, int32_t K,
                           int32_t cout, const int32_t* tsrc, const uint8_t* tdst, const int32_t* tcnt, const int32_t* order,
                           int64_t n_out, const float* scale, const float* shift, const float* residual, int32_t relu,
                           float* out, void* out_split, int32_t ksplit, float* slab, void* stream);
/* The plain-bf16 form of the same kernel - the sparse convolution of the bf16 configuration (BASELINE config 2 names bf16): activations
 * are ONE bf16 plane (n, C) in and out (residual too), a product is one MFMA on bf16(W) (the hi plane of the image
 * xm3d_spconv_pack_weight_split wrote), f32 accumulation, the same tiled rulebook and order of additions (bit-reproducible):
 *     out = bf16( relu( scale * sum_k in[nbr[k]] @ bf16(W[k]) + shift + residual ) )
 * A third of the matrix work and half the gathered / written bytes of xm3d_spconv_fwd_split2; accuracy ~1e-2 at the end of
 * MinkUNet34C (bf16 operands and activations) - the level of the bf16 dense branch, not north_star's 1e-3 (that is the split form).
 * Replaces ME.MinkowskiConvolution(+Transpose) + BN / ReLU / residual tail like the calls above (mink_unet.py:118-178). */
int xm3d_spconv_fwd_bf16(const void* in, int64_t n_in, int32_t cin, const void* Wq, int32_t K, int32_t cout, const int32_t* tsrc,
                         const uint8_t* tdst, const int32_t* tcnt, const int32_t* order, int64_t n_out, const float* scale, const float* shift,
                         const void* residual, int32_t relu, void* out, int32_t ksplit, float* slab, void* stream);
int xm3d_spconv_pack_weight_split(const float* W, int32_t K, int32_t cin, int32_t cout, void* Wq, void* stream);
int xm3d_spconv_split_channels(int32_t cout);
int xm3d_spconv_fwd_split(const float* in, int64_t n_in, int32_t cin, const void* Wq, int32_t K, int32_t cout,
                          const int32_t* tsrc, const uint8_t* tdst, const int32_t* tcnt, const int32_t* order,
                          int64_t n_out, const float* scale, const float* shift, const float* residual,
                          int32_t relu, float* out, int32_t ksplit, float* slab, void* stream);
/* Pre-pack W (K,Cin,Cout) into the MFMA B-fragment layout used by algo 2 (same byte size). */
int xm3d_spconv_pack_weight(const float* W, int32_t K, int32_t cin, int32_t cout, float* Wp, void* stream);
/* dgrad: gin[i,:] = sum over (k,o) with nbr[k,o]==i of gout[o,:] @ W[k]^T, computed as a
 * forward conv over the INVERSE map nbr_t (K,n_in) (xm3d_kernel_map_invert) with the
 * per-offset transposed kernels Wt (K,Cout,Cin) (packed for algo 2, like spconv_fwd).
 * wgrad: gW[k] = sum_o in[nbr[k,o],:]^T gout[o,:]  (gW fully overwritten; f32 atomics). */
int xm3d_kernel_map_invert(const int32_t* nbr, int32_t K, int64_t n_out, int64_t n_in, int32_t* nbr_t,
                           void* stream);
int xm3d_spconv_bwd_data(const float* gout, int64_t n_out, int32_t cout, const float* Wt, int32_t K,
                         int32_t cin, const int32_t* nbr_t, const int32_t* order, int64_t n_in, float* gin,
                         int32_t algo, void* stream);
int xm3d_spconv_bwd_weight(const float* in, int64_t n_in, int32_t cin, const float* gout, int64_t n_out,
                           int32_t cout, const int32_t* nbr, int32_t K, float* gW, void* stream);

/* Row-wise batch norm over a (n,c) matrix (ME.MinkowskiBatchNorm == BatchNorm1d,
 * mink_unet.py:51..104).  stats: sum (c) and sumsq (c) in f64 scratch (2*c doubles). */
int xm3d_bn_stats(const float* x, int64_t n, int32_t c, double* sum_sumsq, void* stream);
int xm3d_affine_act(const float* x, int64_t n, int32_t c, const float* scale, const float* shift,
                    const float* residual, int32_t relu, float* out, void* stream);
/* Training-mode BatchNorm (ME.MinkowskiBatchNorm / MinkowskiSyncBatchNorm in train(), mink_unet.py:51-116; backward reached
 * from loss.backward(), run/train.py:537).  packed = [sum(c), sumsq(c), count] f64 (xm3d_bn_stats output, all-reduced over
 * the ranks for SyncBatchNorm); total >= 0 overrides packed[2c] (single rank: the host knows the row count).
 * -> mean, invstd, scale = invstd*w, shift = -mean*invstd*w + b (all f32 (c)), total_out (1); momentum >= 0 also updates
 * running_mean / running_var (unbiased variance) / num_batches in place. */
int xm3d_bn_finalize(const double* packed, int32_t c, double total, const float* weight, const float* bias, float eps,
                     float momentum, float* running_mean, float* running_var, int64_t* num_batches, float* mean, float* invstd,
                     float* scale, float* shift, float* total_out, void* stream);
/* sums = [sum_r gy, sum_r gy*xhat] f64 (2c) with xhat = (x-mean)*invstd (all-reduce it for SyncBatchNorm), then
 * gx = w*invstd*(gy - sum_dy/total - xhat*sum_dy_xhat/total), gw = sum_dy_xhat, gb = sum_dy (gw/gb may be NULL). */
int xm3d_bn_bwd_reduce(const float* gy, const float* x, int64_t n, int32_t c, const float* mean, const float* invstd, double* sums,
                       void* stream);
int xm3d_bn_bwd_apply(const float* gy, const float* x, int64_t n, int32_t c, const float* mean, const float* invstd,
                      const float* weight, const double* sums, const float* total, float* gx, float* gw, float* gb, void* stream);

/* ---------------------------------------------------------------------------
 * Fused GroupNorm (+SiLU) over NCHW activations (replaces nn.GroupNorm(32,C) [+ x*sigmoid(x)] inside the SD
 * VAE/UNet blocks the extractor runs, models/modeling/meta_arch/ldm.py:386-490, and the GN of the projection
 * bottlenecks, backbone/feature_extractor.py:40-47).  x, y: (B,C,H*W) contiguous, dtype 0 = f32, 1 = bf16;
 * gamma/beta: (C) in the SAME dtype or NULL; silu: 0 = none, 1 = SiLU, 2 = ReLU; stats_ws: xm3d_gn_stats_doubles_nchw(B, C, hw, G)
 * doubles of scratch.  y may alias x.
 * STATISTICS BUFFERS of this section and of the convolution below: the first B*G*2 doubles receive the moments (sum, sum of
 * squares per (sample, group)); the doubles behind them hold the per-workgroup partial pairs the moments are summed from in a FIXED
 * order - no floating-point atomics, so the moments (and everything computed from them) are bit-identical from run to run, like the
 * reference's inference forward (no atomics in ms_deform_im2col_cuda.cuh:242-304 or in nn.GroupNorm).  Consumers
 * (xm3d_group_norm_nhwc_apply, xm3d_conv3x3_nhwc's gn_stats) only read the first B*G*2 doubles.
 * ------------------------------------------------------------------------- */
int64_t xm3d_gn_stats_doubles_nchw(int64_t B, int32_t C, int32_t hw, int32_t G);
/* size in doubles of the `stats` / `stats_ws` argument of the channels-last calls (xm3d_group_norm_nhwc[_res], _stats, xm3d_bias_residual_stats_nhwc) */
int64_t xm3d_gn_stats_doubles_nhwc(int64_t B, int32_t C, int32_t hw, int32_t G, int32_t dtype);
int xm3d_group_norm(const void* x, int32_t dtype, int64_t B, int32_t C, int32_t hw, int32_t G, const void* gamma,
                    const void* beta, float eps, int32_t silu, void* y, double* stats_ws, void* stream);
/* Same for channels-last activations: x, y are (B, H*W, C) contiguous (NHWC); C a multiple of 4 (f32) / 8 (bf16), G <= 64.
 * shift (optional, dtype of x): a per-channel term added to x before the normalisation, y = GN(x + shift[c]) - the bias of
 * the convolution that produced x and/or the timestep-embedding term of ldm's ResBlock (openaimodel.py ResBlock._forward),
 * which then need no pass of their own.  shift_bstride = 0: one (C) vector for all samples; = C: a (B, C) matrix. */
int xm3d_group_norm_nhwc(const void* x, const void* shift, int32_t shift_bstride, int32_t dtype, int64_t B, int32_t C, int32_t hw,
                         int32_t G, const void* gamma, const void* beta, float eps, int32_t silu, void* y, double* stats_ws,
                         void* stream);
/* ... with a residual (B, H*W, C) or NULL added after the affine and before the activation: y = act(GN(x + shift) + residual) -
 * the tail of detectron2's BottleneckBlock in the projection bottlenecks, relu(conv3_norm(out) + shortcut)
 * (backbone/feature_extractor.py:40-47), in the apply pass instead of an add and a ReLU pass of their own. */
int xm3d_group_norm_nhwc_res(const void* x, const void* shift, int32_t shift_bstride, int32_t dtype, int64_t B, int32_t C, int32_t hw,
                             int32_t G, const void* gamma, const void* beta, float eps, int32_t silu, const void* residual, void* y,
                             double* stats_ws, void* stream);

/* out = a + b + bias[c] like xm3d_bias_residual_nhwc, plus the GroupNorm statistics of `out` (sum, sum of squares per (sample,
 * group), over the values as stored) into stats (xm3d_gn_stats_doubles_nhwc doubles, moments first): the GroupNorm that consumes `out` next - norm1 of
 * the following ResBlock, the norm of an attention block - then runs xm3d_group_norm_nhwc_apply with them and skips its
 * statistics pass.  a may be NULL; (B, H*W, C) channels-last, C a multiple of 4 (f32) / 8 (bf16), G <= 64. */
int xm3d_bias_residual_stats_nhwc(const void* a, const void* b, const void* bias, int32_t dtype, int64_t B, int32_t C, int32_t hw, int32_t G,
                                  void* out, double* stats, void* stream);
/* The apply pass of xm3d_group_norm_nhwc_res alone, on statistics the caller already holds (of x + shift if a shift is given). */
int xm3d_group_norm_nhwc_apply(const void* x, const void* shift, int32_t shift_bstride, int32_t dtype, int64_t B, int32_t C, int32_t hw,
                               int32_t G, const void* gamma, const void* beta, float eps, int32_t silu, const void* residual, void* y,
                               const double* stats, void* stream);

/* ---- fused ResnetBlock convolution (conv.hip): the GroupNorm(32) -> SiLU -> Conv2d(3x3, pad 1) half of ldm's ResnetBlock /
 * ResBlock, executed from models/modeling/meta_arch/ldm.py:386-414 (VAE encoder), :425-446 (UNet), :448-490 (VAE decoder)
 * through `ldm.modules.diffusionmodules.model.ResnetBlock.forward` / `openaimodel.ResBlock._forward` (un-vendored
 * stable-diffusion-sdkit 2.1.3) -> torch.nn.GroupNorm + torch.nn.Conv2d (cuDNN / MIOpen there).
 *   out = conv3x3( SiLU( GroupNorm(x) ) ) + bias (+ residual)                      bf16 NHWC in / f32 accumulate / bf16 NHWC out
 * in ONE launch on the matrix cores: the normalisation is applied while the input halo tile is staged in LDS (statistics come
 * in as f64 moments), bias / per-sample embedding term / skip connection are added in the epilogue, and the moments of `out`
 * for the NEXT GroupNorm are accumulated there as well.
 *   xm3d_conv3x3_cout_tile(cout)  -> output-channel tile the kernel uses for `cout` (256 or 128; 0 = unsupported: cout % 32 != 0);
 *   xm3d_conv3x3_packed_elems     -> bf16 elements of the packed image (cout padded to whole tiles: 320 -> 384)
 *   xm3d_conv3x3_pack_weight      : w_ohwi (cout, 3, 3, cin) bf16 (= Conv2d.weight.permute(0,2,3,1)) -> packed (xm3d_conv3x3_packed_elems):
 *                                   per (cout tile, 32-row block) one contiguous stream of MFMA A fragments, 1 KiB per k-step
 *   xm3d_conv3x3_nhwc             : x (B, H>>upsample, W>>upsample, cin) bf16; out / residual (B, H, W, cout) bf16.
 *       gn_stats (B, groups, 2) f64 sum / sum of squares of x over each (sample, group), gamma / beta (cin) f32, act = 1 (SiLU) or
 *       2 (ReLU: detectron2's GroupNorm BottleneckBlock of the projection backbone, backbone/feature_extractor.py:20-60);
 *       in_shift (cin) f32 with in_shift_bstride 0, or (B, cin) with in_shift_bstride = cin, or NULL: the GroupNorm input is x + in_shift
 *       (a convolution bias the producer of x left to its consumer) and gn_stats are the moments of that sum.
 *       gn_stats NULL (act 0): plain convolution of x; upsample = 1 (plain only): x is first nearest-upsampled 2x (ldm Upsample).
 *       bias (cout) f32 with bias_bstride 0, or (B, cout) with bias_bstride = cout (conv bias + timestep-embedding term), or NULL.
 *       stats_out: xm3d_conv3x3_stats_doubles(B, H, W, cout, cout_tile, groups_out, waves) doubles or NULL: the first B*groups_out*2 receive
 *       the moments of the bf16 values stored to out (written, not accumulated: no zeroing needed); the rest is the scratch of the
 *       fixed-order reduction (see "STATISTICS BUFFERS" above).
 *       ws: xm3d_conv3x3_ws_bytes(B, cin) bytes of device scratch (the per-(image, channel) affine derived from the moments by a
 *       small kernel in front of the convolution); may be NULL without GroupNorm.
 *       waves: 0 (choose), 8 or 4 - the workgroup geometry, results do not depend on it.
 *   Constraints: H % 8 == 0 (waves 8) or H % 4 == 0 (waves 4), W % 32 == 0, cin % 64 == 0, cout % 32 == 0, (cout / groups_out) % 4 == 0, 16-byte aligned tensors.
 *   Launched on `stream`, no host synchronisation. */
int xm3d_conv3x3_cout_tile(int32_t cout);
int64_t xm3d_conv3x3_packed_elems(int32_t cout, int32_t cin, int32_t cout_tile);
int xm3d_conv3x3_pack_weight(const void* w_ohwi, int32_t cout, int32_t cin, int32_t cout_tile, void* packed, void* stream);
int64_t xm3d_conv3x3_ws_bytes(int64_t B, int32_t cin);
int64_t xm3d_conv3x3_stats_doubles(int64_t B, int32_t H, int32_t W, int32_t cout, int32_t cout_tile, int32_t groups_out, int32_t waves);
int xm3d_conv3x3_nhwc(const void* x, int64_t B, int32_t H, int32_t W, int32_t cin, const void* wpacked, int32_t cout, int32_t cout_tile,
                      const double* gn_stats, const float* gamma, const float* beta, const float* in_shift, int32_t in_shift_bstride,
                      float eps, int32_t groups, int32_t act, const float* bias, int32_t bias_bstride, const void* residual, void* out, double* stats_out,
                      int32_t groups_out, int32_t upsample, int32_t waves, void* ws, void* stream);
/* waves of a workgroup the call above uses when waves = 0: 8 (8 x 32 pixel tile, one workgroup per CU) or 4 (4 x 32 pixel tile, two
 * workgroups per CU, whose memory-bound prologue / epilogue overlap each other's matrix work) */
int xm3d_conv3x3_default_waves(int32_t H, int32_t W, int32_t cin, int32_t cout);

/* ---- the same convolution to f32 accuracy from bf16 matrix-core passes (conv.hip): the fp32 configuration of the frozen nets (the
 * reference's own arithmetic, torch.nn.Conv2d in f32) on split operands: x = x_hi + x_lo, w = w_hi + w_lo (bf16 each),
 *   conv(x, w) = conv(x_hi, w_hi) + conv(x_hi, w_lo) + conv(x_lo, w_hi)   + O(2^-16 |x w|),   accumulated in the f32 output.
 *   xm3d_split_bf16_nhwc     : x (B, H*W, C) f32 -> hi, lo [, lo2: a third term, or NULL] bf16 (same shape); with gn_stats: y = act(GroupNorm(x + in_shift)) is split
 *                              (act 0 none, 1 SiLU, 2 ReLU; ws = xm3d_conv3x3_ws_bytes(B, C) bytes of scratch)
 *   xm3d_conv3x3_nhwc_f32acc : out f32 (B, H, W, cout) = conv3x3(x bf16, packed bf16 weights) + bias + residual f32 (residual may be `out`:
 *                              accumulation in place); stats_out as xm3d_conv3x3_nhwc (moments of the f32 values written).
 *   A full layer = one split + three accumulating launches (two terms: 2e-5 of max|out|) or six (three terms, hi*hi + hi*lo + lo*hi +
 *   hi*lo2 + lo2*hi + lo*lo: f32-exact to ~1e-6) (ops.conv3x3_f32).  Same shape constraints as xm3d_conv3x3_nhwc. */
int xm3d_split_bf16_nhwc(const float* x, int64_t B, int64_t HW, int32_t C, const double* gn_stats, const float* gamma, const float* beta,
                         const float* in_shift, int32_t in_shift_bstride, float eps, int32_t groups, int32_t act, void* hi, void* lo, void* lo2,
                         void* ws, void* stream);
int xm3d_conv3x3_nhwc_f32acc(const void* x, int64_t B, int32_t H, int32_t W, int32_t cin, const void* wpacked, int32_t cout, int32_t cout_tile,
                             const float* bias, int32_t bias_bstride, const float* residual, float* out, double* stats_out, int32_t groups_out,
                             int32_t upsample, int32_t waves, void* stream);
/* The same split in IEEE HALVES, the default of the fp32 configuration since round 4: two terms x = hi / s + lo / (2048 s) carry 22
 * mantissa bits (three bf16 terms: 24, two: 16), so  conv(x, w) = [hi*whi] / (s t) + [hi*wlo + lo*whi] / (2048 s t)  is f32-exact to
 * ~1e-6 in THREE passes instead of six (dropped: lo*wlo, 2^-22).  Both terms of an operand live at the magnitude of x * s - the half's
 * narrow exponent range costs no bits; s, t are powers of two (exact), chosen so that |x| s <= 65504 (xm3d_split_f16_nhwc sets the
 * sticky range flag beyond that: xm3d_check_flag).
 *   xm3d_split_f16_nhwc       : as xm3d_split_bf16_nhwc, hi / lo are halves, scale_hi = s
 *   xm3d_conv3x3_nhwc_f32acc2 : as xm3d_conv3x3_nhwc_f32acc with f16 = 1: x and wpacked hold halves (the packer only moves 16-bit
 *                               words); out = residual + alpha * conv(x, w) + bias, alpha = the term pair's 1 / (scale product) */
int xm3d_split_f16_nhwc(const float* x, int64_t B, int64_t HW, int32_t C, const double* gn_stats, const float* gamma, const float* beta,
                        const float* in_shift, int32_t in_shift_bstride, float eps, int32_t groups, int32_t act, float scale_hi, void* hi, void* lo,
                        void* ws, void* stream);
int xm3d_conv3x3_nhwc_f32acc2(const void* x, int64_t B, int32_t H, int32_t W, int32_t cin, const void* wpacked, int32_t cout, int32_t cout_tile,
                              const float* bias, int32_t bias_bstride, const float* residual, float* out, double* stats_out, int32_t groups_out,
                              int32_t upsample, int32_t waves, int32_t f16, float alpha, void* stream);

/* ---- linear layer / 1x1 convolution with fused epilogue (gemm.hip): the dense projections of the Stable-Diffusion UNet's
 * SpatialTransformer (to_q / to_k / to_v / to_out, GEGLU feed-forward, 1x1 proj_in / proj_out: models/modeling/meta_arch/ldm.py:425-446
 * -> ldm.modules.attention.{CrossAttention, FeedForward, SpatialTransformer}.forward -> torch.nn.Linear / Conv2d(1x1), cuBLAS /
 * hipBLASLt there), the 1x1 q / k / v / proj_out of the VAE AttnBlock (:386-414, :448-490) and the c_fc / c_proj / in_proj / out_proj
 * of the mask-CLIP ViT (models/modeling/meta_arch/clip.py:239-270 -> open_clip ResidualAttentionBlock).
 *   out = act( x @ W^T + bias ) (+ residual)           act 0 none, 1 GELU (erf), 2 QuickGELU
 *   out = (x @ Wv^T + bv) * GELU(x @ Wg^T + bg)        act 3 GEGLU: W = [Wv; Wg] (N = 2 N_out rows), out (M, N / 2)
 * bf16 rows in / f32 accumulate / bf16 rows out, one launch.
 *   xm3d_gemm_col_tile(N)        -> column tile for N rows of W (256 or 128)
 *   xm3d_gemm_packed_elems       -> bf16 elements of the packed image (N padded to whole column tiles)
 *   xm3d_gemm_pack_weight        : W (N, K) row-major, f32 (w_is_f32 = 1) or bf16 -> packed; `act` must be the epilogue the image will
 *                                  be used with (GEGLU interleaves value and gate rows)
 *   xm3d_gemm_bf16               : x (M, K) bf16 with row stride ldx; out / residual (M, N_out) with row strides ldo / ldr (elements);
 *                                  bias (N) f32 or NULL; residual may be NULL.
 *                                  col_tile: 256 or 128 - the packed image serves both (an image packed for 256 may be run with 128);
 *                                  waves: 0 (choose: xm3d_gemm_default_waves), 8 (256-row workgroups) or 4 (128-row workgroups, col_tile 128
 *                                  only: small M); results do not depend on either.
 *   Constraints: K % 64 == 0, N % 32 == 0, row strides multiples of 8 elements, M * ldx < 2^31, 16-byte aligned tensors; any M > 0.
 *   Launched on `stream`, no host synchronisation. */
int xm3d_gemm_col_tile(int32_t n_rows);
int64_t xm3d_gemm_packed_elems(int32_t n_rows, int32_t K, int32_t col_tile);
int xm3d_gemm_pack_weight(const void* w, int32_t w_is_f32, int32_t N, int32_t K, int32_t act, int32_t col_tile, void* packed, void* stream);
int xm3d_gemm_default_waves(int64_t M, int32_t N, int32_t col_tile);
int xm3d_gemm_bf16(const void* x, int64_t M, int32_t K, int64_t ldx, const void* wpacked, int32_t N, int32_t col_tile, const float* bias, int32_t act,
                   const void* residual, int64_t ldr, void* out, int64_t ldo, int32_t waves, void* stream);
/* ---- implicit-GEMM convolution on the same kernel (gemm.hip, GF_CONV): the convolutions of the frozen nets that the halo-tile kernel
 * above does not take - ldm's strided Downsample (VAE: F.pad(x, (0,1,0,1)) + Conv2d(3, stride 2, padding 0); UNet: Conv2d(3, stride 2,
 * padding 1)), the 3x3 convolutions of the 16^2 / 8^2 UNet levels, 1x1 convolutions with large K (models/modeling/meta_arch/ldm.py:386-490
 * -> torch.nn.Conv2d, cuDNN / MIOpen there).  x (B, Hin, Win, Cin) channels-last bf16, out / residual (B, Ho, Wo, N) bf16:
 *   out[b, oy, ox, :] = sum_{ky, kx} W[:, ky, kx, :] . x[b, oy * stride - pad_t + ky, ox * stride - pad_l + kx, :] + bias (+ residual)
 * (taps outside the image are zero: any bottom / right padding follows from Ho, Wo).  The token rows of the GEMM are gathered from the
 * image while they are staged in LDS - no im2col tensor.  wpacked = xm3d_gemm_pack_weight of W viewed as (N, ksize*ksize*Cin) in
 * (ky, kx, cin) order (Conv2d.weight.permute(0, 2, 3, 1)); Cin % 64 == 0, N % 32 == 0, ksize 1..3.
 * Grids smaller than two workgroups per CU run a DETERMINISTIC split-K: the K slices write f32 partial slabs into ws
 * (xm3d_conv_gemm_ws_bytes(M = B*Ho*Wo, N, K, col_tile) bytes; 0 = no split for this shape) and a second launch adds them in slice
 * order - bit-reproducible, unlike the atomically accumulated split-K of the library convolutions these calls replace. */
int64_t xm3d_conv_gemm_ws_bytes(int64_t M, int32_t N, int32_t K, int32_t col_tile);
int xm3d_conv_gemm_bf16(const void* x, int64_t B, int32_t Hin, int32_t Win, int32_t Cin, const void* wpacked, int32_t N, int32_t col_tile,
                        int32_t ksize, int32_t stride, int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo, const float* bias,
                        const void* residual, void* out, void* ws, void* stream);
/* ---- f32-accurate GEMM / implicit-GEMM convolution (gemm.hip, GF_F16 | GF_OUT32): the fp32 configuration's Linear / 1x1 / strided /
 * small-map convolutions (torch.nn.Linear / Conv2d in f32 in the reference: run/train.py:178 - no autocast) on the matrix cores, from
 * operands split in IEEE halves as above:  x w = [xhi whi] / s + [xhi wlo + xlo whi] / (2048 s).  ONE pass per call:
 *     out (f32) = act( alpha * (x_term @ W_term^T) + bias + accin ) + residual          act 0 none, 1 GELU, 2 QuickGELU
 * x: (M, K) halves with row stride ldx - or, conv = 1, the channels-last image (B, Hin, Win, Cin) in halves, rows = output pixels as in
 * xm3d_conv_gemm_bf16 (M = B*Ho*Wo, K = ksize*ksize*Cin); wpacked: xm3d_gemm_pack_weight of the half-valued term (16-bit words are
 * moved unchanged); accin (M, N) f32 with row stride ldo (usually `out` itself) or NULL; residual (M, N) f32 row stride ldr or NULL.
 * No atomics, no split-K: bit-reproducible. */
int xm3d_gemm_f32acc(const void* x, int64_t M, int32_t K, int64_t ldx, const void* wpacked, int32_t N, int32_t col_tile, const float* bias, int32_t act,
                     float alpha, const float* accin, const float* residual, int64_t ldr, float* out, int64_t ldo, int32_t waves, int32_t conv,
                     int64_t B, int32_t Hin, int32_t Win, int32_t Cin, int32_t ksize, int32_t stride, int32_t pad_t, int32_t pad_l, int32_t Ho,
                     int32_t Wo, void* stream);
/* The same product in ONE launch (gemm.hip GF_SPLIT3): both half planes of x and both packed half images of W, three MFMAs per k-step into
 * one accumulator.  Needs both terms of an operand at ONE scale (xm3d_split_f16t_nhwc: x s = hi + lo): alpha = 1 / (s t).
 *     out (f32) = act( alpha * (x_hi W_hi^T + x_hi W_lo^T + x_lo W_hi^T) + bias ) + residual
 * A third of the token / output traffic and launches of the three accumulating passes; range |x| s <= 65504 (sticky flag beyond). */
int xm3d_split_f16t_nhwc(const float* x, int64_t B, int64_t HW, int32_t C, const double* gn_stats, const float* gamma, const float* beta,
                         const float* in_shift, int32_t in_shift_bstride, float eps, int32_t groups, int32_t act, float scale_hi, void* hi, void* lo,
                         void* ws, void* stream);
int xm3d_gemm_f32(const void* x_hi, const void* x_lo, int64_t M, int32_t K, int64_t ldx, const void* wp_hi, const void* wp_lo, int32_t N, const float* bias,
                  int32_t act, float alpha, const float* residual, int64_t ldr, float* out, int64_t ldo, int32_t waves, int32_t conv, int64_t B, int32_t Hin,
                  int32_t Win, int32_t Cin, int32_t ksize, int32_t stride, int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo, void* stream);
/* xm3d_gemm_f32 straight from the F32 activation (gemm.hip GF_XF32): x (M, K) f32 rows with row stride ldx floats (conv = 1: the channels-last
 * f32 image) is split into its half planes x x_scale = hi + lo WHILE IT IS STAGED - no xm3d_split_f16t_nhwc pass in front, no half planes in
 * memory; bit-identical to that pass followed by xm3d_gemm_f32.  x_scale: a power of two; |x| x_scale beyond 65504 raises the sticky range flag
 * (xm3d_check_flag).  alpha = 1 / (x_scale t).  Replaces the frozen f32 Linear / Conv2d layers of ldm's UNet / VAE and open_clip's ViT in the
 * reference's fp32 arithmetic (/root/reference/models/modeling/meta_arch/ldm.py:386-490, clip.py:239-270, run/train.py:178). */
int xm3d_gemm_f32x(const float* x, float x_scale, int64_t M, int32_t K, int64_t ldx, const void* wp_hi, const void* wp_lo, int32_t N, const float* bias,
                   int32_t act, float alpha, const float* residual, int64_t ldr, float* out, int64_t ldo, int32_t waves, int32_t conv, int64_t B,
                   int32_t Hin, int32_t Win, int32_t Cin, int32_t ksize, int32_t stride, int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo,
                   void* stream);
/* GroupNorm moments alone, in the layout the calls above take: stats[0 .. B*G*2) f64 <- (sum, sum of squares) of x (+ shift) per
 * (sample, group); x (B, H*W, C) channels-last, dtype 0 = f32 / 1 = bf16; stats holds xm3d_gn_stats_doubles_nhwc doubles. */
int xm3d_group_norm_nhwc_stats(const void* x, const void* shift, int32_t shift_bstride, int32_t dtype, int64_t B, int32_t C, int32_t hw,
                               int32_t G, double* stats, void* stream);

/* ---- pointwise fusions around the frozen nets' convolutions / GEMMs (channels-last, dtype 0 = f32, 1 = bf16) ----
 * out = a + b + bias[c] over (pixels, C) NHWC tensors; a may be NULL (out = b + bias).  Replaces the separate broadcast
 * bias kernel PyTorch-ROCm appends to every MIOpen convolution plus the residual add (ldm ResnetBlock.forward `x + h`). */
int xm3d_bias_residual_nhwc(const void* a, const void* b, const void* bias, int32_t dtype, int64_t pixels, int32_t C, void* out,
                            void* stream);
/* GEGLU gate of ldm's FeedForward (attention.py GEGLU.forward): x (rows, 2*D) contiguous -> out (rows, D) =
 * x[:, :D] * gelu(x[:, D:]) (exact erf GELU, f32 arithmetic). */
int xm3d_geglu(const void* x, int32_t dtype, int64_t rows, int32_t D, void* out, void* stream);
/* LayerNorm over the last dimension with an optional residual add in front: s = x (+ delta); y = LN(s) * gamma + beta.
 * x, delta, sum_out, y: (rows, C) contiguous, gamma / beta (C) or NULL, all of dtype 0 = f32 / 1 = bf16; C a multiple of 4 / 8
 * up to 2048 / 4096.  sum_out (needs delta): receives s, the new residual stream (y then normalises s as stored).  The
 * LayerNorms of ldm's BasicTransformerBlock and open_clip's ResidualAttentionBlock (meta_arch/ldm.py:425-446, clip.py). */
int xm3d_layer_norm(const void* x, const void* delta, int32_t dtype, int64_t rows, int32_t C, const void* gamma, const void* beta, float eps,
                    void* sum_out, void* y, void* stream);
/* y = LayerNorm(x + delta) * gamma + beta over an f32 residual stream x (rows, C), C % 4 == 0, C <= 1024; delta f32 (delta_dtype 0) /
 * bf16 (1) / null.  Written as any of: y (f32), y_bf = bf16(y), ypos_bf = bf16(y + pos) with pos (pos_rows, C) f32 (pos_dtype 0) / bf16 (1),
 * stream row r reading pos row r % pos_rows; out_dtype 0 writes y_bf / ypos_bf as f32 instead (the fp32 configuration: ypos = y + pos).  The post-norm residual blocks of the pixel decoder's deformable-attention encoder and of the
 * masked-attention transformer decoder under bf16 inference (/root/reference/models/modeling/pixel_decoder/msdeformattn.py:35-60,
 * .../transformer_decoder/mask2former_transformer_decoder.py:17-178): one launch instead of cast + add + LayerNorm + add + casts. */
int xm3d_add_layer_norm(const float* x, const void* delta, int32_t delta_dtype, int64_t rows, int32_t C, const float* gamma, const float* beta,
                        float eps, const void* pos, int32_t pos_dtype, int64_t pos_rows, float* y, void* y_bf, void* ypos_bf, int32_t out_dtype, void* stream);
/* Backward of LayerNorm over (rows, C) f32, C % 4 == 0, C <= 2048 (<= 1024 with parameter gradients): dx from x, dy, gamma (f32 or null) - the
 * row statistics are recomputed, nothing but x is kept from the forward.  dgamma / dbeta (C) or null; with either, ws =
 * xm3d_layer_norm_bwd_ws_floats(rows, C) floats of scratch (per-workgroup partial rows, added in index order: no atomics, bit-reproducible).
 * Replaces torch's layer_norm backward under /root/reference/run/train.py:504-540 (nn.LayerNorm of mask2former_transformer_decoder.py:17-178,
 * msdeformattn.py:35-60, the frozen UNet's BasicTransformerBlock norms, meta_arch/ldm.py:425-446). */
int64_t xm3d_layer_norm_bwd_ws_floats(int64_t rows, int32_t C);
int xm3d_layer_norm_bwd(const float* x, const float* dy, const float* gamma, int64_t rows, int32_t C, float eps, float* dx, float* dgamma, float* dbeta,
                        float* ws, void* stream);
/* out[j] = sum over rows of x[r, j]; x (rows, n) f32, row stride ld; ws: xm3d_column_sum_ws_floats(rows, n) floats.  The bias gradient of the
 * trainable linear layers (run/train.py:504-540 through nn.Linear of msdeformattn.py:35-60 / mask2former_transformer_decoder.py): two launches,
 * fixed summation order - bit-reproducible and safe inside a replayed HIP graph. */
int64_t xm3d_column_sum_ws_floats(int64_t rows, int32_t n);
int xm3d_column_sum(const float* x, int64_t rows, int32_t n, int64_t ld, float* out, float* ws, void* stream);
/* Backward of xm3d_group_norm (NCHW f32, (B, C, hw) contiguous, hw % 4 == 0, B * C <= 65535) INCLUDING its fused activation (act 0 none / 1
 * SiLU / 2 ReLU): dx from x, dy, the forward's moments (`stats_ws` of xm3d_group_norm: sum and sum of squares per (sample, group), f64), gamma /
 * beta (f32 (C) or null).  dgamma / dbeta (C) or null (frozen norms).  ws: xm3d_group_norm_bwd_ws_floats(B, C, G) floats.  All sums in a
 * fixed order.  nn.GroupNorm of backbone/feature_extractor.py:40-47, msdeformattn.py and the UNet's ResBlocks under run/train.py:504-540. */
int64_t xm3d_group_norm_bwd_ws_floats(int64_t B, int32_t C, int32_t G);
int xm3d_group_norm_bwd(const float* x, const float* dy, const double* stats, const float* gamma, const float* beta, int64_t B, int32_t C, int32_t hw,
                        int32_t G, float eps, int32_t act, float* dx, float* dgamma, float* dbeta, float* ws, void* stream);
/* out (B, H + pad_bottom, W + pad_right, C) <- zero-padded channels-last x (B, H, W, C); C a multiple of 4 (f32) / 8 (bf16).
 * The (0, 1, 0, 1) padding of ldm's VAE Downsample in one pass instead of F.pad's fill + strided copy. */
int xm3d_pad_nhwc(const void* x, int32_t dtype, int64_t B, int32_t H, int32_t W, int32_t C, int32_t pad_bottom, int32_t pad_right, void* out,
                  void* stream);
/* out = x * sigmoid(1.702 x) (QuickGELU of CLIP's MLPs, meta_arch/clip.py via open_clip), f32 / bf16 contiguous, numel a
 * multiple of 4 / 8; in place allowed. */
int xm3d_quick_gelu(const void* x, int32_t dtype, int64_t numel, void* out, void* stream);
/* probs[r, :] = softmax(scale * scores[r, :]): scores (rows, cols) f32 contiguous, probs (rows, cols) bf16; cols a multiple of
 * 4 up to 8192, scale > 0.  The softmax of the VAE's single-head 4096 x 512 attention between its two library GEMMs (ldm
 * AttnBlock; models/modeling/meta_arch/ldm.py:448-482) - see pointwise.hip. */
int xm3d_softmax_rows_f32_bf16(const float* scores, int64_t rows, int32_t cols, float scale, void* probs, void* stream);

/* blocked (maps, (S/P)^2) u8 = max_pool2d(sigmoid(bilinear_resize(logits (maps, h, w) f32 -> (S, S), align_corners = False)), P, stride P) < 0.5: the
 * patch mask of mask-CLIP (models/modeling/meta_arch/clip.py:272-310: F.interpolate + sigmoid + F.max_pool2d + compare) without the (maps, S, S)
 * intermediate.  S % P == 0. */
int xm3d_clip_mask_blocked(const float* logits, int64_t maps, int32_t h, int32_t w, int32_t S, int32_t P, uint8_t* blocked, void* stream);

/* ---- masked cross-attention bias (replaces the mask handling of Mask2Former's decoder, XMask3D copy
 * third_party/.../odise.py:395,445-491: bilinear shrink -> sigmoid -> < 0.5 -> repeat over heads -> all/and-not -> -inf fill).
 * logits (maps, H, W) mask logits (maps = B*Q), in_dtype/out_dtype 0 = f32, 1 = bf16; (H,W) -> (h,w) must be a shrink by
 * an even integer factor, h*w <= 8192.  out (maps, h*w): 0 where the query may attend, -inf where its predicted mask
 * is < 0.5; a query with no open position gets all zeros.  Bit-identical to the op chain it replaces. */
int xm3d_attn_mask_bias(const void* logits, int32_t in_dtype, int64_t maps, int32_t H, int32_t W, int32_t h, int32_t w, void* out,
                        int32_t out_dtype, void* stream);

/* ---- prediction heads over mask_features (maskhead.hip): forward_prediction_heads + the mask handling of the masked transformer
 * decoder's forward + MaskPooling, models/modeling/meta_arch/odise.py:395,445-491,509-547 (einsum "bqc,bchw->bqhw", bilinear shrink ->
 * sigmoid -> threshold -> empty-mask rule -> additive mask; hard mask pooling einsum "bchw,bqhw->bqc").
 *   xm3d_mask_logits_bias : mask_embed (B, Q, 256) bf16, mask_features (B, H*W, 256) bf16 (channels-last image) ->
 *       logits (B, Q, H, W) bf16 or NULL (not wanted: only the rows the bias needs are computed), and
 *       bias (B, Q, h*w) f32 (bias_dtype 0) / bf16 (1) = 0 / -inf additive attention mask for the (h, w) level, or NULL.
 *       Constraints: Q <= 64, mask_dim 256, (H, W) -> (h, w) a shrink by an even integer factor, W % 32 == 0, 32 % (W / w) == 0.
 *   xm3d_mask_pool : logits (B, Q, H*W) bf16 + mask_features -> pooled_partial (chunks, B, Q, 256) f32 = per pixel chunk the sum of the
 *       feature rows of the pixels with logit > 0 (sigmoid > 0.5), count_partial (chunks, B, Q) f32 = their number, chunks =
 *       xm3d_mask_pool_chunks(H*W); every entry is written.  MaskPooling = sum_chunks pooled / (sum_chunks count + 1e-8).
 *   Launched on `stream`, no host synchronisation. */
int xm3d_mask_logits_bias(const void* mask_embed, const void* mask_features, int64_t B, int32_t Q, int32_t C, int32_t H, int32_t W, void* logits,
                          int32_t h, int32_t w, void* bias, int32_t bias_dtype, void* stream);
int32_t xm3d_mask_pool_chunks(int64_t HW);
int xm3d_mask_pool(const void* logits, const void* mask_features, int64_t B, int32_t Q, int32_t C, int64_t HW, float* pooled_partial,
                   float* count_partial, void* stream);

/* ---- per-point class labels of the inference post-processing (pointclass.hip): run/infer.py:489-507 (base / novel gate) and
 * :556-612 (fused, 2D-only, 3D-only predictions: F.normalize -> @ text -> * logit_scale -> softmax -> geometric ensemble with the
 * open-vocabulary probabilities of the point's mask -> gate -> arg-max), one pass over the (Np, K) f32 features.
 *   x (rows, K) f32 with row stride ldx; row_index (Np) int64 or NULL: point p is computed from row row_index[p];
 *   text (C, K) f32 UNIT rows, C <= 32, K % 8 == 0, K <= 1024; scale: device f32 scalar or NULL;
 *   binary_pred (Np) int64 (!= 0: base-predicted point: novel classes are excluded, else base classes); base_mask / novel_mask (C) bool.
 *   mode 0: label = arg-max over the allowed classes of <x, t_c>                                   (2D-only, 3D-only labels)
 *   mode 1: p = softmax(scale * <x / |x|, t>); a point inside a mask (masks (Np, Q) bool, at most one set, its first set column q):
 *           value_c = log(p_c^r po_c^(1-r)), r = base_ratio where overlap[c] else novel_ratio, po = open_p[vid[p], q] ((B, Q, C));
 *           other points: value_c = p_c; label = arg-max over the allowed classes (first maximum; NaN maximal, as torch.argmax).
 *   label (Np) int64.  Launched on `stream`, no host synchronisation. */
int xm3d_point_class(const float* x, int64_t ldx, const int64_t* row_index, int64_t np, const float* text, int32_t C, int32_t K, const float* scale,
                     const int64_t* binary_pred, const uint8_t* base_mask, const uint8_t* novel_mask, int32_t mode, const uint8_t* masks, int32_t Q,
                     const int64_t* vid, const float* open_p, const float* overlap, float base_ratio, float novel_ratio, int64_t* label,
                     void* stream);

/* ---------------------------------------------------------------------------
 * Fused softmax attention forward, bf16 in / f32 softmax and accumulation / bf16 out (replaces the library attention behind
 * torch.nn.functional.scaled_dot_product_attention at the reference's call sites: ldm CrossAttention reached from
 * models/modeling/meta_arch/ldm.py:425-446, open_clip's ResidualAttentionBlock from clip.py:239-270, nn.MultiheadAttention of
 * mask2former_transformer_decoder.py:17-80):   out = softmax(q k^T * scale + bias) v   per (batch, head).
 *   q (B,Nq,H,D), k / v (B,Nk,H,D), out (B,Nq,H,D): bf16, channels contiguous, ELEMENT strides {batch, row, head} given per
 *   tensor (so (B,N,H*D) projections, (N,B,E) sequences and packed qkv buffers are consumed in place); D a multiple of 8, <= 160.
 *   bias: NULL / bias_dtype 0 = none; 1 = f32, 2 = bf16 additive term of shape (B,H,Nq,Nk) with element strides {batch, head,
 *   query row} (0 to broadcast), keys contiguous; values < -1e29 mask the key (a row with every key masked yields zeros).
 * ------------------------------------------------------------------------- */
int xm3d_attention_fwd(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t H, int32_t Nq, int32_t Nk,
                       int32_t D, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                       const int64_t* o_strides, const void* bias, int32_t bias_dtype, const int64_t* bias_strides,
                       float scale, void* stream);
/* The same forward, also writing the per-row log-sum-exp the backward needs: lse2 (B, H, Nq) f32 = log2(sum_k exp2(score * log2 e)),
 * i.e. in the log2 domain the kernel works in (1e30 for a row whose keys are all masked: its probabilities are zero). */
int xm3d_attention_fwd_lse(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t D,
                           const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides,
                           const void* bias, int32_t bias_dtype, const int64_t* bias_strides, float scale, float* lse2, void* stream);
/* The same attention to f32 ACCURACY on the 16-bit matrix cores (attention_f32.hip) - the fp32 configuration (the reference's own
 * arithmetic: f32 torch attention, run/train.py:178 - no autocast): q, k, v, out f32 with element strides as above (multiples of 4),
 * bias NULL or additive f32.  Both products run on operands split in IEEE halves (22 mantissa bits, three MFMAs per product, leading
 * and small terms in separate f32 accumulators), the softmax in f32; nothing of size Nq x Nk touches memory (torch's MATH path wrote
 * a 10.7 GB score tensor per 64^2 self attention of 20 views).  Head channels a multiple of 8 up to 64. */
int xm3d_attention_fwd_f32(const float* q, const float* k, const float* v, float* out, int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t D,
                           const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides,
                           const float* bias, const int64_t* bias_strides, float scale, void* stream);
/* Backward of the attention above (attention_bwd.hip): dq (B,Nq,H,D), dk / dv (B,Nk,H,D) bf16 CONTIGUOUS outputs, from q, k, v, the
 * forward's out and lse2, and dout (the gradient w.r.t. out; element strides like the other operands).  The additive bias is a
 * constant (no gradient).  delta_ws: B*H*Nq floats of scratch.  Replaces autograd through the reference's attention
 * (loss.backward(), run/train.py:537, through ldm's CrossAttention: models/modeling/meta_arch/ldm.py:425-446,670-676).  Flash-style
 * recomputation, two launches (dq + delta; dk, dv), no atomics: results are bit-reproducible. */
int xm3d_attention_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse2, int32_t B, int32_t H,
                       int32_t Nq, int32_t Nk, int32_t D, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                       const int64_t* o_strides, const int64_t* do_strides, const void* bias, int32_t bias_dtype, const int64_t* bias_strides,
                       float scale, void* dq, void* dk, void* dv, float* delta_ws, void* stream);

/* ---------------------------------------------------------------------------
 * Multi-scale deformable attention (replaces the pybind module
 * MultiScaleDeformableAttention: third_party/Mask2Former/mask2former/modeling/
 * pixel_decoder/ops/src/vision.cpp:18-21, ms_deform_attn.h:25-66,
 * cuda/ms_deform_attn_cuda.cu:25-157).  Same argument meaning and layout:
 *   value (B,S,H,D) f32, spatial_shapes (L,2) i64 [h,w], level_start (L) i64,
 *   loc (B,Lq,H,L,P,2) f32 (x,y in [0,1]), attn (B,Lq,H,L,P) f32.
 * forward : out (B,Lq,H*D) f32, fully overwritten.
 * backward: grad_value/grad_loc/grad_attn must be ZEROED by the caller
 *           (the reference allocates them with at::zeros); grad_value is
 *           accumulated with float atomics like the reference.
 * ------------------------------------------------------------------------- */
int xm3d_msda_forward(const float* value, const int64_t* spatial_shapes, const int64_t* level_start,
                      const float* loc, const float* attn, int32_t B, int32_t S, int32_t H, int32_t D,
                      int32_t L, int32_t Lq, int32_t P, float* out, void* stream);
int xm3d_msda_backward(const float* value, const int64_t* spatial_shapes, const int64_t* level_start,
                       const float* loc, const float* attn, const float* grad_out, int32_t B, int32_t S,
                       int32_t H, int32_t D, int32_t L, int32_t Lq, int32_t P, float* grad_value,
                       float* grad_loc, float* grad_attn, void* stream);
/* double instantiation: the reference dispatches the op over float AND double (AT_DISPATCH_FLOATING_TYPES,
 * cuda/ms_deform_attn_cuda.cu:64,134) and its own gradcheck runs in double (ops/test.py:66-81). */
int xm3d_msda_forward_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start,
                          const double* loc, const double* attn, int32_t B, int32_t S, int32_t H, int32_t D,
                          int32_t L, int32_t Lq, int32_t P, double* out, void* stream);
int xm3d_msda_backward_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start,
                           const double* loc, const double* attn, const double* grad_out, int32_t B, int32_t S,
                           int32_t H, int32_t D, int32_t L, int32_t Lq, int32_t P, double* grad_value,
                           double* grad_loc, double* grad_attn, void* stream);

/* ---------------------------------------------------------------------------
 * 2D->3D mask fusion epilogue (replaces the per-query Python loops of
 * models/xmask3d.py:421-451 and models/utils/fuser.py:24-35):
 *   masks (Q,Hm,Wm) u8 (0/1), x/y (n) i64 pixel row/col of each point, embed (Q,C) f32
 *   feat2d[p,:] = mean over queries q with masks[q,x[p],y[p]] of embed[q,:]  (0 if none)
 *   count[p]    = number of such queries
 * ------------------------------------------------------------------------- */
int xm3d_mask_point_fuse(const uint8_t* masks, int32_t Q, int32_t Hm, int32_t Wm, const int64_t* x,
                         const int64_t* y, int64_t n, const float* embed, int32_t C, float* feat2d,
                         int32_t* count, void* stream);
/* Pixel ownership among the mask queries (models/xmask3d.py:372-392): logits (B,Q,hw) f32 mask logits at mask_shape,
 * score (B,Q) f32, keep (B,Q) u8.  owner (B,hw) i32 = the first arg-max over q of (keep ? score : -1) * sigmoid(logit) if
 * that query is kept and its sigmoid >= 0.5 there, else -1.  The reference's per-query binary masks are owner == q. */
int xm3d_mask_owner(const float* logits, const float* score, const uint8_t* keep, int32_t B, int32_t Q, int64_t hw, int32_t* owner,
                    void* stream);

/* ---------------------------------------------------------------------------
 * Batched linear sum assignment (replaces cost.cpu() + scipy.optimize.linear_sum_assignment in the reference's matcher,
 * third_party/Mask2Former/mask2former/modeling/matcher.py:95-156, called ten times per training iteration):
 *   cost (n_mat, Q, T_max) f32 on the device, matrix m uses its first n_targets[m] columns (n_targets (n_mat) i32, device),
 *   1 <= Q <= 64, T_max <= 256.  Per matrix: min(Q, n_targets[m]) (query, target) pairs, every query and every target at most
 *   once, of minimum summed cost (scipy's rectangular semantics; f64 arithmetic like scipy; with tied costs one of the optima).
 *   out_q / out_t (n_mat, T_max) i64: the matched pairs sorted by ascending query index (scipy's order) in the first
 *   min(Q, n_targets[m]) slots; the remaining slots are left untouched (callers pre-fill them).
 * ------------------------------------------------------------------------- */
int xm3d_linear_sum_assignment(const float* cost, int64_t n_mat, int32_t Q, int32_t T_max, const int32_t* n_targets,
                               int64_t* out_q, int64_t* out_t, void* stream);

/* ---------------------------------------------------------------------------
 * Point cloud -> pixel mapping (replaces PointCloudToImageMapper.compute_mapping, models/utils/fusion_util.py:46-142, run per
 * view by the reference's data loaders): pts (n,3) f64 DEVICE; world_to_camera 4x4 and intrinsic 4x4 row-major f64 on the HOST
 * (the inverse of the pose is taken by the caller, like np.linalg.inv); image (width, height), cut_bound; depth NULL or a
 * (depth_h, depth_w) f64 DEVICE map with vis_thres.  mapping (n,3) i32 = [row, col, 1] of visible points, [0,0,0] otherwise.
 * ------------------------------------------------------------------------- */
int xm3d_compute_mapping(const double* pts, int64_t n, const double* world_to_camera, const double* intrinsic4,
                         int32_t width, int32_t height, int32_t cut_bound, const double* depth, int32_t depth_h,
                         int32_t depth_w, double vis_thres, int32_t* mapping, void* stream);

/* ---------------------------------------------------------------------------
 * Exact 1-nearest-neighbour index (replaces sklearn.neighbors.KDTree(...).query(k=1) in run/infer.py:523-553,
 * :682-694): query (n,3) f32, ref (m,3) f32, out (n) i64 = arg-min squared distance, lowest index on ties.
 * ref_valid (m) u8 or NULL: reference points with 0 are ignored (if no reference point is valid the result is 0).
 * counts (2) i64 DEVICE or NULL: {live queries, live references}: only the first counts[0] rows of query and the first
 * counts[1] rows of ref take part (out rows beyond counts[0] are left untouched) - lets the caller order "live" rows
 * first on the device and skip the rest without reading a count back to the host.
 * ws: NULL or 8*n bytes of device scratch; with it, few-query / many-reference problems are cut into reference slices
 * merged by a 64-bit atomicMin (same result, better occupancy of the device).
 * ------------------------------------------------------------------------- */
int xm3d_nearest_index(const float* query, int64_t n, const float* ref, int64_t m, const uint8_t* ref_valid,
                       const int64_t* counts, int64_t* out, void* ws, void* stream);
/* Segmented form (all views of a scene batch in one launch).  pts (n,3) f32 holds, per segment, its query points followed
 * by its reference points; desc (n_seg,4) int64 ON THE DEVICE = {q_off, q_cnt, r_off, r_cnt} per segment; max_queries = a
 * host-side upper bound of q_cnt (sizes the grid).  out[q_off+i] = r_off + index of the nearest reference point of the
 * same segment; entries of segments without reference points, and all non-query entries, are left untouched. */
int xm3d_nearest_index_segmented(const float* pts, const int64_t* desc, int32_t n_seg, int64_t max_queries, int64_t* out,
                                 void* stream);
/* Scene votes (run/infer.py:642-661: scene_pred[mask_2d, logits_pred] += 1 per view for the fused / 2D-only / 3D-only
 * predictions, counter[mask_2d] += 1; :690-694 torch.max(scene_pred, dim=1)) for all views of a group of scenes at once.
 * rows (n_points) i64: table row of every visible point (scene offset + point index); pred (n_kinds, n_points) i64 class ids;
 * votes (n_kinds, n_rows, n_cls) i32 scratch (zeroed here); label (n_kinds, n_rows) i64 <- first maximal class per row (0 for
 * rows without votes); seen (n_rows) u8 <- row received a vote of kind 0.  Out-of-range rows / classes set the sticky device
 * flag (xm3d_check_flag) and are skipped.  No host synchronisation. */
int xm3d_scene_votes(const int64_t* rows, const int64_t* pred, int32_t n_kinds, int64_t n_points, int64_t n_rows, int32_t n_cls,
                     int32_t* votes, int64_t* label, uint8_t* seen, void* stream);
/* "Every point without a value takes the nearest point that has one" in one call (run/infer.py:682-694: labels of never-seen
 * scene points from a KD-tree over the seen ones): xyz (n,3) f32, valid (n) u8; out[i] = i where valid, else the index of the
 * nearest valid point - same squared-f32 distance and lowest-index tie rule as xm3d_nearest_index, identity when nothing is
 * valid.  The valid points are binned into a 64^3 grid over their bounding box (cell edge >= `cell`) by a counting sort
 * on the cells' Morton codes; every query descends the implied octree nearest child first; no host synchronisation.  ws: xm3d_nearest_valid_fill_workspace_bytes(n) bytes. */
int64_t xm3d_nearest_valid_fill_workspace_bytes(int64_t n);
int xm3d_nearest_valid_fill(const float* xyz, int64_t n, const uint8_t* valid, float cell, int64_t* out, void* ws, void* stream);
/* The same contract, Morton-sorted and tile-pruned (nearest_sorted.hip): queries and valid points are sorted along a Morton
 * curve (one 31-bit radix sort), a wave of 64 neighbouring queries scans - LDS broadcast, like xm3d_nearest_index - only the
 * 64-point reference tiles whose bounding box is not farther from the queries' box than their worst best-distance.
 * For many queries far from the valid points (scene votes with large unseen regions).  ws: 256-byte aligned,
 * xm3d_nearest_valid_fill_sorted_workspace_bytes(n) bytes.  No host synchronisation. */
int64_t xm3d_nearest_valid_fill_sorted_workspace_bytes(int64_t n);
int xm3d_nearest_valid_fill_sorted(const float* xyz, int64_t n, const uint8_t* valid, int64_t* out, void* ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XM3D_H */

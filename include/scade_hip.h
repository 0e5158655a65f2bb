/* scade_hip.h -- C ABI of libscade_hip.so, the MI355X (gfx950) implementation of
 * SCADE's per-ray rendering hot path.
 *
 * The reference (mikacuy/scade) has NO native/FFI layer: its boundary for this
 * path is the Python operator API of model/run_nerf_helpers.py and
 * run_scade_scannet.py.  Each entry point below names the reference operator
 * (file:line, relative to the reference checkout) whose arithmetic it replaces;
 * scade_amd/ binds them with ctypes behind those same Python names.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 (row-major, contiguous unless a
 *    *_stride argument says otherwise; strides are in elements); the caller
 *    (PyTorch) owns all memory, the library never allocates or retains pointers;
 *  - `stream` is a hipStream_t; kernels are only enqueued, never synchronised;
 *  - return 0 on success, a negative argument-error code or a positive
 *    hipError_t otherwise; scade_last_error() returns the thread-local message;
 *  - optional pointers may be NULL where the comment says so.
 */
#ifndef SCADE_HIP_H
#define SCADE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

int scade_version(void);
const char* scade_last_error(void);

/* ---- NeRF MLP (model/run_nerf_helpers.py:193-247 NeRF, :131-139 DenseLayer) --- */

/* floats in the packed parameter blob consumed by scade_mlp_fwd */
long scade_mlp_packed_floats(void);
/* dynamic LDS bytes of the fused forward kernel (reported for occupancy checks) */
int scade_mlp_lds_bytes(void);

/* Re-layout the 24 parameter tensors of NeRF(D=8,W=256,input_ch=57,input_ch_views=3,
 * skips=[4],use_viewdirs=True) into MFMA A-fragment order.  params[] is a HOST
 * array of 24 DEVICE pointers in the order
 *   pts_linears.{0..7}.{weight,bias}, views_linears.0.{weight,bias},
 *   feature_linear.{weight,bias}, alpha_linear.{weight,bias}, rgb_linear.{weight,bias}
 * (nn.Linear [out,in] layout, helpers:205-220).  Re-run after every optimizer step. */
int scade_mlp_pack(const float* const* params, float* packed, void* stream);

/* Fused embed + MLP forward:  out[P,4] = [rgb(3, pre-sigmoid), softplus(alpha, beta=10)].
 *  mode 0: `in` is x[P,60] = [gamma(x)(57) | viewdir(3)]  == NeRF.forward(x), helpers:223-247
 *  mode 1: `in` is pts[P,3]; viewdirs[P/S] rows of 3 at stride vd_stride; bb = {cx,cy,cz,scale};
 *          the kernel applies
 *          (pts-bb_center)*bb_scale, the 9-frequency positional encoding and the
 *          per-ray view broadcast  == run_network(...), run_scade_scannet.py:48-63 with
 *          get_embedder(9,0)/get_embedder(0,0), helpers:142-189.
 *  acts (nullable): training workspace of scade_mlp_acts_floats(P) floats receiving every
 *          hidden layer's activations, the embedding, alpha and the ReLU sign words (consumed by
 *          scade_mlp_bwd; the same layout is written by scade_mlp_fwd_f16).
 *  Launches with P < 64 * 2 * CUs use 32-point workgroups (twice as many) instead of 64-point ones;
 *  results per point and the workspace layout do not depend on that choice. */
int scade_mlp_fwd(const float* packed, int mode, const float* in, const float* viewdirs,
                  int vd_stride, const float* bb, int P, int S, float* out, float* acts,
                  void* stream);

/* The weight packs of a train step for one or two networks in ONE launch (instead of scade_mlp_pack +
 * scade_mlp_pack_t, or scade_mlp_pack_lp + scade_mlp_pack_t_lp, per network).  params: n_nets x 24 pointers,
 * network-major, each network in the scade_mlp_pack order; format 0 = exact fp32 layouts, 1 = bf16, 2 = fp16
 * (the 16-bit forward blob includes the NaN census of the fp32 parameters); packed_fwd / packed_t: n_nets
 * output blobs each, sized as for the stand-alone entries, NULL entries are skipped.  Same bytes as the
 * stand-alone pack entries write. */
int scade_mlp_pack_step(int n_nets, const float* const* params, int format, void* const* packed_fwd,
                        void* const* packed_t, void* stream);
/* The same for the split-precision ("f16x3") training kernels: per network the scade_mlp_pack blob (the dgrad's
 * heads read the fp32 head weights from it), the scade_mlp_pack_f16 blob and the scade_mlp_pack_t_f16 blob - six
 * stand-alone launches per step otherwise.  n_nets output pointers each, NULL entries are skipped; same bytes as
 * the stand-alone entries write.  (Host-side mirror of what the reference does implicitly: its nn.Linear weights
 * ARE the operands, model/run_nerf_helpers.py:136-153; here the MFMA kernels read re-ordered copies.) */
int scade_mlp_pack_step_f16x3(int n_nets, const float* const* params, float* const* packed_exact,
                              void* const* packed_f16, void* const* packed_t_f16, void* stream);

/* Backward of scade_mlp_fwd w.r.t. the 24 parameter tensors (what autograd computes for
 * NeRF.forward in the reference).  packed_t = scade_mlp_pack_t(params) (transposed weight
 * pack, scade_mlp_packed_t_floats() floats); acts = the workspace the forward filled;
 * g_out[P,4] = dL/d out; workspace = scade_mlp_bwd_workspace_floats(P) floats;
 * grad_flat[589700] receives the gradients concatenated in the scade_mlp_pack order, each
 * tensor in its nn.Linear layout. */
long scade_mlp_acts_floats(long P);
long scade_mlp_packed_t_floats(void);
int scade_mlp_pack_t(const float* const* params, float* packed_t, void* stream);
long scade_mlp_bwd_workspace_floats(int P);
int scade_mlp_bwd_chunks(int P);
int scade_mlp_bwd(const float* packed, const float* packed_t, const float* acts,
                  const float* g_out, int P, float* workspace, float* grad_flat, void* stream);

/* The backward of TWO network calls in one go - the coarse and the fine NeRF of a train step, whose backward
 * chains are independent (run_scade_scannet.py:711 detaches the samples between them): ONE dgrad launch, ONE
 * weight-gradient launch and ONE reduce over both instead of three launches each.  Every argument is a HOST
 * array of two entries with the meaning it has in scade_mlp_bwd; workspace[i] must hold
 * scade_mlp_bwd2_workspace_floats(P[i], P[1-i]) floats (never less than scade_mlp_bwd_workspace_floats(P[i])).
 * Same result per network as two scade_mlp_bwd calls up to the summation order over point chunks (the joint
 * launch picks one chunk length for both). */
long scade_mlp_bwd2_workspace_floats(int P, int P_other);
int scade_mlp_bwd2(const float* const* packed, const float* const* packed_t, const float* const* acts,
                   const float* const* g_out, const int* P, float* const* workspace, float* const* grad_flat,
                   void* stream);
/* The same backward in phases - phases is a mask: bit 0 = the joint dgrad chain of both networks, bit 1 / bit 2 =
 * weight gradient + reduce of network 0 / 1 (both = ONE joint weight-gradient launch: scade_mlp_bwd2 = all three).
 * A ray-sharded step (replaces nn.DataParallel, run_scade_scannet.py:438,455) calls 1, then 2, starts the all-reduce of
 * network 0's gradient behind it, and calls 4: the exchange runs under network 1's weight gradient.  Same bits per
 * network as the joint launch. */
int scade_mlp_bwd2_phases(const float* const* packed, const float* const* packed_t, const float* const* acts,
                          const float* const* g_out, const int* P, float* const* workspace, float* const* grad_flat,
                          int phases, void* stream);

/* Split-precision INFERENCE variant of scade_mlp_fwd (opt-in): every fp32 value is
 * carried as two fp16 numbers x ~= h + l*2^-11 and every product as three f16 MFMAs into two fp32
 * accumulators (~2^-21 relative error per product; requires |activations|,|weights| < 65504).
 * packed_f16 = scade_mlp_pack_f16(params), scade_mlp_packed_f16_bytes() bytes.  Same modes,
 * arguments, output and (optional) training workspace as scade_mlp_fwd.  mode + 2 (with acts): the saved rows are
 * written in the 3-byte split form scade_mlp_bwd_f16(wgrad_f16 = 1) reads (mlp_tile_f16.h r24_store4 / l8_of4,
 * mlp_wgrad.h R24): of the kernel's own split x ~= h + l * 2^-11, the fp16 h plane [P][256] at byte 0 of the row
 * slot and the l plane rounded to one e5m2 byte per value, [P][256] at byte P * 512 - x is recovered as
 * h + l8 * 2^-11 with a relative error of about 2^-14 (14 significant bits); 768 of the slot's 1024 bytes per point
 * (the weight gradient streams these rows at the memory system's rate).  The dZ rows scade_mlp_bwd_f16 stores for
 * its weight gradient use the same two planes in the per-point scaled domain, the factors 1 / s_p as fp32 [P] at
 * byte P * 768 of slot 0.  Without + 2: fp32 rows, for scade_mlp_bwd_f16(wgrad_f16 = 0) / scade_mlp_bwd. */
long scade_mlp_packed_f16_bytes(void);
int scade_mlp_pack_f16(const float* const* params, void* packed_f16, void* stream);
int scade_mlp_fwd_f16(const void* packed_f16, int mode, const float* in, const float* viewdirs,
                      int vd_stride, const float* bb, int P, int S, float* out, float* acts,
                      void* stream);

/* Single-plane 16-bit INFERENCE variant of scade_mlp_fwd (opt-in; BASELINE.json config 5's "bf16 MFMA
 * path"): activations and weights rounded to fp16 (bf16 = 0) or bfloat16
 * (bf16 = 1), one v_mfma_f32_32x32x16_{f16,bf16} per product, fp32 accumulate, fp32 biases, heads and
 * outputs.  Ordinary mixed-precision accuracy (relative ~1e-3 fp16 / ~1e-2 bf16), NOT the 1e-4 parity
 * bar.  The fp16 variant needs |activations|, |weights| < 65504; bfloat16 has fp32's range.  packed_lp = scade_mlp_pack_lp(params, bf16), scade_mlp_packed_lp_bytes() bytes (the format is
 * baked into the pack: pass the same bf16 flag to both).  Same modes and arguments as scade_mlp_fwd;
 * acts (nullable) = training workspace of scade_mlp_acts_lp_bytes(P) bytes: 16-bit activations, the
 * embedding rows, fp32 alpha_pre and the ReLU sign words consumed by scade_mlp_bwd_lp. */
long scade_mlp_packed_lp_bytes(void);
long scade_mlp_acts_lp_bytes(long P);
int scade_mlp_pack_lp(const float* const* params, void* packed_lp, int bf16, void* stream);
int scade_mlp_fwd_lp(const void* packed_lp, int bf16, int mode, const float* in, const float* viewdirs,
                     int vd_stride, const float* bb, int P, int S, float* out, void* acts, void* stream);

/* Mixed-precision backward of scade_mlp_fwd_lp (same mathematics as scade_mlp_bwd): 16-bit dgrad chain
 * with a per-point power-of-two gradient scale, every layer's dZ stored as 16-bit rows under ONE
 * launch-wide power-of-two loss scale (from max|g_out|, removed exactly from the result), weight
 * gradient on 16-bit MFMAs with fp32 accumulation.  packed: not read, may be NULL (the fp32 head
 * weights travel in the tail of packed_t_lp, so a 16-bit training step packs no fp32 blob);
 * packed_t_lp = scade_mlp_pack_t_lp(params, bf16); workspace = scade_mlp_bwd_lp_workspace_bytes(P)
 * bytes; grad_flat[589700] fp32 as in scade_mlp_bwd. */
long scade_mlp_packed_t_lp_bytes(void);
int scade_mlp_pack_t_lp(const float* const* params, void* packed_t_lp, int bf16, void* stream);
long scade_mlp_bwd_lp_workspace_bytes(int P);
int scade_mlp_bwd_lp(const float* packed, const void* packed_t_lp, int bf16, const void* acts,
                     const float* g_out, int P, void* workspace, float* grad_flat, void* stream);

/* Two network calls in one dgrad / weight-gradient / reduce launch each (see scade_mlp_bwd2): host arrays of
 * two entries; workspace[i] = scade_mlp_bwd_lp2_workspace_bytes(P[i], P[1-i]) bytes.  The ReLU sign words of
 * the 16-bit workspace are indexed by the workgroups of the forward launch that wrote them, so both launches
 * must have used the same point tiling: scade_mlp_lp_point_tiles(P[0]) == scade_mlp_lp_point_tiles(P[1])
 * (otherwise error -3: call scade_mlp_bwd_lp twice). */
int scade_mlp_lp_point_tiles(int P);
/* The launch plan of the 16-bit weight gradient (scade_mlp_bwd_lp / _lp2) for networks of P[0], P[1] (0 = absent)
 * points, host code only (tests, tools): the work is the list of (network, job) entries, each a run of 32-point
 * stages weighted by the job's measured per-stage cost; workgroup w owns the weighted positions [bound[w],
 * bound[w + 1]) (one workgroup per CU), segment k of an entry writes partial row k.  info[7] = {workgroups, jobs per
 * network, points per stage, partial rows per workspace, chunk, gx0, gx1} (chunk > 0: launches below 100k points
 * use a (job, chunk) grid instead); bound[257], cum[33], first_wg[32], nseg[32], weight[16]. */
int scade_mlp_wgrad_lp_plan(const int* P, int s8, int* info, int* bound, int* cum, int* first_wg, int* nseg,
                            int* weight);
long scade_mlp_bwd_lp2_workspace_bytes(int P, int P_other);
int scade_mlp_bwd_lp2(const void* const* packed_t_lp, int bf16, const void* const* acts,
                      const float* const* g_out, const int* P, void* const* workspace,
                      float* const* grad_flat, void* stream);
/* scade_mlp_bwd_lp2 in phases (see scade_mlp_bwd2_phases; bit 0 also takes the loss-scale maxima). */
int scade_mlp_bwd_lp2_phases(const void* const* packed_t_lp, int bf16, const void* const* acts,
                             const float* const* g_out, const int* P, void* const* workspace,
                             float* const* grad_flat, int phases, void* stream);

/* Split-precision variant of scade_mlp_bwd (opt-in training mode): the dgrad chain runs on
 * f16 MFMAs with a per-point power-of-two gradient scale (exactly removed on store); the weight
 * gradient runs on the exact fp32 kernel (wgrad_f16 = 0) or on f16 MFMAs with one power-of-two
 * scale per launch (wgrad_f16 = 1: acts must hold the 24-bit rows of scade_mlp_fwd_f16(mode + 2), and the dZ rows
 * of the workspace are written and read in the same form).  packed = the fp32 forward pack (head weights);
 * packed_t_f16 = scade_mlp_pack_t_f16(params); other arguments as scade_mlp_bwd. */
long scade_mlp_packed_t_f16_bytes(void);
int scade_mlp_pack_t_f16(const float* const* params, void* packed_t_f16, void* stream);
int scade_mlp_bwd_f16(const float* packed, const void* packed_t_f16, const float* acts,
                      const float* g_out, int P, int wgrad_f16, float* workspace, float* grad_flat,
                      void* stream);
/* The same for TWO network calls (the coarse + fine NeRF of a train step, run_scade_scannet.py:976 backward of
 * both) in one launch sequence - one dgrad launch, one weight-gradient launch, one reduce: every pointer argument is
 * a HOST array of two, workspace[i] sized by scade_mlp_bwd_workspace_floats(P[i]).  Same bits as two
 * scade_mlp_bwd_f16 calls. */
int scade_mlp_bwd_f16_2(const float* const* packed, const void* const* packed_t_f16, const float* const* acts,
                        const float* const* g_out, const int* P, int wgrad_f16, float* const* workspace,
                        float* const* grad_flat, void* stream);

/* ---- positional encoding (Embedder.embed, helpers:142-172; get_embedder :174-189) */
/* out[P, D*(1+2*multires)] = [x, sin(x*pi*2^0), cos(x*pi*2^0), ..., cos(x*pi*2^(L-1))] */
int scade_embed(const float* x, int P, int D, int multires, float* out, void* stream);

/* ---- ray sampling (run_scade_scannet.py:638-657, perturb_z_vals :564-579) ------ */
/* z_vals[N,S] = near*(1-t)+far*t (or the lindisp form), optional stratified jitter with
 * t_rand[N,S] (NULL == perturb 0), and pts[N,S,3] = o + d*z (pts nullable).
 * rays rows: o(0..2) d(3..5) near(6) far(7); t_vals = linspace(0,1,S). */
int scade_ray_points(const float* rays, int ray_stride, const float* t_vals, const float* t_rand,
                     int N, int S, int lindisp, float* z_vals, float* pts, void* stream);
/* The same with the training step's uniform draws made INSIDE the kernel (the reference draws them with three
 * torch.rand calls per step: run_scade_scannet.py:570, helpers:350, helpers:399): Philox4x32-10 keyed by `seed`,
 * counter = (draw block, ray, step) - a pure function of (seed, step, ray, draw index), no generator state.
 * step = *step_dev (a float holding the number of steps taken, e.g. the fused optimizer's device state: graph
 * captured steps) when step_dev is given, else `step`.  Per ray the stream is [S jitter draws | Si draws of the
 * coarse importance sampler | Si draws of the depth-hypothesis sampler], each part padded to a multiple of four;
 * a draw is (32 random bits >> 8) * 2^-24 in [0,1).  The jitter is consumed in registers; u_a / u_b [N,Si]
 * (nullable) receive the samplers' draws for scade_ray_tail / scade_sample_pdf_fwd. */
int scade_ray_points_draw(const float* rays, int ray_stride, const float* t_vals, int N, int S, int lindisp,
                          unsigned long long seed, unsigned long long step, const float* step_dev, int Si,
                          float* z_vals, float* pts, float* u_a, float* u_b, void* stream);

/* stratified jitter of an existing z tensor (perturb_z_vals :564-579), t_rand[N,S] ~ U[0,1) */
int scade_perturb_z(const float* z_vals, const float* t_rand, int N, int S, float* out,
                    void* stream);

/* ---- alpha compositing (compute_weights :511-522, raw2outputs :530-562) -------- */
/* noise (nullable) is the pre-drawn sigma noise [N,S] (raw_noise_std * randn). */
int scade_composite_fwd(const float* raw, const float* z_vals, const float* rays_d, int d_stride,
                        const float* noise, int N, int S, float* rgb_map, float* disp_map,
                        float* acc_map, float* weights, float* depth_map, void* stream);
/* gradient w.r.t. raw[N,S,4]; each g_* may be NULL (== zero). */
int scade_composite_bwd(const float* raw, const float* z_vals, const float* rays_d, int d_stride,
                        const float* noise, int N, int S, const float* g_rgb, const float* g_disp,
                        const float* g_acc, const float* g_weights, const float* g_depth,
                        float* g_raw, void* stream);

/* ---- inverse-CDF sampler (helpers:337-383 sample_pdf, :385-436 sample_pdf_return_u,
 *      :439-538 joint variants share the kernel with u_stride = 0) ------------------ */
/* bins: M values per ray (bins_are_mids = 0) or M+1 z values whose midpoints are the
 * bins (bins_are_mids = 1, run_scade_scannet.py:702/:723).  weights: M-1 values per ray
 * at weights[n*w_stride + j] (a strided view such as weights[...,1:-1] is passed as
 * base+1 with the parent row stride).  cdf_in (nullable): use this [N,M] cdf instead of
 * building it from weights (bit-exact index test).  u: S draws per ray, u_stride = 0
 * broadcasts one row.  Optional outputs: inds[N,S] (int64, searchsorted right=True),
 * cdf_out[N,M], z_std[N] (std of the samples, unbiased=False, :744). */
int scade_sample_pdf_fwd(const float* bins, int bins_stride, int bins_are_mids,
                         const float* weights, int w_stride, const float* cdf_in, const float* u,
                         int u_stride, int N, int M, int S, float* samples, long long* inds,
                         float* cdf_out, float* z_std, void* stream);
/* gradient w.r.t. the M-1 weights (dense [N,M-1] output). */
int scade_sample_pdf_bwd(const float* bins, int bins_stride, int bins_are_mids,
                         const float* weights, int w_stride, const float* u, int u_stride,
                         const float* g_samples, int N, int M, int S, float* g_weights,
                         void* stream);

/* ---- coarse+fine merge (run_scade_scannet.py:713-714) --------------------------- */
/* z_out[N,Sa+Sb] = torch.sort(cat(z_a, z_b)).values (neither input needs to be sorted; NaN last;
 * -0 before +0); pts (nullable) = o + d*z_out from rays rows.  Sa + Sb <= 4096; an operand with
 * S == 0 may be a null pointer. */
int scade_merge_sorted(const float* z_a, int Sa, const float* z_b, int Sb, const float* rays,
                       int ray_stride, int N, float* z_out, float* pts, void* stream);

/* ---- everything between two MLP launches of a ray batch, in one launch ----------------
 * (run_scade_scannet.py:660-714 for the coarse stage, :720-744 for the fine stage)
 * raw2outputs(raw[N,S,4], z_vals[N,S], rays_d) -> rgb_map, disp_map, acc_map, weights, depth_map;
 * samples[N,Si] = sample_pdf(z_mid, weights[:,1:-1], u)  (u as in scade_sample_pdf_fwd; nullable
 * output), z_std[N] (nullable) = std of the samples; and, when z_out is given (coarse stage),
 * z_out[N,S+Si] = sort(cat(z_vals, samples)), pts (nullable) = o + d*z_out.  rays: the [N,
 * ray_stride] rows (o at 0..2, d at 3..5).  Bit-identical to scade_composite_fwd ->
 * scade_sample_pdf_fwd(bins_are_mids=1) -> scade_merge_sorted on the same inputs: the weights and
 * the samples stay in LDS instead of making two round trips through HBM.  3 <= S <= 512 (with
 * z_out: S <= 256 and S+Si <= 512; larger rows: use the three separate entries). */
int scade_ray_tail(const float* raw, const float* z_vals, const float* rays, int ray_stride,
                   const float* noise, int N, int S, const float* u, int u_stride, int Si,
                   float* rgb_map, float* disp_map, float* acc_map, float* weights, float* depth_map,
                   float* samples, float* z_std, float* z_out, float* pts, void* stream);
/* Backward of the fine form of scade_ray_tail (z_out = NULL): d loss / d raw [N,S,4] from the gradients of
 * rgb_map / disp_map / acc_map / weights / depth_map (each may be NULL) and of the drawn samples [N,Si],
 * one launch = scade_sample_pdf_bwd on weights[1:-1] followed by scade_composite_bwd, bit for bit. */
int scade_ray_tail_bwd(const float* raw, const float* z_vals, const float* rays, int ray_stride,
                       const float* noise, int N, int S, const float* u, int u_stride, int Si,
                       const float* g_rgb, const float* g_disp, const float* g_acc, const float* g_weights,
                       const float* g_depth, const float* g_samples, float* g_raw, void* stream);

/* ---- space-carving loss (helpers:93-128) ---------------------------------------- */
/* pred[N,P]; hyp[K,N] (the reference's [K,N,1], hypothesis-major); mask[N] nullable;
 * threshold <= 0 disables; is_joint selects helpers:115-119.  workspace holds
 * scade_carve_workspace_floats() floats and must be passed unchanged to the backward. */
long scade_carve_workspace_floats(int N, int P, int K, int is_joint);
int scade_carve_fwd(const float* pred, const float* hyp, const float* mask, float threshold,
                    int is_joint, int N, int P, int K, float* workspace, float* loss,
                    void* stream);
int scade_carve_bwd(const float* pred, const float* hyp, const float* mask, float threshold,
                    int is_joint, int N, int P, int K, const float* workspace,
                    const float* g_loss, float* g_pred, float* g_hyp, void* stream);
/* The joint forward in two phases, for a ray-sharded job (run_nerf_helpers.py:115-119 takes the mean
 * over ALL rays before the min over K): colmean writes workspace[K*P] = mean over this shard's N rays;
 * the caller combines the shards (all-reduce of N_shard/N_total-weighted means); min takes the min over
 * K / mean over samples of the combined means and stores the argmin after them, as scade_carve_fwd
 * does, so scade_carve_bwd(is_joint = 1) can follow with g_loss pre-multiplied by N_shard/N_total. */
int scade_carve_joint_colmean(const float* pred, const float* hyp, const float* mask, float threshold,
                              int N, int P, int K, float* workspace, void* stream);
int scade_carve_joint_min(float* workspace, int P, int K, float* loss, void* stream);
/* The same entries for hypotheses cached PER SAMPLE, hyp [K,N,P] - the "each quantile here already picked
 * a hypothesis" branch of compute_space_carving_loss (run_nerf_helpers.py:100-102); g_hyp is [K,N,P].
 * Workspace sizes and scade_carve_joint_min are shared with the [K,N] entries. */
int scade_carve_knp_fwd(const float* pred, const float* hyp, const float* mask, float threshold,
                        int is_joint, int N, int P, int K, float* workspace, float* loss, void* stream);
int scade_carve_knp_bwd(const float* pred, const float* hyp, const float* mask, float threshold,
                        int is_joint, int N, int P, int K, const float* workspace, const float* g_loss,
                        float* g_pred, float* g_hyp, void* stream);
int scade_carve_knp_joint_colmean(const float* pred, const float* hyp, const float* mask, float threshold,
                                  int N, int P, int K, float* workspace, void* stream);

/* ---- photometric loss (helpers:11 img2mse; row mask = run_scade_wild.py:978-986) - */
int scade_mse_fwd(const float* x, const float* y, const float* row_mask, int n, int c,
                  float* loss, void* stream);
int scade_mse_bwd(const float* x, const float* y, const float* row_mask, int n, int c,
                  const float* g_loss, float* g_x, void* stream);

/* ---- the train loop's three-term loss in one forward / one backward entry (run_scade_scannet.py:954,
 *      :968-983; the wild variant's masked photometric terms run_scade_wild.py:977-1008):
 *          target_h = hyp * scales[img] + shifts[img]
 *          loss = mse(rgb, target) + carve_weight * space_carving(pred, target_h) + mse(rgb0, target)
 *      (non-joint space carving, hyp [K,N]).  img = img_i, or - when the device pointer img_i_dev is given
 *      (graph captured steps) - *img_i_dev (int64), with img_i = n_images as its bound: a device index outside
 *      [0, n_images) makes the loss NaN and writes no scale / shift gradient (never out of bounds).  mask [N] or NULL multiplies the carving distances, and the squared
 *      errors too when mse_masked.  carve_on = 0 drops the middle term (warm start, :973).  out_scale
 *      multiplies the total (a rank's share of a ray-sharded batch).  loss4 = {total, img_loss, carve,
 *      img_loss0}; workspace [4 N] floats.  The backward writes g_rgb / g_rgb0 [N,3], g_pred [N,P] and ADDS
 *      the scale / shift gradients into g_scales[img] / g_shifts[img]. */
int scade_train_loss_fwd(const float* rgb, const float* rgb0, const float* target, const float* pred,
                         const float* hyp, const float* scales, const float* shifts,
                         const long long* img_i_dev, int img_i, const float* mask, int mse_masked, int carve_on,
                         float carve_weight, float threshold, float out_scale, int N, int P, int K,
                         float* workspace, float* loss4, void* stream);
int scade_train_loss_bwd(const float* rgb, const float* rgb0, const float* target, const float* pred,
                         const float* hyp, const float* scales, const float* shifts,
                         const long long* img_i_dev, int img_i, const float* mask, int mse_masked, int carve_on,
                         float carve_weight, float threshold, float out_scale, int N, int P, int K,
                         float* workspace, const float* g_loss, float* g_rgb, float* g_rgb0, float* g_pred,
                         float* g_scales, float* g_shifts, void* stream);
/* Forward AND backward of that loss in ONE pair of launches, for callers that differentiate the total with a
 * UNIT gradient (a train step does; nothing in the backward depends on the reduced loss value): the arguments of
 * scade_train_loss_fwd + the gradient outputs of scade_train_loss_bwd; workspace [8 N] floats.  n_ss > 0: g_scales /
 * g_shifts [n_ss] are WRITTEN (zero but for the batch's image - the caller needs no zero fill of those rows);
 * n_ss = 0: the batch's image row is accumulated into, as scade_train_loss_bwd does. */
int scade_train_loss_fb(const float* rgb, const float* rgb0, const float* target, const float* pred,
                        const float* hyp, const float* scales, const float* shifts,
                        const long long* img_i_dev, int img_i, const float* mask, int mse_masked, int carve_on,
                        float carve_weight, float threshold, float out_scale, int N, int P, int K,
                        float* workspace, float* loss4, float* g_rgb, float* g_rgb0, float* g_pred,
                        float* g_scales, float* g_shifts, int n_ss, void* stream);

/* ---- ray generation + training-batch gather (helpers:285-305 get_ray_dirs/get_rays; the ray
 *      rows of render()/render_hyp(), run_scade_scannet.py:122-141; the gathers of
 *      get_ray_batch_from_one_image_hypothesis_idx, :784-821) ------------------------------- */
/* coords[N,2] int32 (row, col) or NULL for every pixel row-major (N = H*W); intrinsic = fx fy cx
 * cy; c2w rows 0..2 of [R|t].  Any output may be NULL.  rays rows: o d near far viewdir.
 * target_s gathers image[H,W,3]; target_h[K,N] gathers hyps[K,H,W]; mask[N] is 0 inside the
 * corner_px corner squares (--mask_corners) / the edge_px border (wild --mask_edges). */
int scade_gen_rays(const int* coords, int N, int H, int W, const float* intrinsic, const float* c2w,
                   int c2w_stride, float near, float far, const float* image, const float* hyps,
                   int K, int corner_px, int edge_px, float* rays, float* rays_o, float* rays_d,
                   float* target_s, float* target_h, float* mask, void* stream);

/* The training loop's per-iteration batch assembly (run_scade_scannet.py:946 image pick, :786 pixel pick,
 * get_ray_batch_from_one_image_hypothesis_idx :772-827, ray rows :200-219) as the ONE launch in front of a
 * graph-captured train step.  pix[N] int64: flat pixel indices row * W + col, every one in [0, H*W) (the caller's
 * contract - e.g. a slice of a device-resident permutation of the pixels); intrinsic / c2w / image [H,W,3] /
 * hyps [K,H,W] are those of the step's training view.  Outputs as scade_gen_rays (same arithmetic, same bits):
 * rays [N,11], target_s [N,3], target_h [K,N], mask [N]; any may be NULL.  scalar_dst != NULL: one 8-byte store
 * (the view's index, :951-954); tick_states != NULL: up to two device-resident optimizer states advanced by one
 * step, exactly as scade_stage_inputs does (pass ticked = 1 to scade_adam_step2 then). */
int scade_gather_batch(const long long* pix, int N, int H, int W, const float* intrinsic, const float* c2w,
                       int c2w_stride, float near, float far, const float* image, const float* hyps, int K,
                       int corner_px, int edge_px, float* rays, float* target_s, float* target_h, float* mask,
                       long long* scalar_dst, long long scalar, float* const* tick_states, void* stream);

/* ---- optimizer step (torch.optim.Adam defaults, run_scade_scannet.py:469, :888, :993-997) -- */
/* One launch over a flat buffer holding every trainable tensor; grads are multiplied by
 * grad_scale first (1/world_size after a sum all-reduce).  step counts from 1. */
int scade_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n,
                    float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                    void* stream);
/* The same update with every per-step quantity on the device (for train steps captured in a HIP
 * graph): state = float[16] {t, lr0, decay_rate, decay_step, beta1, beta2, eps, grad_scale, ...; [13] = t before
 * the last tick = the index of the step in flight (scade_ray_points_draw's step_dev)};
 * each call first advances t and derives the staircase learning rate (hyperparameter_update.py:8-13)
 * and the bias corrections on the device, then applies the update. */
int scade_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n,
                        float* state, void* stream);
/* Both optimizers of a train step - the networks' Adam and the depth scale / shift Adam (run_scade_scannet.py:469,
 * :888, :993-997), which differ in learning rate and step count - in ONE update launch.  Host arrays of two
 * entries; n[1] = 0 skips the second segment (frozen scale / shift, :996).  state[i] != NULL: that segment's
 * scalars are device resident (scade_adam_step_dev's state layout) and one tick launch advances both first
 * (ticked != 0: the caller already advanced them, scade_stage_inputs);
 * otherwise lr / beta1 / beta2 / eps / step / grad_scale [i] are read from the host arrays. */
int scade_adam_step2(float* const* params, const float* const* grads, float* const* exp_avg,
                     float* const* exp_avg_sq, const long* n, const float* lr, const float* beta1,
                     const float* beta2, const float* eps, const int* step, const float* grad_scale,
                     float* const* state, int ticked, void* stream);

/* scade_ray_tail_train that also leaves the loss-scale maxima of the two output gradients it writes, for the 16-bit
 * MLP backward of the same step (scade_mlp_bwd_lp2_deferred's gmax_pre: its own maxima launch is then skipped).
 * gmax_ws: [2 N] floats of workspace (per-ray maxima, reduced by the loss's one-workgroup reduce); gmax_fine /
 * gmax_coarse: [256] floats each - slot 0 = the maximum over the finite entries of the EFFECTIVE gradient of g_raw /
 * g_raw0 (colour channels as they are, the density channel times 1 - exp(-10 sigma) = the softplus' derivative
 * sigmoid(10 alpha_pre), model/run_nerf_helpers.py:242), the rest zero.  Needs the coarse ray; no raw noise. */
int scade_ray_tail_train_gmax(const float* raw, const float* z_vals, const float* rays, int ray_stride, int N, int S,
                              const float* u, int u_stride, int Si, float* rgb_map, float* disp_map, float* acc_map,
                              float* weights, float* depth_map, float* samples, float* z_std, const float* rgb0,
                              const float* target, const float* hyp, const float* scales, const float* shifts,
                              const long long* img_i_dev, int img_i, const float* mask, int mse_masked, int carve_on,
                              float carve_weight, float threshold, float out_scale, int K, float* workspace,
                              float* loss4, float* g_scales, float* g_shifts, int n_ss, float* g_raw,
                              const float* raw0, const float* z0, int S0, float* g_raw0, float* gmax_ws,
                              float* gmax_fine, float* gmax_coarse, void* stream);

/* ---- the launch that OPENS a graph-captured step also runs its first per-ray kernel and its weight packs (round 6) --
 * scade_stage_inputs / scade_gather_batch + scade_ray_points_draw (run_scade_scannet.py:638-657, :564-579) +
 * scade_mlp_pack_step / scade_mlp_pack_step_f16x3 as ONE launch, so that the captured step starts at its first MLP
 * launch:
 *  - extra workgroups of four waves = four rays compute z_vals [N,S], the coarse sample positions pts [N,S,3] and the
 *    step's sampler draws u_a / u_b [N,Si] (nullable) into the captured step's static buffers - from the SOURCE ray
 *    rows being staged (scade_stage_inputs_points: rays [N, ray_stride]; N = 0: no points) or from the ray rows they
 *    derive themselves from the pixel ids (scade_gather_batch_points).  ``step`` is the host's count of optimizer steps
 *    taken so far (what scade_ray_points_draw reads from the device state inside a captured step: these launches run
 *    outside it);
 *  - further workgroups re-pack the networks' weight blobs IN PLACE from the parameters the previous step's optimizer
 *    launch left: pack_format -1 none, 0 exact (packed_exact = scade_mlp_pack layout, packed_t = scade_mlp_pack_t),
 *    1 bf16 / 2 fp16 (packed_fwd = scade_mlp_pack_lp, packed_t = scade_mlp_pack_t_lp), 3 split precision
 *    (packed_exact, packed_fwd = scade_mlp_pack_f16, packed_t = scade_mlp_pack_t_f16); host arrays of n_nets device
 *    pointers, entries may be NULL (skipped); net_params: n_nets x 24 parameter pointers.
 * Same bits as the separate launches. */
int scade_stage_inputs_points(const void* const* src, void* const* dst, const long* bytes, int n,
                              long long* scalar_dst, long long scalar, float* const* tick_states, const float* rays,
                              int ray_stride, const float* t_vals, int N, int S, int lindisp,
                              unsigned long long seed, unsigned long long step, int Si, float* z_vals, float* pts,
                              float* u_a, float* u_b, int pack_format, int n_nets, const float* const* net_params,
                              float* const* packed_exact, void* const* packed_fwd, void* const* packed_t,
                              void* stream);
int scade_gather_batch_points(const long long* pix, int N, int H, int W, const float* intrinsic, const float* c2w,
                              int c2w_stride, float near, float far, const float* image, const float* hyps, int K,
                              int corner_px, int edge_px, float* rays, float* target_s, float* target_h, float* mask,
                              long long* scalar_dst, long long scalar, float* const* tick_states,
                              const float* t_vals, int S, int lindisp, unsigned long long seed,
                              unsigned long long step, int Si, float* z_vals, float* pts, float* u_a, float* u_b,
                              int pack_format, int n_nets, const float* const* net_params,
                              float* const* packed_exact, void* const* packed_fwd, void* const* packed_t,
                              void* stream);

/* ---- the weight gradient's reduce inside the optimizer's launch (round 6) ----------------------------------------
 * run_scade_scannet.py:985-997: loss.backward() ends in the weight gradient's sum over its partial rows, then
 * optimizer.step() / optimizer_ss.step() - two launches over the same 1.18 M floats (reduce, scade_adam_step2).
 *
 * scade_mlp_bwd2_deferred / scade_mlp_bwd_lp2_deferred / scade_mlp_bwd_f16_2_deferred = scade_mlp_bwd2 /
 * scade_mlp_bwd_lp2 / scade_mlp_bwd_f16_2 (wgrad_f16 = 1) WITHOUT their reduce launch: the partial rows stay in the
 * workspaces (which must stay alive until scade_step_finish has run) and *reduce_desc - 64 bytes of HOST memory,
 * opaque: pointers and row counts of the two entries, in the order of the call - describes them.  For train steps
 * whose gradient is not exchanged between ranks (with ranks: reduce -> all-reduce -> scade_step_finish without
 * descriptor).  scade_mlp_bwd_lp2_deferred also takes gmax_pre (NULL, or per entry NULL / 256 floats whose maximum is
 * the loss-scale maximum of that entry's EFFECTIVE output gradient - colour channels as they are, the density channel
 * times sigmoid(10 alpha_pre) - when the caller's loss launch already produced it: the maxima launch is then
 * skipped) and reduce_desc may be NULL there (then grad_flat is written as by scade_mlp_bwd_lp2). */
typedef struct { unsigned char opaque[64]; } scade_reduce_desc;
int scade_mlp_bwd2_deferred(const float* const* packed, const float* const* packed_t, const float* const* acts,
                            const float* const* g_out, const int* P, float* const* workspace, void* reduce_desc,
                            void* stream);
int scade_mlp_bwd_lp2_deferred(const void* const* packed_t_lp, int bf16, const void* const* acts,
                               const float* const* g_out, const int* P, void* const* workspace,
                               float* const* grad_flat, const float* const* gmax_pre, void* reduce_desc,
                               void* stream);
int scade_mlp_bwd_f16_2_deferred(const float* const* packed, const void* const* packed_t_f16,
                                 const float* const* acts, const float* const* g_out, const int* P,
                                 float* const* workspace, void* reduce_desc, void* stream);
/* scade_adam_step2 with the weight gradient's last stage inside it: [sum of the partial rows reduce_desc describes ->
 * grads[0]] -> Adam on both segments (arguments as scade_adam_step2; segment 0 = the n_nets networks' parameters,
 * n_nets x 589,700 consecutive floats in scade_mlp_pack order; device states, when given, were ALREADY advanced for
 * this step).  reduce_desc NULL: Adam only.  Same arithmetic, same summation order, same bits as the separate
 * launches.  (The next step's weight packs as a second phase of this launch behind a grid-wide barrier were built
 * and measured in round 6: 5 - 8 x slower than the separate launch on this 8-XCD part - step_finish.hip.) */
int scade_step_finish(float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                      const long* n, const float* lr, const float* beta1, const float* beta2, const float* eps,
                      const int* step, const float* grad_scale, float* const* state, const void* reduce_desc,
                      int n_nets, void* stream);

/* The fine tail, the train loss and the backward of both tails of a TRAIN step in one launch (+ the loss's
 * one-workgroup reduce): scade_ray_tail (fine form, run_scade_scannet.py:720-730) -> scade_train_loss_fb (:954,
 * :968-983) -> scade_ray_tail_bwd, and - raw0 != NULL - scade_composite_bwd of the coarse ray; same arithmetic, same
 * bits.  Outputs: the tail's (rgb_map .. depth_map, samples = the depth hypotheses, z_std), loss4, the scale / shift
 * gradient rows (n_ss as in scade_train_loss_fb), g_raw [N,S,4] and g_raw0 [N,S0,4].  workspace: 8 N floats. */
int scade_ray_tail_train(const float* raw, const float* z_vals, const float* rays, int ray_stride,
                         const float* noise, int N, int S, const float* u, int u_stride, int Si,
                         float* rgb_map, float* disp_map, float* acc_map, float* weights, float* depth_map,
                         float* samples, float* z_std,
                         const float* rgb0, const float* target, const float* hyp, const float* scales,
                         const float* shifts, const long long* img_i_dev, int img_i, const float* mask,
                         int mse_masked, int carve_on, float carve_weight, float threshold, float out_scale,
                         int K, float* workspace, float* loss4, float* g_scales, float* g_shifts, int n_ss,
                         float* g_raw, const float* raw0, const float* z0, const float* noise0, int S0,
                         float* g_raw0, void* stream);

/* Batch staging for graph-captured steps: n <= 8 device-to-device copies (whole, 4-byte-aligned words) and,
 * scalar_dst != NULL, one 8-byte scalar (the step's training-image index, run_scade_scannet.py:930) in ONE launch -
 * what the reference's per-step batch assembly (:930-960: rays, target colours, depth hypotheses of the sampled
 * pixels) writes into the step's input tensors.  tick_states != NULL: two device-resident optimizer states (or NULL
 * entries) are advanced by one step here (the tick of scade_adam_step_dev; pass ticked = 1 to scade_adam_step2 then):
 * the step's prologue is one launch. */
int scade_stage_inputs(const void* const* src, void* const* dst, const long* bytes, int n,
                       long long* scalar_dst, long long scalar, float* const* tick_states, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCADE_HIP_H */

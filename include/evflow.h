/*
 * evflow.h -- C ABI of libevflow_hip.so, the MI355X (gfx950) implementation of
 * the tudelft/event_flow hot path.
 *
 * The reference has no FFI of its own: its boundary is the Python API of
 * dataloader/encodings.py, utils/iwe.py, loss/flow.py and models/{model,...}.py
 * (SURVEY.md section 8b).  The host-side mirror of that API lives in
 * event_flow_amd/ and binds exactly these entry points through ctypes
 * (event_flow_amd/_lib.py); INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to contiguous float32 unless noted;
 *  - `stream` is a hipStream_t (passed as void*); work is only enqueued, the
 *    call never synchronises, allocates or frees;
 *  - return 0 on success, -22 (EINVAL) on a bad argument, -(1000+hipError_t)
 *    when a launch fails; nothing throws;
 *  - images are row-major [H][W]; events are rows (t, y, x, p) as in the
 *    reference's `event_list` (dataloader/base.py:197-208, [B,N,4] after
 *    custom_collate :248-265); flow maps are [B,2,H,W] with channel 0 = x and
 *    channel 1 = y (loss/flow.py:74-76);
 *  - activations inside the network are channels-last: v [B,H,W,C] float32,
 *    spikes bit-packed one uint32 per pixel per 32 channels [B,H,W,C/32].
 */
#ifndef EVFLOW_H
#define EVFLOW_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVF_OK 0
#define EVF_EINVAL (-22)
#define EVF_ENOTSUP (-95) /* a run-time dependency is absent (evf_comm_*: no librccl), or this form does not serve the request */

/* library / device probe (no GPU work) */
int evf_version(void);          /* 100*major + minor */
int evf_device_count(void);     /* hipGetDeviceCount, 0 when there is no GPU */

/* ------------------------------------------------------------------ encodings
 * dataloader/encodings.py:30-45  events_to_image: img[y][x] (+)= val.
 * xs, ys, vals: [n] float32 (indices are truncated like .long()).
 * accumulate=0 -> plain store (last writer wins), out is zero-filled first. */
int evf_events_to_image(const float* xs, const float* ys, const float* vals, int n,
                        int H, int W, int accumulate, float* out, void* stream);

/* Batched window encoding straight from the event list (replaces the
 * per-sample CPU loop of dataloader/h5.py:282-286):
 *   cnt   [B,2,H,W]  dataloader/encodings.py:70-85  (events_to_channels)
 *   mask  [B,1,H,W]  dataloader/base.py:159-171     (create_mask_encoding)
 *   voxel [B,nb,H,W] dataloader/encodings.py:48-67  (events_to_voxel)
 *   pol   [B,N,2]    dataloader/base.py:210-222     (create_polarity_mask)
 * Any output pointer may be NULL.  Events with p == 0 are padding and are
 * ignored everywhere. */
int evf_encode_events(const float* ev, int B, int N, int H, int W, int num_bins, int round_ts,
                      float* cnt, float* mask, float* voxel, float* pol, void* stream);
/* The same binning for all P passes of a BPTT window in one launch.  ev [B][P][N][4] is batch-major (the loss's window
 * event list [B][P*N][4] is the same memory); network inputs come out pass-major, loss inputs batch-major:
 *   dense = [cnt P*B*2*H*W | voxel P*B*num_bins*H*W | mask B*P*H*W] (the parts `want` selects: 1 cnt, 2 voxel, 4 mask; ONE
 *   allocation, zero-filled here), pol [B][P*N][2] (null = not wanted). */
int evf_encode_window(const float* ev, int B, int P, int N, int H, int W, int num_bins, int round_ts, int want,
                      float* dense, float* pol, void* stream);

/* ------------------------------------------------------------------ IWE
 * Generic warp + splat (utils/iwe.py:20-92 get_interpolation + interpolate).
 *   flow      [n_maps,B,2,H,W]   flow maps; event e of sample b uses map
 *                                 map_of_event[e] (NULL -> map 0)
 *   ev        [B,M,4]            events (t,y,x,p); t is shifted by
 *                                 ts_shift[e] (int32 [M], NULL -> 0) before use
 *                                 (loss/flow.py:90)
 *   w0, w1    per-event weights with element stride wstride and batch stride
 *             M*wstride (the polarity masks); NULL -> 1.0 / unused
 *   out       [B,nch,H,W], zero-filled by the call.
 *   mode bits: 1 = round indices (torch.round, half-to-even) instead of bilinear
 *              2 = zero flow (FWL/RSAT "image of events", loss/flow.py:491-494)
 *              4 = also accumulate w*tau images (tau = t, or tref_ts - t when
 *                  mode&8): channel order (I_w0, I_w1, TS_w0, TS_w1)
 *   nch = 1 (w0 only / no weights), 2 (w0,w1) or 4 (with mode&4).
 * mode=1, nch=2, tref=1 is compute_pol_iwe (utils/iwe.py:132-153): integer
 * valued, bit-exact. */
int evf_iwe_splat(const float* flow, const float* ev, const int32_t* map_of_event, const int32_t* ts_shift,
                  const float* w0, const float* w1, int wstride,
                  int B, int M, int H, int W, float flow_scaling, float tref, float tref_ts,
                  int mode, int nch, float* out, void* stream);

/* Materialising forms kept for API parity with utils/iwe.py:
 *   get_interpolation (:20-74): ev [B,N,4], evflow [B,N,2] (fy,fx) ->
 *     idx, weights [B,M,1] float32, M = N (round_idx) or 4N in corner blocks
 *     (top-left, top-right, bottom-left, bottom-right), purged like :4-17,65-72;
 *   interpolate (:77-92): scatter-add weights (* pol_mask, element stride
 *     pstride) into out [B,1,H,W] (zero-filled by the call). */
int evf_get_interpolation(const float* ev, const float* evflow, int B, int N, int H, int W,
                          float flow_scaling, float tref, int round_idx,
                          float* idx, float* weights, void* stream);
int evf_interpolate(const float* idx, const float* weights, const float* pol_mask, int pstride,
                    int B, int M, int H, int W, float* out, void* stream);

/* Contrast-maximisation loss, loss/flow.py:176-301 (EventWarping.forward).
 * One call handles all S flow scales of one window.
 *   flow   [S,Pm,B,2,H,W] (Pm = P, or 1 when overwrite_intermediate)
 *   ev     [B,M,4], pol [B,M,2], ev_pass int32 [M] (pass index of each event)
 *   mask   [B,Pk,H,W] event masks (Pk = P or 1), may be NULL when !use_mask
 *   images [S,B,8,H,W] workspace: (fw,bw) x (pos,neg) x (IWE, TS)   (saved)
 *   stats  [S,B,2,2]   workspace: per direction (sum A^2, #nonzero px) (saved)
 *   smooth_part [S,nblk_smooth] workspace (evf_cm_smooth_blocks gives nblk)
 *   loss   [1] output.
 * flags: 1 = use smoothing mask, 2 = overwrite_intermediate, 4 = loss_scaling */
int evf_cm_smooth_blocks(int B, int P, int H, int W);
/* The loss launches merged (default): forward = [pre-warp | smoothness partials] + [striped splat + image statistics + the
 * last block finishes the loss]; backward = [smoothness gradient | image gradients] + [event gather].  evf_cm_merge(0): one
 * launch per pass as before (process-wide; the equivalence test, A/B measurements; environment EVF_CM_MERGE=0 likewise). */
int evf_cm_merge(int on);
/* dL/dflow of the events (loss/flow.py:56-119 under autograd: an index_add at the events' source pixels): device-scope float
 * atomics, or -- many events per flow map -- one block per stripe of rows of a map that compacts its events into an LDS queue
 * and sums in LDS (no global atomics).  mode -1: by size (default), 0: atomics, 1: stripes whenever they fit (process-wide;
 * environment EVF_CM_BWD_LDS at load). */
int evf_cm_bwd_lds(int mode);
/* ws: NULL, or evf_cm_loss_ws(S,B,M,H,W) floats of scratch (when that is > 0): the images are then accumulated in
 * LDS stripes from pre-warped events instead of with device-scope atomics (same sums, other summation order). */
int64_t evf_cm_loss_ws(int S, int B, int M, int H, int W);
int evf_cm_loss_fwd(const float* flow, const float* ev, const float* pol, const int32_t* ev_pass,
                    const float* mask, int S, int P, int B, int M, int H, int W,
                    float flow_scaling, float regul_weight, int flags,
                    float* images, float* stats, float* smooth_part, float* loss, float* ws, void* stream);

/* Backward of the above: dflow [S,Pm,B,2,H,W] = grad_out * dL/dflow (written,
 * not accumulated).  gimages is scratch of S*B*8*H*W floats (16-byte aligned;
 * internal layout [S][B][2][H*W][4]).  Reproduces the
 * max(0,1-|d|) tie sub-gradient (0.5) and the #nonzero-px denominator path
 * of the reference under torch>=1.8 autograd. */
int evf_cm_loss_bwd(const float* flow, const float* ev, const float* pol, const int32_t* ev_pass,
                    const float* mask, int S, int P, int B, int M, int H, int W,
                    float flow_scaling, float regul_weight, int flags,
                    const float* images, const float* stats, const float* grad_out,
                    float* gimages, float* dflow, void* stream);

/* Per-sample reductions used by FWL / RSAT (loss/flow.py:481-579):
 *   evf_image_variance: unbiased variance over H*W of img [B,1,H,W] -> [B]
 *   evf_avg_ts_ratio  : images [B,4,H,W] (I_pos,I_neg,TS_pos,TS_neg) ->
 *                       sum((TS/(I+1e-9)/P)^2)/#{I_pos+I_neg>0}  [B]            */
int evf_image_variance(const float* img, int B, int HW, float* out, void* stream);
int evf_avg_ts_ratio(const float* images, int B, int HW, float P, float* out, void* stream);

/* AEE, loss/flow.py:594-628.  flow, gt [B,2,H,W]; mask [B,H,W]; ratio [B] =
 * dt_gt/dt_input per sample; out [B,3] = (sum err, n valid, n outliers). */
int evf_aee(const float* flow, const float* gt, const float* mask, const float* ratio,
            int B, int H, int W, float flow_scaling, float* out, void* stream);

/* ------------------------------------------------------------------ network
 * Conv + spiking-neuron cells, models/spiking_submodules.py.
 * Weights are passed in a packed layout produced by evf_pack_conv_weight from
 * the reference's [Cout,Cin,k,k] tensors.
 *
 * neuron kinds */
#define EVF_LIF 0
#define EVF_PLIF 1
#define EVF_ALIF 2
#define EVF_XLIF 3
/* surrogate gradients, models/spiking_util.py:28-93 */
#define EVF_ARCTAN 0
#define EVF_SUPERSPIKE 1
#define EVF_TRIANGLE 2
#define EVF_MULTIGAUSS 3

/* Repack a conv weight [Cout,Cin,3,3] (float32, torch layout) into the MFMA
 * operand layout used by the conv kernels: fwd -> B operand of
 * out[pix][co] += x[pix+tap][ci]*w ; transposed=1 builds the (flipped,
 * transposed) operand of the input-gradient conv.  dst has 9*Cin*Cout floats
 * (Cin, Cout multiples of 32). */
int evf_pack_conv_weight(const float* w, int Cout, int Cin, int transposed, float* dst, void* stream);
/* inverse for weight gradients: packed [9][Cin][Cout] accumulators -> torch layout, dst (+)= */
int evf_unpack_conv_wgrad(const float* packed, int Cout, int Cin, int accumulate, float* dst, void* stream);

/* Head cell: dense small-Cin 3x3 conv (+ LIF update).  ConvLIF.forward
 * (spiking_submodules.py:96-126) with real-valued input x [B,Cin,H,W] (NCHW,
 * the event count / voxel tensor), Cin <= 8, Cout = 32.
 *   w [32,Cin,3,3] torch layout; leak, thresh [32] raw parameters
 *   v_prev [B,H,W,32] or NULL (zeros); z_prev bits [B,H,W] or NULL
 *   outputs v_out [B,H,W,32], z_out bits [B,H,W] and (optional, may be NULL) zT_out:
 *   the same spikes as channel-major bit planes [B,H,32,ceil(W/32)] (bit = x mod 32),
 *   the layout the weight-gradient kernel consumes */
int evf_head_lif_fwd(const float* x, const float* w, const float* leak, const float* thresh,
                     const float* v_prev, const uint32_t* z_prev, int B, int Cin, int H, int W,
                     int hard_reset, float* v_out, uint32_t* z_out, uint32_t* zT_out, void* stream);

/* 32->32 channel spiking conv cell on bit-packed spikes.
 * ConvLIF.forward (:96-126) when z_rec_w == NULL, ConvLIFRecurrent.forward
 * (:516-551) otherwise.  x bits [B,H,W] is the input spike map, z_prev the
 * cell's own previous spikes.  w_ff / w_rec packed (evf_pack_conv_weight). */
int evf_conv_lif_fwd(const uint32_t* x, const float* w_ff, const float* w_rec,
                     const float* leak, const float* thresh,
                     const float* v_prev, const uint32_t* z_prev, int B, int H, int W,
                     int hard_reset, float* v_out, uint32_t* z_out, uint32_t* zT_out, void* stream);

/* Same cell on the bf16 matrix cores with fp32-equivalent numerics ("bf16x3"):
 * spikes are exact in bf16, every fp32 weight is split exactly into three bf16
 * terms (evf_pack_conv_weight_b3, dst = 54 KiB per conv), products are exact and
 * the accumulation is fp32 -- 5.3x fewer matrix-core cycles than the fp32 MFMA
 * form, same rounding class as any re-ordered fp32 convolution. */
int evf_pack_conv_weight_b3(const float* w, int Cout, int Cin, void* dst, void* stream);
int evf_conv_lif_fwd_b3(const uint32_t* x, const void* wb_ff, const void* wb_rec,
                        const float* leak, const float* thresh,
                        const float* v_prev, const uint32_t* z_prev, int B, int H, int W,
                        int hard_reset, float* v_out, uint32_t* z_out, uint32_t* zT_out, void* stream);
/* The same for the layer under the prediction head, with the head (evf_pred_fwd: models/model.py:197-199, :265) in its
 * epilogue: pred_w [2][32], pred_b [2], flow [B,2,H,W] (written). */
int evf_conv_lif_fwd_b3_pred(const uint32_t* x, const void* wb_ff, const void* wb_rec, const float* leak,
                             const float* thresh, const float* v_prev, const uint32_t* z_prev, int B, int H, int W,
                             int hard_reset, float* v_out, uint32_t* z_out, uint32_t* zT_out,
                             const float* pred_w, const float* pred_b, float* flow, void* stream);
/* Several (pass, layer) cells of one window in ONE launch.  The reference's loop (train_flow.py:98-139) runs the stack
 * pass by pass (models/model.py:255-265); cell (t, l) needs cells (t, l-1) and (t-1, l) only, so all cells with equal
 * t + l are independent.  Between evf_fwd_defer_begin() and evf_fwd_defer_flush(), evf_conv_lif_fwd_b3[_pred] RECORD their
 * launch under the index last given to evf_fwd_defer_slot() (0 .. 95) instead of launching; the flush launches the recorded
 * cells, one kernel per non-empty index in increasing order (blockIdx.z = cell x sample; same kernel body, bit-identical
 * results), and ends the recording.  The caller guarantees that cells under one index are independent, that a cell's
 * operands come from lower indices or from launches made before, and that nothing reads a cell's outputs before the
 * flush.  evf_fwd_defer_pending() = cells recorded and not yet launched.
 * CONTEXTS / THREADS: a recording (this kind and evf_bwd_defer_*) belongs to the STREAM it was opened on -- the library keeps
 * one recorder per stream with an open recording (up to 16) and every entry point consults the recorder of its `stream`
 * argument only.  Host threads driving different models on different streams record and launch independently, also through
 * PyTorch's autograd (one backward worker thread per device, but each node on the stream of its forward); calls that belong to
 * ONE recording must not race with each other (one thread at a time per stream).  A second begin of the same kind on a stream
 * with an open recording returns EVF_EINVAL; begin with all contexts in use as well.
 * STALE READS: a recorded cell has not run, so its outputs hold old bytes until the flush.  evf_defer_poison(1) (debug aid)
 * makes every recorded forward cell fill its v_out with NaN bit patterns and its spike word outputs with 0xFFFFFFFF at record
 * time (on `stream`), so that a read outside the library before the flush is conspicuous instead of plausible. */
int evf_fwd_defer_begin(void* stream);
int evf_fwd_defer_slot(int index, void* stream);
int evf_fwd_defer_pending(void* stream);
int evf_fwd_defer_flush(void* stream);
int evf_defer_poison(int on);
/* The same for the backward of a window (autograd of train_flow.py:141-154 over the passes of models/model.py:255-265): a
 * pass's backward is a chain of 13 steps (fused backward of the top layer, its input gradient, the next layer, ..., the head
 * layer); step s of pass t needs step s - 1 of pass t and steps s, s + 1 of pass t + 1.  Under the index 2 (P - 1 - t) + s
 * the cells of one index are independent and of one kind.  While recording, evf_lif_bwd_wgrad[2|_top] (default neuron, fp32
 * g_cur), evf_conv_dgrad_b3_f32[_pair] (no accumulation, no PLIF term) and evf_head_lif_bwd_wgrad record their launch under
 * the index last given to evf_bwd_defer_slot (0 .. 95); any other call of these entry points first launches everything
 * recorded.  evf_bwd_defer_flush launches index after index -- the fused-backward cells of an index as one kernel, its
 * input-gradient cells as one kernel, head cells one by one -- and ends the recording.  Same kernel bodies as the one-cell
 * launches.  Per-stream recorder (see above); the caller guarantees the index order, that nothing else reads a cell's outputs
 * before the flush, and that every buffer a recorded cell refers to stays allocated until then.  With the gradient
 * pre-split (evf_lif_bwd_wgrad* given g_split, evf_conv_dgrad_b3[_pair]) the input-gradient cells are recorded as well. */
int evf_bwd_defer_begin(void* stream);
int evf_bwd_defer_slot(int index, void* stream);
int evf_bwd_defer_pending(void* stream);
int evf_bwd_defer_flush(void* stream);
/* evf_bwd_defer_hold_heads(1): the recorded head cells (evf_head_lif_bwd_wgrad, evf_head_plif_bwd_wgrad) wait for
 * evf_bwd_defer_flush even when a call that cannot be recorded launches everything else recorded so far -- and then run all
 * passes in one launch.  The caller promises one dL/d(spikes) buffer of the head layer per pass (nothing launched in between
 * rewrites what a waiting head cell reads).  PLIF networks: their hidden cells' input gradients are not recordable, every pass
 * flushes.  Reset by evf_bwd_defer_begin. */
int evf_bwd_defer_hold_heads(int on, void* stream);
/* Measurement aid: evf_defer_profile(1) brackets every launch of the following flushes with HIP events;
 * evf_defer_profile_read synchronises the device, returns per kind (0 forward cells, 1 fused-backward cells, 2
 * input-gradient cells, 3 head backward one pass per launch, 4 head forward of a window in one launch, 5 head backward of
 * a window in one launch; 6 a feed-forward hidden layer's forward of a window in one launch; 7 an empty bracket; 8 a hidden
 * layer's backward of a window in one launch; 9 evf_conv_dgrad_b3_multi; 10..15 unused) the summed duration in ms and the number
 * of launches -- SIXTEEN entries each -- and switches it off.  evf_defer_profile(2): the brackets are recorded INTO a stream capture as one-thread timestamp
 * kernels (wall clock) in front of and behind every launch; evf_defer_profile(0) after the capture stops recording and
 * keeps them; after replays of the graph evf_defer_profile_read returns the durations inside the LAST replay. */
int evf_defer_profile(int on);
int evf_defer_profile_read(float* ms16, int* count16);

/* Neuron backward (autograd of :103-126 / :523-551 with the surrogate of
 * spiking_util.py:88-93).  Per element:
 *   g_v = g_v_out + g_z_out * sg(v_out - thresh)
 *   g_cur = g_v * (1 - leak)                 -> g_cur [B,H,W,32]
 *   g_v_prev = g_v * leak * (1 - z_prev)     (hard)  |  g_v * leak (soft)
 * and per-channel sums into g_leak[32], g_thresh[32] (accumulated).
 * g_z_out may be NULL (zeros), g_v_out may be NULL (zeros). g_v_prev may alias g_v_out. */
int evf_lif_bwd(const float* g_z_out, const float* g_v_out, const float* v_out,
                const float* v_prev, const uint32_t* z_prev,
                const float* leak, const float* thresh, int B, int H, int W,
                int hard_reset, int surrogate, float act_width,
                float* g_cur, float* g_v_prev, float* g_leak, float* g_thresh, void* stream);

/* Fused neuron backward + weight gradients of one 32->32 cell (layers fed by spikes):
 * evf_lif_bwd and evf_conv_wgrad_bits (for the ff input and, when zT_prev != NULL, the
 * recurrent input) in one pass over the data, with the matrix-core part in exact
 * bf16x3 form.  xT / zT_prev are the channel-major spike bit planes written by the
 * forward kernels (zT_out).  slab_* [nslab][9][32][32] with nslab =
 * evf_lif_bwd_wgrad_slabs(B,H,W); written, or += when accumulate. */
int evf_lif_bwd_wgrad_slabs(int B, int H, int W);
int evf_lif_bwd_wgrad(const float* g_z_out, const float* g_v_out, const float* v_out,
                      const float* v_prev, const uint32_t* z_prev,
                      const uint32_t* xT, const uint32_t* zT_prev,
                      const float* leak, const float* thresh, int B, int H, int W,
                      int hard_reset, int surrogate, float act_width,
                      float* g_cur, void* g_split, float* g_v_prev, float* g_leak, float* g_thresh,
                      float* slab_ff, float* slab_rec, int accumulate, void* stream);
/* The same with dL/d(output spikes) in two parts, g_z_out + g_z_out2 (either may be NULL): the part from the
 * layer above (this pass) and the part from the cell's own recurrent input gradient (one pass later, autograd of
 * spiking_submodules.py:523-551) stay in separate buffers and are added here, in that order -- what the accumulating form of
 * evf_conv_dgrad_b3_f32 did by reading and rewriting one buffer. */
int evf_lif_bwd_wgrad2(const float* g_z_out, const float* g_z_out2, const float* g_v_out, const float* v_out,
                       const float* v_prev, const uint32_t* z_prev,
                       const uint32_t* xT, const uint32_t* zT_prev,
                       const float* leak, const float* thresh, int B, int H, int W,
                       int hard_reset, int surrogate, float act_width,
                       float* g_cur, void* g_split, float* g_v_prev, float* g_leak, float* g_thresh,
                       float* slab_ff, float* slab_rec, int accumulate, void* stream);
/* The same for the non-recurrent layer directly under the prediction head (models/model.py:197-199, :265), with the
 * head's backward (evf_pred_bwd) inside: flow / g_flow [B,2,H,W], pred_w [2][32], z_out [B,H,W] = this layer's output
 * spikes; d_pred_w [2][32] and d_pred_b [2] are accumulated.  The layer's dL/d(spikes) rows are formed in registers. */
int evf_lif_bwd_wgrad_top(const float* flow, const float* g_flow, const float* pred_w, const uint32_t* z_out,
                          float* d_pred_w, float* d_pred_b, const float* g_v_out, const float* v_out,
                          const float* v_prev, const uint32_t* z_prev, const uint32_t* xT,
                          const float* leak, const float* thresh, int B, int H, int W,
                          int hard_reset, int surrogate, float act_width,
                          float* g_cur, void* g_split, float* g_v_prev, float* g_leak, float* g_thresh,
                          float* slab_ff, int accumulate, void* stream);

/* Input-gradient conv on the exact bf16 split: g_split = three bf16 planes
 * [3][B,H,W,32] (g = hi + mid + lo, optional output of evf_lif_bwd_wgrad; g_cur may then
 * be NULL), wT_b3 = evf_pack_conv_weight_b3t(w) (54 KiB).  g_x [B,H,W,32] fp32 is written,
 * or += when accumulate.  Six-term product, fp32 accumulation (fp32 round-off class). */
int evf_pack_conv_weight_b3t(const float* w, int Cout, int Cin, void* dst, void* stream);
/* Both layouts of up to 16 conv weights [32][32][3][3] in one launch.  w, dst_b3, dst_b3t: HOST arrays of `count`
 * device pointers (dst_b3 or dst_b3t may be NULL to skip that layout).  The weights change at every optimizer step
 * (train_flow.py:163), so this runs once per step. */
int evf_pack_conv_weights_b3_multi(const void* const* w, void* const* dst_b3, void* const* dst_b3t, int count, void* stream);
int evf_conv_dgrad_b3(const void* g_split, const void* wT_b3, float* g_x, int accumulate,
                      int B, int H, int W, const float* g_P, const uint32_t* x_bits, void* stream);
/* The same from the fp32 gradient g_cur [B,H,W,32] of evf_lif_bwd_wgrad (g_split = NULL there): the exact 3-way split
 * happens while the halo is staged, the result is bit-identical; 128 instead of 192 B/pixel on both sides. */
int evf_conv_dgrad_b3_f32(const float* g_cur, const void* wT_b3, float* g_x, int accumulate,
                          int B, int H, int W, const float* g_P, const uint32_t* x_bits, void* stream);
/* Both input gradients of a recurrent cell from the pre-split planes in one call (one halo read):
 * g_x (+)= conv^T(g, W_ff), g_x2 = conv^T(g, W_rec) (written) -- models/spiking_submodules.py:520,530. */
int evf_conv_dgrad_b3_pair(const void* g_split, const void* wT_b3, float* g_x, int accumulate,
                           const void* wT2_b3, float* g_x2, int B, int H, int W,
                           const float* g_P, const uint32_t* x_bits, void* stream);
/* Input gradients of nprod <= 16 products (gradient planes, weight set, output) in ONE persistent launch (k_dgrad_diag_dma),
 * straight from the caller instead of through a backward recording: host arrays of device pointers.  g_split[k]: the three bf16
 * planes of dL/d(current) (as evf_conv_dgrad_b3); g_x[k] is WRITTEN; g_P_raw / x_bits (the arrays or single entries may be NULL):
 * the PLIF trace term of product k from the RAW map (evf_conv_dgrad_b3 with `accumulate | 2`).  Bit-identical to one
 * evf_conv_dgrad_b3 call per product.  EVF_ENOTSUP: the shape does not fit the persistent kernel's index arithmetic. */
int evf_conv_dgrad_b3_multi(int nprod, const void* const* g_split, const void* const* wT_b3, void* const* g_x,
                            const void* const* g_P_raw, const void* const* x_bits, int B, int H, int W, void* stream);
int evf_conv_dgrad_b3_multi_fits(int B, int H, int W); /* 1: the shape fits, 0: EVF_ENOTSUP */
/* Which kernel serves evf_conv_dgrad_b3_f32[_pair] (results are bit-identical): -1 chosen by shape (default), 0 the
 * one-phase-after-the-other LDS kernel, 1 the wave-specialised one (producer / consumer waves, double-buffered planes).
 * Process-wide; for A/B measurements and the equivalence test. */
int evf_conv_dgrad_select(int which);
/* Which kernel launches the RECORDED input-gradient cells of a backward index (evf_bwd_defer_*; results are bit-identical):
 * -1 default (environment EVF_DGRAD_DIAG=lds|ws, else 1), 0 k_dgrad_diag (the LDS kernel's body, one block per tile pair),
 * 1 k_dgrad_diag_ws / k_dgrad_diag_dma (persistent blocks over the flat list of products: fp32 gradient in / pre-split planes
 * staged by LDS-DMA), 2 like 1 with k_dgrad_diag_ring for the pre-split planes (halo rows in a ring: vertically adjacent tiles
 * share two of six; measured slower, EVF_DGRAD_RING=1).  Process-wide. */
int evf_dgrad_diag_select(int which);
/* Which kernel launches the RECORDED forward cells of an index (evf_fwd_defer_*; results are bit-identical): -1 default
 * (environment EVF_FWD_DIAG=tile|persistent|teams, else 2), 0 k_fwd_diag (one 8 x 32 tile per block, the body of the one-cell
 * launch), 1 k_fwd_diag_p (persistent blocks, a strip per wave), 2 k_fwd_diag_t (persistent blocks of a matrix team and an
 * element-wise team, evf_fwd_teams.hip).  Process-wide; for A/B measurements and the equivalence test. */
int evf_fwd_diag_select(int which);
/* Which kernel launches the RECORDED fused-backward cells of a backward index: -1 default (environment
 * EVF_BWD_DIAG=fused|teams4|teams, else 2), 0 k_bwd_diag (every wave through load / neuron backward / staging / matrix phase,
 * the body of the one-cell launch), 1 / 2 k_bwd_diag_ws (four / eight waves stream and stage, four contract: vector and
 * matrix pipes of a SIMD busy at the same time; 512 / 768 threads).  Same state gradients bit for bit; the weight-gradient slabs and per-channel sums agree to
 * fp32 round-off (other partial-sum grouping).  Process-wide. */
int evf_bwd_diag_select(int which);
/* ... and for a recurrent cell both input gradients in one launch: g_x (+)= conv^T(g_cur, W_ff) as above,
 * g_x2 = conv^T(g_cur, W_rec) (written) -- dL/d(previous output spikes), models/spiking_submodules.py:530. */
int evf_conv_dgrad_b3_f32_pair(const float* g_cur, const void* wT_b3, float* g_x, int accumulate,
                               const void* wT2_b3, float* g_x2, int B, int H, int W,
                               const float* g_P, const uint32_t* x_bits, void* stream);

/* PLIF cells (models/spiking_submodules.py:129-227, :554-657): LIF + a per-channel
 * pre-synaptic trace pt' = pt*s(leak_pt) + (1-s(leak_pt))*AvgPool3x3(mean_c|input|),
 * current = ff (+rec) - s(add_pt)*pt'.  Forward = the LIF kernels plus pt_prev/pt_out
 * [B,H,W,32] and P_out [B,H,W] (the pooled activity, saved for the backward).  Backward =
 * evf_lif_bwd_wgrad / evf_lif_bwd (they yield g_cur) followed by evf_plif_trace_bwd
 * (g_pt carry, d leak_pt, d add_pt; g_P_raw [B,H,W] scratch = d loss / d pooled activity,
 * g_P_in [B,H,W] = its adjoint through AvgPool3x3 and the channel mean); the gradient that
 * reaches the input spikes through the trace is added by evf_conv_dgrad_b3 when g_P = g_P_in
 * and x_bits are given -- or when g_P = g_P_raw with bit 1 of `accumulate` set (`accumulate | 2`): the input-gradient kernel
 * then applies AvgPool3x3^T / 32 itself (the same sums in the same order), and evf_plif_trace_bwd may be called with
 * g_P_in = NULL (one launch less per cell).  evf_plif_trace_bwd's `pt_out` is not read (may be NULL): pt' is recomputed from
 * pt_prev and P with the forward kernels' own expression (evf_plif_trace, csrc/evf_common.h: the same bits).
 *
 * XLIF cells (models/spiking_submodules.py:337-435, :771-875) ride on these entry points: the SAME pre-synaptic trace, which
 * raises the threshold -- thresh = t0.clamp_min(0.01) + t1.clamp_min(0) * pt' -- instead of being subtracted from the current
 * (current = ff (+rec)).  Bits 1-2 of `hard_reset` (evf_conv_plif_fwd_b3[_pred], evf_head_plif_fwd, evf_plif_bwd_wgrad2 / _top,
 * evf_head_plif_bwd_wgrad: pass `hard_reset | 2`) or of `accumulate` (evf_plif_bwd_wgrad_window[_top]: `accumulate | 2`) select
 * them; `thresh` / `g_thresh` then carry t0 and its gradient, `add_pt` / `g_add_pt` carry t1 and its gradient (no sigmoid:
 * the clamp's sub-gradient).  Backward forms: hard reset + arctan surrogate only, like the PLIF ones; the forward also takes
 * the soft reset (v' -= z * (t0 + t1 * pt)) cell by cell, except for the head (EVF_ENOTSUP).  evf_plif_trace_bwd has no XLIF
 * form (the trace backward of an XLIF cell lives in the fused backward kernels).
 *
 * ALIF cells (:230-334, :660-768): the value 2 in those two bits (`hard_reset | 4`, `accumulate | 4`).  The XLIF arithmetic with
 * the trace t' = t * s(leak_t) + (1 - s(leak_t)) * z driven by the cell's OWN previous spikes z (`leak_pt` = leak_t, `pt_*` = the
 * trace t; P_out is still written by the forward, nobody reads it).  z enters t' un-detached (:311): (1 - s(leak_t)) * dL/d(t') is a
 * part of dL/d(spikes) of the pass BEFORE.  Window launches carry it in registers; a one-pass launch writes it to g_zx [B,H,W,32],
 * which takes the place of an argument the ALIF cell does not need -- evf_plif_bwd_wgrad2: `g_P_raw` IS g_zx (hand it back as
 * `g_z_out2` of the pass before; it may be the same buffer: read, then written, by the same thread; a recurrent cell's own input
 * gradient is then ADDED to it, evf_conv_dgrad_b3_f32 with accumulate = 1); evf_head_plif_bwd_wgrad: `P` IS g_zx (read when
 * g_pt_carry != NULL, i.e. when there is a pass after, and written).  The input gradient has no trace term (g_P = NULL).
 * evf_plif_bwd_wgrad_top has no ALIF form (EVF_ENOTSUP: evf_pred_bwd + evf_plif_bwd_wgrad2). */
int evf_conv_plif_fwd_b3(const uint32_t* x, const void* wb_ff, const void* wb_rec,
                         const float* leak_v, const float* leak_pt, const float* add_pt, const float* thresh,
                         const float* v_prev, const uint32_t* z_prev, const float* pt_prev,
                         int B, int H, int W, int hard_reset,
                         float* v_out, uint32_t* z_out, uint32_t* zT_out, float* pt_out, float* P_out, void* stream);
/* ... with the prediction head (evf_pred_fwd: models/model.py:197-199, :265) in its epilogue, for the layer under the head:
 * pred_w [2][32], pred_b [2], flow [B,2,H,W] (written).  Like evf_conv_lif_fwd_b3_pred it makes a window's cells recordable
 * (evf_fwd_defer_*): no launch of another kind between the last cell of a pass and the first cell of the next. */
int evf_conv_plif_fwd_b3_pred(const uint32_t* x, const void* wb_ff, const void* wb_rec,
                              const float* leak_v, const float* leak_pt, const float* add_pt, const float* thresh,
                              const float* v_prev, const uint32_t* z_prev, const float* pt_prev,
                              int B, int H, int W, int hard_reset,
                              float* v_out, uint32_t* z_out, uint32_t* zT_out, float* pt_out, float* P_out,
                              const float* pred_w, const float* pred_b, float* flow, void* stream);
int evf_head_plif_fwd(const float* x, const float* w, const float* leak_v, const float* leak_pt,
                      const float* add_pt, const float* thresh, const float* v_prev, const uint32_t* z_prev,
                      const float* pt_prev, int B, int Cin, int H, int W, int hard_reset,
                      float* v_out, uint32_t* z_out, uint32_t* zT_out, float* pt_out, float* P_out, void* stream);
int evf_plif_trace_bwd(const float* g_cur, const float* g_pt_carry, const float* pt_prev, const float* pt_out,
                       const float* P, const float* leak_pt, const float* add_pt, int B, int H, int W,
                       float* g_pt_prev, float* g_P_raw, float* g_P_in, float* g_leak_pt, float* g_add_pt,
                       int row_ld, void* stream);

/* PLIF hidden cells, one pass: evf_lif_bwd_wgrad2 / evf_lif_bwd_wgrad_top with the presynaptic trace's backward inside
 * (reference models/spiking_submodules.py:634-652, autograd of :557-632): what evf_plif_trace_bwd computes from g_cur in a
 * pass of its own -- g_pt_prev (may alias g_pt_carry), the raw map g_P_raw and the sums for leak_pt / add_pt (per-block rows
 * of pitch `accumulate >> 8`, like g_leak) -- comes out of the pass that produces g_cur (the same bits per element as the two
 * calls).  g_pt_carry / pt_prev may be NULL (last pass of a window / zero state).  Recordable like the LIF forms
 * (evf_bwd_defer_*).  Default neuron only (hard reset, arctan surrogate): EVF_ENOTSUP otherwise -- use the two calls. */
int evf_plif_bwd_wgrad2(const float* g_z_out, const float* g_z_out2, const float* g_v_out, const float* v_out,
                        const float* v_prev, const uint32_t* z_prev, const uint32_t* xT, const uint32_t* zT_prev,
                        const float* leak, const float* thresh, int B, int H, int W, int hard_reset, int surrogate,
                        float act_width, float* g_cur, void* g_split, float* g_v_prev, float* g_leak, float* g_thresh,
                        float* slab_ff, float* slab_rec, int accumulate, const float* g_pt_carry, const float* pt_prev,
                        const float* P, const float* leak_pt, const float* add_pt, float* g_pt_prev, float* g_P_raw,
                        float* g_leak_pt, float* g_add_pt, void* stream);
/* LIF feed-forward hidden cells, all passes of a window in ONE launch (k_bwd_win_lif[_top]): what np calls of
 * evf_lif_bwd_wgrad2 (flow == NULL: g_z per pass) or evf_lif_bwd_wgrad_top (flow / g_flow / z_out per pass, pred_w, d_pred_w,
 * d_pred_b) compute, with dL/dv and the potential carried in registers.  Outputs per pass: g_cur (fp32) and / or g_split (the three
 * bf16 planes evf_conv_dgrad_b3 reads); either array may be NULL.  Arrays as in evf_plif_bwd_wgrad_window below (index 0 = the
 * window's last pass).  The layers on top of a LIF-FireNet know their dL/d(spikes) of every pass before anything below them has
 * run: `engine` runs them first, the rest of the window on diagonals. */
int evf_lif_bwd_wgrad_window(int np, const void* const* g_z, const void* const* flow, const void* const* g_flow,
                             const float* pred_w, const void* const* z_out, float* d_pred_w, float* d_pred_b,
                             const void* const* v_out, const void* const* v_prev, const void* const* z_prev,
                             const void* const* xT, void* const* g_cur, void* const* g_split, const float* leak,
                             const float* thresh, int B, int H, int W, float act_width, float* g_v_prev, float* g_leak,
                             float* g_thresh, float* slab_ff, int accumulate, void* stream);
/* A FEED-FORWARD PLIF hidden cell, all passes of a window in ONE launch (k_bwd_win_plif): what np calls of evf_plif_bwd_wgrad2
 * compute, with dL/dv and dL/d(pt) carried in registers from pass to pass and every potential read once -- 640 instead of 1152
 * bytes per pixel and pass.  Host arrays of np <= 16 device pointers, index 0 = the window's LAST pass (backward order); per pass:
 * g_z (may be NULL), v_out (only v_out[0] is read: v_out[s] = v_prev[s - 1]), v_prev (NULL: zero state), z_prev (NULL: none), xT,
 * pt_prev (NULL: zero), P; outputs per pass: g_cur (fp32) and / or g_split (its three bf16 planes: evf_conv_dgrad_b3[_multi]; either
 * array may be NULL), g_P_raw.  The carries start at zero behind the last pass; g_v_prev /
 * g_pt_prev (may be NULL) receive the gradients on the state entering the window.  Same arithmetic per element as the one-pass
 * form (bit-identical g_cur / g_P_raw / carries); slab and per-channel sums in another order.  Default neuron only. */
int evf_plif_bwd_wgrad_window(int np, const void* const* g_z, const void* const* v_out, const void* const* v_prev,
                              const void* const* z_prev, const void* const* xT, void* const* g_cur, void* const* g_split,
                              const void* const* pt_prev, const void* const* P, void* const* g_P_raw, const float* leak,
                              const float* thresh, const float* leak_pt, const float* add_pt, int B, int H, int W,
                              float act_width, float* g_v_prev, float* g_pt_prev, float* g_leak, float* g_thresh,
                              float* g_leak_pt, float* g_add_pt, float* slab_ff, int accumulate, void* stream);
/* ... of the feed-forward layer under the prediction head, the head's backward inside (evf_plif_bwd_wgrad_top per pass): per
 * pass flow / g_flow [B,2,H,W] and z_out [B,H,W] instead of g_z; pred_w [2][32]; d_pred_w / d_pred_b rows like g_leak. */
int evf_plif_bwd_wgrad_window_top(int np, const void* const* flow, const void* const* g_flow, const float* pred_w,
                                  const void* const* z_out, float* d_pred_w, float* d_pred_b, const void* const* v_out,
                                  const void* const* v_prev, const void* const* z_prev, const void* const* xT,
                                  void* const* g_cur, void* const* g_split, const void* const* pt_prev, const void* const* P,
                                  void* const* g_P_raw, const float* leak, const float* thresh, const float* leak_pt,
                                  const float* add_pt, int B, int H, int W, float act_width, float* g_v_prev,
                                  float* g_pt_prev, float* g_leak, float* g_thresh, float* g_leak_pt, float* g_add_pt,
                                  float* slab_ff, int accumulate, void* stream);
int evf_plif_bwd_wgrad_top(const float* flow, const float* g_flow, const float* pred_w, const uint32_t* z_out,
                           float* d_pred_w, float* d_pred_b, const float* g_v_out, const float* v_out,
                           const float* v_prev, const uint32_t* z_prev, const uint32_t* xT, const float* leak,
                           const float* thresh, int B, int H, int W, int hard_reset, int surrogate, float act_width,
                           float* g_cur, void* g_split, float* g_v_prev, float* g_leak, float* g_thresh, float* slab_ff,
                           int accumulate, const float* g_pt_carry, const float* pt_prev, const float* P,
                           const float* leak_pt, const float* add_pt, float* g_pt_prev, float* g_P_raw, float* g_leak_pt,
                           float* g_add_pt, void* stream);

/* Input-gradient conv: g_x[pix][ci] (+)= sum_tap,co g_cur[pix-tap][co]*w[co][ci][tap]
 * with wT packed by evf_pack_conv_weight(transposed=1).  g_cur, g_x [B,H,W,32].
 * Second (optional) weight/output pair shares the g_cur tile (ff + rec). */
int evf_conv_dgrad(const float* g_cur, const float* wT_a, float* g_a, int acc_a,
                   const float* wT_b, float* g_b, int acc_b, int B, int H, int W, void* stream);

/* Weight-gradient conv on bit-packed inputs:
 * dW[tap][ci][co] = sum_pix x[pix+tap][ci] * g_cur[pix][co].  Every block reduces its
 * rows into one slab wg_partial[blk][9][32][32] (written, or += when accumulate);
 * evf_conv_wgrad_slabs gives the slab count, evf_reduce_slabs sums the slabs into the
 * torch layout [Cout][Cin][3][3] (dst = or +=). */
int evf_conv_wgrad_bits(const uint32_t* x, const float* g_cur, int B, int H, int W,
                        float* wg_partial, int accumulate, void* stream);
int evf_conv_wgrad_slabs(int B, int H, int W);
int evf_reduce_slabs(const float* partial, int nslab, int n, int accumulate, float* dst, void* stream);
/* evf_reduce_slabs (accumulate = 1) for up to 16 weight tensors in one launch: partial / dst are HOST arrays of `count`
 * device pointers, every partial [nslab][n]. */
int evf_reduce_slabs_multi(const void* const* partial, void* const* dst, int count, int nslab, int n, void* stream);

/* Head layer, neuron backward + weight gradient fused (one pass over the gradient tensors):
 * evf_lif_bwd plus dW partials per block into slab [evf_head_lif_bwd_wgrad_slabs(B,H,W)][32*Cin*9]
 * (torch layout [32][Cin][3][3]; accumulate = 1 adds to the slab), reduced once per window with
 * evf_sum_rows.  x_in [B,Cin,H,W] = the network input, Cin = 2.  g_cur may be null. */
int evf_head_lif_bwd_wgrad_slabs(int B, int H, int W);
int evf_head_lif_bwd_wgrad(const float* g_z_out, const float* g_v_out, const float* v_out,
                           const float* v_prev, const uint32_t* z_prev, const float* x_in, const float* leak,
                           const float* thresh, int B, int Cin, int H, int W, int hard_reset, int surrogate,
                           float act_width, float* g_cur, float* g_v_prev, float* g_leak, float* g_thresh,
                           float* slab, int accumulate, void* stream);
/* PLIF head (reference models/spiking_submodules.py:191-227, backward of :634-652): evf_head_lif_bwd_wgrad with the presynaptic
 * trace's backward in the same pass -- what evf_plif_trace_bwd computes from g_cur in a launch of its own (g_cur is not written;
 * the head's input is the event tensor, so dL/d(pooled activity) is not needed): g_pt_prev (may alias g_pt_carry) and the sums for
 * leak_pt / add_pt (rows of pitch accumulate >> 8, like g_leak).  g_pt_carry / pt_prev may be NULL.  Recordable; the recorded
 * cells of a window run in ONE launch with dL/dv and dL/d(pt) carried in registers.  Default neuron only (EVF_ENOTSUP otherwise). */
int evf_head_plif_bwd_wgrad(const float* g_z_out, const float* g_v_out, const float* v_out, const float* v_prev,
                            const uint32_t* z_prev, const float* x_in, const float* leak, const float* thresh, int B,
                            int Cin, int H, int W, int hard_reset, int surrogate, float act_width, float* g_v_prev,
                            float* g_leak, float* g_thresh, float* slab, int accumulate, const float* g_pt_carry,
                            const float* pt_prev, const float* P, const float* leak_pt, const float* add_pt,
                            float* g_pt_prev, float* g_leak_pt, float* g_add_pt, void* stream);
/* dst[k][i] += src[off[k] + i], i < n[k], for nseg <= 32 segments (host arrays of device pointers / ints):
 * the per-channel gradients of a window added into the optimizer's flat gradient buffer in one launch.
 * clear != 0: the consumed source elements are zeroed (a persistent accumulator, handed back clean). */
int evf_add_segments(float* src, void* const* dst, const int* off, const int* n, int nseg, int clear, void* stream);
/* dst[e] (+)= sum_k rows[k][e], e < n.  accumulate bit 0: add to dst; bit 1 (value 2): zero the rows after reading them.
 * PER-BLOCK PARAMETER-GRADIENT ROWS: evf_lif_bwd_wgrad[_top], evf_head_lif_bwd_wgrad (bits 8.. of their `accumulate`
 * argument) and evf_plif_trace_bwd (`row_ld`) take the pitch, in floats, of a [blocks][pitch] buffer; their per-channel
 * outputs (g_leak, g_thresh, d_pred_w, d_pred_b, g_leak_pt, g_add_pt) then address ROW 0 of it and every block adds its
 * partial sums to its own row by plain read-modify-write -- 256 blocks adding atomically into the same 64 words keep a
 * kernel alive ~4.5 us after its last block has finished.  evf_sum_rows reduces the rows once per window.  Pitch 0 =
 * dense outputs, atomic adds. */
int evf_sum_rows(float* rows, int nrows, int n, int accumulate, float* dst, void* stream);
/* Head weight gradient: dW[co][ci][ky][kx] += sum g_cur[pix][co]*x[b][ci][pix+tap] (torch layout out). */
int evf_head_wgrad(const float* x, const float* g_cur, int B, int Cin, int H, int W, float* dw, void* stream);

/* norm_input of every model's forward (models/model.py:247-252): out = x with its NON-ZERO entries replaced by
 * (x - mean) / std, mean and unbiased std taken over the non-zero entries of the whole tensor.  ws3: 3 doubles of
 * scratch.  out may alias x. */
int evf_norm_nonzero(const float* x, int64_t n, float* out, double* ws3, void* stream);

/* Prediction head: 1x1 conv 32->2 + bias + tanh (models/submodules.py:52-61,
 * model.py:197-199) on bit-packed spikes -> flow [B,2,H,W] (NCHW). */
int evf_pred_fwd(const uint32_t* x, const float* w, const float* bias, int B, int H, int W,
                 float* flow, void* stream);
/* backward: g_x [B,H,W,32] (written), dw [2,32] and dbias [2] accumulated */
int evf_pred_bwd(const uint32_t* x, const float* flow, const float* g_flow, const float* w,
                 int B, int H, int W, float* g_x, float* dw, float* dbias, void* stream);

/* spike bit maps <-> float tensors (state API, models/model.py:203-209) */
int evf_bits_to_nchw(const uint32_t* bits, int B, int H, int W, float* out, void* stream);
/* pixel-major spike words [B,H,W] -> channel-major bit planes [B,H,32,ceil(W/32)] */
int evf_bits_transpose(const uint32_t* bits, int B, int H, int W, uint32_t* planes, void* stream);
int evf_nchw_to_bits(const float* in, int B, int H, int W, uint32_t* bits, void* stream);
int evf_nhwc_to_nchw(const float* in, int B, int C, int H, int W, float* out, void* stream);
int evf_nchw_to_nhwc(const float* in, int B, int C, int H, int W, float* out, void* stream);

/* Hot-pixel mask of the loader applied in place to a batch of encodings (dataloader/h5.py:289-295):
 * x [B,C,H,W] *= mask [B,H,W] (one binary mask per batch slot, dataloader/base.py:224-243). */
int evf_apply_pixel_mask(float* x, const float* mask, int B, int C, int H, int W, void* stream);

/* Window masks: out [B,1,H,W] = min(sum_p masks[B,P,H,W], 1) (loss/flow.py:149-150); masked mean of the
 * per-pass flow maps [B,P,2,H,W] -> [B,2,H,W], sum_p maps*mask / (sum_p mask + 1e-9) (loss/flow.py:443-452). */
int evf_mask_union(const float* masks, int B, int P, int H, int W, float* out, void* stream);
int evf_masked_flow_mean(const float* maps, const float* masks, int B, int P, int H, int W, float* out,
                         void* stream);

/* ------------------------------------------------------------------ general path
 * Layers outside the fused 32->32 FireNet kernels: the spiking EV-FlowNet
 * (models/unet.py:418-465, spiking_submodules.py:878-1013), the ANN FireNet
 * (models/submodules.py:64-83,377-418), the 1x1 prediction heads
 * (submodules.py:12-61) and stand-alone Conv{LIF,PLIF,ALIF,XLIF}[Recurrent]
 * cells (spiking_submodules.py:24-875).  Activations are NHWC fp32 with an
 * explicit pixel stride (ld, in floats); k in {1,3,5,7} (models/unet.py:51 defaults to 5), stride in {1,2}, padding
 * k/2 (F.conv2d as used at spiking_submodules.py:99,519).  fp32 matrix-core
 * arithmetic (v_mfma_f32_32x32x2_f32): results equal an fp32 CPU convolution
 * up to summation order. */

/* Packed-weight size in floats for evf_pack_conv2d_weight (0 on bad arguments). */
int64_t evf_conv2d_packed_size(int Cout, int Cin, int ksz, int transpose);
/* w: torch layout [Cout][cin_total][k][k]; packs input channels cin_off..cin_off+Cin (channels past cin_total
 * are zero: alignment padding of the activation tensor).
 * transpose = 0 -> operand of evf_conv2d_fwd, 1 -> operand of evf_conv2d_dgrad. */
int evf_pack_conv2d_weight(const float* w, int Cout, int Cin, int ksz, int transpose, int cin_total,
                           int cin_off, float* dst, void* stream);
/* y [B,Ho,Wo,Cout] (+)= conv(x [B,H,W,Cin]) + bias  (nn.Conv2d.forward) */
int evf_conv2d_fwd(const float* x, int ldx, const float* w_packed, const float* bias, float* y, int ldy,
                   int B, int H, int W, int Cin, int Cout, int ksz, int stride, int accumulate,
                   void* stream);
/* g_x [B,H,W,Cin] (+)= conv^T(g_y [B,Ho,Wo,Cout])  (autograd of the above w.r.t. x) */
int evf_conv2d_dgrad(const float* g_y, int ldg, const float* wT_packed, float* g_x, int ldx, int B, int H,
                     int W, int Cin, int Cout, int ksz, int stride, int accumulate, void* stream);
/* The same two products on the bf16 matrix cores with exact operand splits (csrc/evf_conv_b3gen.hip; the host's default,
 * EVF_CONV=f32 keeps the fp32 kernels above): weights packed as three bf16 planes (w = hi + mid + lo exactly), the
 * activation / gradient converted on the fly; a wave whose 32 pixels x 16 channels are all exactly representable in bf16
 * (binary spikes, event counts, bilinear blends of spikes) issues 3 products, any other wave the 6 terms above 2^-24 of
 * the leading one.  fp32 accumulation; no promise needed from the caller, the vote never changes a result.
 * Same arguments as the fp32 entry points; packed size in floats. */
int64_t evf_conv2d_b3_packed_size(int Cout, int Cin, int ksz, int transpose);
int evf_pack_conv2d_weight_b3(const float* w, int Cout, int Cin, int ksz, int transpose, int cin_total,
                              int cin_off, void* dst, void* stream);
/* The same for n tensors in one launch (host arrays of device pointers; meta: 6 ints per tensor = Cout, Cin, ksz,
 * transpose, cin_total, cin_off): every packed operand of a network after an optimizer step. */
int evf_pack_conv2d_weights_b3_multi(const void* const* w, void* const* dst, const int* meta, int n, void* stream);
/* ws: optional scratch of ws_floats floats (evf_conv2d_b3_ws() of the OUTPUT shape; 0 = this shape never needs it): layers
 * whose output tiles alone cannot fill the chip split their contraction into up to 8 slabs, summed in a fixed order
 * (deterministic; no atomics).  Null = never split.
 * evf_conv2d_fwd_b3, accumulate bit 2 (value 4) = "x is exactly representable in bf16 BY CONSTRUCTION from channel
 * (accumulate >> 4) & 31 (<= 16) on" (spikes, small sums of spikes, their bilinear x2 blends behind a decoder's two flow channels;
 * models/hip_ops.py spike_tag): 3x3 stride-1 products then take the single-plane kernel of csrc/evf_conv_b3small.hip (64 input
 * channels of the halo tile staged ONCE in LDS as one bf16 plane for all nine taps, the first 16 channels as the exact 3-way split
 * when they hold real values); an x that breaks the promise yields NaN there, never a rounded product.  Without the bit nothing
 * changes (the kernels vote per wave). */
int64_t evf_conv2d_b3_ws(int B, int Ho, int Wo, int Cout);
int evf_conv2d_fwd_b3(const float* x, int ldx, const void* w_packed, const float* bias, float* y, int ldy,
                      int B, int H, int W, int Cin, int Cout, int ksz, int stride, int accumulate,
                      float* ws, int64_t ws_floats, void* stream);
int evf_conv2d_dgrad_b3(const float* g_y, int ldg, const void* wT_packed, float* g_x, int ldx, int B, int H,
                        int W, int Cin, int Cout, int ksz, int stride, int accumulate, float* ws,
                        int64_t ws_floats, void* stream);
/* A 1x1 layer with <= 4 outputs in ONE pass each way (the tanh flow prediction of every scale: models/submodules.py:52-61,
 * models/unet.py:355-369): y[m][0..Cout) = act(W x[m] + bias) and its backward -- g_pre = g_y * act'(y); g_x = W^T g_pre; g_w, g_bias
 * summed over the M pixels (per-block partials in ws, evf_head1x1_ws() floats; accumulate: += instead of =).  W [Cout][Cin] is the
 * torch weight of the 1x1 conv; Cin / 4 a power of two <= 64; act 0 none / 1 tanh / 2 sigmoid / 3 relu; g_y as [M][Cout] rows or,
 * with gy_nchw_hw = H*W, as NCHW planes [b][Cout][H*W] (how the loss hands the flow maps' gradient back). */
int64_t evf_head1x1_ws(int Cin, int Cout);
int evf_head1x1_fwd(const float* x, int ldx, const float* w, const float* bias, int act, int64_t M, int Cin, int Cout,
                    float* y, int ldy, void* stream);
int evf_head1x1_bwd(const float* x, int ldx, const float* y, int ldy, const float* g_y, int64_t gy_nchw_hw, const float* w,
                    int act, int64_t M, int Cin, int Cout, float* g_x, int ldgx, float* g_w, float* g_bias, int accumulate,
                    float* ws, void* stream);
/* evf_conv2d_fwd_b3 without bias / accumulation whose K-split partial sums stay IN PARTS for the consumer (evf_lif_fwd_parts):
 * *nparts = 0: y holds the result; n > 0: ws holds n slabs [B*Ho*Wo][Cout] to be added in index order.  flags: bit 2 as above. */
int evf_conv2d_fwd_b3_parts(const float* x, int ldx, const void* w_packed, float* y, int ldy, int B, int H, int W,
                            int Cin, int Cout, int ksz, int stride, int flags, float* ws, int64_t ws_floats,
                            int* nparts, void* stream);
/* LIF update (evf_neuron_fwd, kind LIF: spiking_submodules.py:96-126, :516-551) on a current given in parts:
 * cur = sum_{z < na} a[z * a_stride + .] + sum_{z < nb} b[z * b_stride + .] (b null: no recurrent part; strides in floats). */
int evf_lif_fwd_parts(const float* a, int na, int64_t a_stride, const float* b, int nb, int64_t b_stride,
                      const float* v_prev, const float* z_prev, const float* residual, const float* leak,
                      const float* thresh, int64_t npix, int C, int hard_reset, float* v_out, float* z_out,
                      float* out, void* stream);
/* Kernel choice behind the two entry points above for 3x3 stride-1 products: -1 by shape (default: the spatially tiled
 * kernel of csrc/evf_conv_b3tile.hip for wide high-resolution layers, environment EVF_CONV_TILE=0|2 at load), 0 never,
 * 2 whenever the operands are aligned for it.  Same results up to summation order. */
int evf_conv_tile_select(int mode);
/* K splits of the two entry points above: 0 by shape (default; environment EVF_CONV_SPLIT=n at load), n > 0 forces n
 * splits wherever the caller's scratch allows (tests). */
int evf_conv_split_select(int n);
/* 3x3 stride-1 weight gradients with > 32 input and output channels: the two-team kernel k_wgrad9_b3v (csrc/evf_wgrad_b3gen.hip:
 * four matrix waves with nine taps each + four loader waves) -- 0 never (default; environment EVF_WGRAD_TEAMS at load: measured
 * level with the tap-per-wave kernel k_wgrad9_b3 inside a train step), 1 where a block walks >= 16 pixel tiles, 2 wherever its
 * block shape fits (tests).  Bit-identical slabs. */
int evf_wgrad_teams_select(int mode);
/* g_w [Cout][cin_total][k][k] (input channels cin_off ..) and optional g_bias [Cout]
 * (autograd w.r.t. weight / bias).  accumulate = 0 overwrites the outputs and needs cin_off = 0 and
 * Cin >= cin_total (channels past cin_total are activation padding and are skipped).  ws: scratch of evf_conv2d_wgrad_ws() floats (3x3, and 1x1 with Cout <= 4; null is accepted for 1x1 and selects the atomic split-K kernel):
 * partial sums of the pixel splits, reduced without atomics. */
/* 3x3 stride 1: the contraction over pixels runs on the bf16 matrix cores first (csrc/evf_wgrad_b3gen.hip: x as one bf16
 * plane, g_y as three; exact for spike-valued x), and the fp32 kernel recomputes only the input-channel tiles whose x was
 * not exactly representable.  accumulate bit 1 (value 2) = "x is not spike-valued": fp32 kernel only.  accumulate bit 2
 * (value 4) = "x IS exactly representable in bf16 by construction" (spikes, small residual sums of spikes, bilinear x2 blends
 * of those): no fp32 verification pass (its launch would exit at once in every block); with EVF_WGRAD_FUSE=1 in the environment
 * also no k_wgrad_reduce launch for <= 8 pixel splits -- the bf16 kernel's last block per weight tile sums the splits in index
 * order (measured slower than the reduce launch, hence off).  An x that breaks the promise yields NaN in the gradient, never a
 * silently rounded one.  EVF_WGRAD=f32 in the environment disables the bf16 kernel. */
int64_t evf_conv2d_wgrad_ws(int B, int H, int W, int Cin, int Cout, int ksz, int stride);
int evf_conv2d_wgrad(const float* x, int ldx, const float* g_y, int ldg, float* g_w, float* g_bias, int B,
                     int H, int W, int Cin, int Cout, int ksz, int stride, int cin_total, int cin_off,
                     int accumulate, float* ws, void* stream);

/* Batch / instance normalisation of NHWC activations (nn.BatchNorm2d / nn.InstanceNorm2d(track_running_stats=True) of the
 * reference's ANN layers, models/submodules.py:46-56, 122-132, 169-180, 273-301) = two per-element passes; the per-channel
 * arithmetic in between is host-side (a few hundred floats).  Statistics group g covers npg consecutive pixels (batch norm:
 * G = 1, npg = B*H*W; instance norm: G = B, npg = H*W); tensors [G*npg][C], pixel strides ld*.
 *   evf_chan_reduce: out[g*C+c] = sum_pixels { mode 0: x | 1: (x - center)^2 | 2: y * (x - center) * scale }
 *   evf_chan_affine: out = (g ? A*g : 0) + Bc*x + Cc with per-(group, channel) coefficients [G*C]. */
int evf_chan_reduce(const float* x, int ldx, const float* y, int ldy, const float* center, const float* scale, int mode,
                    int G, int64_t npg, int C, float* out, void* stream);
/* Weight normalisation of a conv weight [Cout][n = Cin * k * k] (norm = "weight" cells: models/spiking_submodules.py:87-88,
 * :502-504, nn.utils.weight_norm with dim = 0):  w[o] = v[o] * g[o] / ||v[o]||; nrm [Cout] keeps the norms for the backward, which
 * maps dL/dw to dL/dv [Cout][n] and dL/dg [Cout]. */
int evf_weight_norm_fwd(const float* v, const float* g, int Cout, int n, float* w, float* nrm, void* stream);
int evf_weight_norm_bwd(const float* gw, const float* v, const float* g, const float* nrm, int Cout, int n,
                        float* gv, float* gg, void* stream);
int evf_chan_affine(const float* g, int ldg, const float* x, int ldx, const float* A, const float* Bc, const float* Cc,
                    int G, int64_t npg, int C, float* out, int ldo, void* stream);

/* Neuron update of one spiking cell on a precomputed input current `cur`
 * (ff [+ rec] conv), all tensors [npix][C] fp32, C % 4 == 0, null previous
 * state = zeros.  kind = EVF_LIF / PLIF / ALIF / XLIF; per-channel parameters:
 *   LIF : p0 leak     p1 thresh                      (spiking_submodules.py:104-123)
 *   PLIF: p0 leak_v   p1 thresh  p2 leak_pt p3 add_pt (:199-224)  aux = pt, needs P
 *   ALIF: p0 leak_v   p1 t0      p2 t1      p3 leak_t (:307-331)  aux = t
 *   XLIF: p0 leak_v   p1 t0      p2 t1      p3 leak_pt(:407-432)  aux = pt, needs P
 * P [npix] = evf_pretrace_fwd of the cell input.  out (optional) = z_out + residual. */
int evf_neuron_fwd(int kind, const float* cur, const float* v_prev, const float* z_prev,
                   const float* aux_prev, const float* P, const float* residual, const float* p0,
                   const float* p1, const float* p2, const float* p3, int64_t npix, int C, int hard_reset,
                   float* v_out, float* z_out, float* aux_out, float* out, void* stream);
/* Surrogate-gradient backward of evf_neuron_fwd (spiking_util.py:24-25,38-93; the reset
 * uses z detached, spiking_submodules.py:116-120).  Upstream gradients (any may be null):
 * g_v_out, g_z_out + g_z_out2, g_aux_out.  Writes g_cur, g_v_prev, g_aux_prev (kind != LIF),
 * g_z_prev (ALIF only: its threshold trace integrates z), g_P [npix] (PLIF/XLIF); parameter
 * gradients g_p0..g_p3 [C] are ACCUMULATED (null = skip).  g_v_prev null = the previous state takes no gradient
 * (first pass of a window, detached state): g_v_prev / g_aux_prev / g_z_prev are not written.  Absent operand groups
 * (no upstream state gradient, no previous state) select kernel variants without their loads.
 * ws: optional scratch of EVF_NEURON_BWD_WS floats, ZERO on entry and left zero on exit: the blocks' parameter-gradient
 * sums meet in 32 replicas there instead of 1024 blocks adding atomically into the same 2..4 x C words (null: they do); the
 * block that finishes last sums the replicas into the outputs (arrival ticket in the word behind the replicas).  Launches of
 * <= 256 blocks (at least ~8 float4 per thread) add straight into the outputs and leave the scratch alone. */
#define EVF_NEURON_BWD_WS (32 * 4096 + 64)
int evf_neuron_bwd(int kind, const float* g_v_out, const float* g_z_out, const float* g_z_out2,
                   const float* g_aux_out, const float* v_out, const float* aux_out, const float* v_prev,
                   const float* z_prev, const float* aux_prev, const float* P, const float* p0,
                   const float* p1, const float* p2, const float* p3, int64_t npix, int C, int hard_reset,
                   int surrogate, float act_width, float* g_cur, float* g_v_prev, float* g_z_prev,
                   float* g_aux_prev, float* g_P, float* g_p0, float* g_p1, float* g_p2, float* g_p3,
                   float* ws, void* stream);
/* P [B,Ho,Wo] = avg_pool2d(mean_c |x|, k, stride, k/2) (spiking_submodules.py:212,418);
 * absmean_ws [B*H*W] workspace.  Backward adds/writes sign(x)/C * pool^T(g_P) into g_x. */
int evf_pretrace_fwd(const float* x, int ldx, int B, int H, int W, int C, int ksz, int stride,
                     float* absmean_ws, float* P, void* stream);
int evf_pretrace_bwd(const float* x, int ldx, const float* g_P, int B, int H, int W, int C, int ksz,
                     int stride, float* g_x, int ldg, int accumulate, void* stream);

/* F.interpolate(scale_factor=2, mode="bilinear", align_corners=False) on NHWC
 * (spiking_submodules.py:1011) and its adjoint. */
int evf_upsample2x_fwd(const float* x, int B, int H, int W, int C, float* y, void* stream);
int evf_upsample2x_bwd(const float* g_y, int B, int H, int W, int C, float* g_x, void* stream);
/* Channel concatenation of n <= 6 NHWC activations into out (pixel stride ldo): torch.cat(..., 1) of the decoder inputs
 * (models/unet.py:303-306, model_util.py:14-19).  src[k] null = C[k] channels of zeros (alignment padding). */
int evf_concat_channels(const void* const* src, const int* C, const int* ld, int n, int64_t npix, float* out, int ldo,
                        void* stream);
/* The concatenation above READ THROUGH a bilinear x2 up-sampling (a decoder's input, spiking_submodules.py:1011 on the cat of
 * models/unet.py:303-306): parts [B,H,W,C[k]] (pixel stride ld[k]) -> out [B,2H,2W,sum C] without writing the low-resolution
 * concatenation.  All C[k], ld[k] even, sum C and ldo multiples of four (a thread stores four channels read as two pairs); a null
 * part (not the first) = zero padding. */
int evf_concat_up2_fwd(const void* const* src, const int* C, const int* ld, int n, int B, int H, int W, float* out,
                       int ldo, void* stream);
/* F.interpolate(scale_factor=f) (nearest) of `planes` images [h][w] -> [h f][w f]
 * (models/model.py:529-539) and its adjoint. */
int evf_upsample_nearest_fwd(const float* x, int64_t planes, int h, int w, int factor, float* y,
                             void* stream);
int evf_upsample_nearest_bwd(const float* g_y, int64_t planes, int h, int w, int factor, float* g_x,
                             void* stream);

/* y = act(x [+ residual]); kind 0 identity, 1 tanh, 2 sigmoid, 3 relu
 * (submodules.py:52-61,78-81).  Backward through the OUTPUT y: g_x = g_y act'(y). */
#define EVF_ACT_NONE 0
#define EVF_ACT_TANH 1
#define EVF_ACT_SIGMOID 2
#define EVF_ACT_RELU 3
int evf_act_fwd(int kind, const float* x, const float* residual, int64_t n, float* y, void* stream);
int evf_act_bwd(int kind, const float* y, const float* g_y, int64_t n, float* g_x, void* stream);

/* Leaky non-spiking cells (ANN comparisons): ConvLeaky models/submodules.py:502-554, ConvLeakyRecurrent :454-499.
 *   mix = prev * sigmoid(leak[c]) + (1 - sigmoid(leak[c])) * (cur [+ residual]);   out = act(mix)   (act as evf_act_fwd)
 * NHWC fp32 [npix, C], C % 4 == 0; prev / residual null = zeros; out null = not written.  Backward (saved: mix, prev):
 *   g_mix = g_state + g_out * act'(out);  g_cur = g_mix (1 - lam) (also the residual's gradient);  g_prev = g_mix lam
 *   (null = not needed);  g_leak[C] is ADDED to (null = not needed).  g_out / g_state: either may be null. */
int evf_leaky_fwd(const float* cur, const float* prev, const float* residual, const float* leak, int act, int64_t npix, int C,
                  float* mix, float* out, void* stream);
int evf_leaky_bwd(const float* g_out, const float* g_state, const float* mix, const float* prev, const float* leak, int act,
                  int64_t npix, int C, float* g_cur, float* g_prev, float* g_leak, void* stream);

/* Stand-alone spike functions (models/spiking_util.py:13-25 forward, :38-93 surrogates): z = (x - thresh > 0)
 * as fp32; g_x = g * surrogate(x - thresh, width) with surrogate = EVF_ARCTAN / SUPERSPIKE / TRIANGLE / MULTIGAUSS.
 * thresh: one scalar (thresh_per_element = 0) or one value per element of x (1). */
int evf_spike_fwd(const float* x, const float* thresh, int thresh_per_element, int64_t n, float* z, void* stream);
int evf_spike_bwd(int surrogate, const float* x, const float* thresh, int thresh_per_element, const float* g,
                  float width, int64_t n, float* g_x, void* stream);

/* ConvGRU gate algebra (submodules.py:404-416): u = sigmoid(cu), r = sigmoid(cr), hr = h r;
 * o = tanh(co), h_new = h (1 - u) + o u.  h null = zeros.  Backward: evf_gru_out_bwd writes
 * g_co, g_cu (pre-activation) and g_h = g_new (1-u); evf_gru_gates_bwd writes g_cr and ADDS
 * g_hr r into g_h. */
int evf_gru_gates_fwd(const float* cu, const float* cr, const float* h, int64_t n, float* u, float* r,
                      float* hr, void* stream);
int evf_gru_out_fwd(const float* co, const float* h, const float* u, int64_t n, float* o, float* h_new,
                    void* stream);
int evf_gru_out_bwd(const float* g_new, const float* h, const float* u, const float* o, int64_t n,
                    float* g_co, float* g_cu, float* g_h, void* stream);
int evf_gru_gates_bwd(const float* g_hr, const float* h, const float* r, int64_t n, float* g_cr,
                      float* g_h, void* stream);

/* ConvLSTM gate algebra (models/submodules.py:357-374).  gates [npix, 4*Ch] NHWC = Gates(cat(x, hidden)) with the
 * channel chunks in | remember | out | cell (gates.chunk(4, 1)):  i, r, o = sigmoid, cg = tanh;
 * cell' = r * prev_cell + i * cg;  hidden = o * tanh(cell').  evf_lstm_fwd overwrites `gates` with the activated
 * gates (the backward's saved tensor); prev_cell null = zeros.  evf_lstm_bwd: g_gates = d loss / d (pre-activation
 * gates), g_prev_cell (null = not needed); g_hidden / g_cell: either may be null. */
int evf_lstm_fwd(float* gates, const float* prev_cell, int64_t npix, int Ch, float* cell, float* hidden, void* stream);
int evf_lstm_bwd(const float* g_hidden, const float* g_cell, const float* gates, const float* cell, const float* prev_cell,
                 int64_t npix, int Ch, float* g_gates, float* g_prev_cell, void* stream);

/* ------------------------------------------------------------------ optimiser
 * train_flow.py:157-163: clip_grad_norm_(max_norm) + Adam(lr) on one flat
 * parameter buffer.  norm_ws [2] float workspace: [0] receives the squared
 * gradient norm, [1] is a device-side step counter.  step >= 1: bias corrections
 * from the host value.  step <= 0: the kernel advances norm_ws[1] and uses it
 * (zero it once at start) -- needed when the step is replayed from a hipGraph.
 * zero_grad != 0: the gradient buffer is cleared as it is consumed (optimizer.zero_grad(), train_flow.py:164, without a
 * fill kernel of its own). */
int evf_clip_adam_step(float* param, float* grad, float* m, float* v, int64_t n,
                       float max_norm, float lr, float beta1, float beta2, float eps, int step,
                       float* norm_ws, int zero_grad, void* stream);
/* The same step in ONE launch for parameter counts up to 2^20 (a few fat blocks; every block sums the whole gradient itself --
 * a reproducible norm without atomics; no block waits for another: the block that draws the last departure ticket clears the
 * gradient, publishes the norm and advances the counter; larger n: the two launches above).
 * ws: >= 8 floats, zeroed ONCE by the caller and owned by this entry point afterwards: [0] squared gradient norm of the last
 * step, [1] the device-side step counter (as norm_ws[1] above), [4] the departure ticket, left zero ([3] unused). */
int evf_clip_adam_fused(float* param, float* grad, float* m, float* v, int64_t n,
                        float max_norm, float lr, float beta1, float beta2, float eps, int step,
                        float* ws, int zero_grad, void* stream);
/* Every partial sum a window's backward leaves behind, added to the (flat) parameter gradients in one launch
 * (evf_reduce_slabs_multi + evf_sum_rows x 2 + evf_add_segments; train_flow.py:154 loss.backward()'s parameter gradients):
 *   slabs[t] [nslab][9*32*32] partial sums of conv weight t (nslabs <= 16) -> slab_dst[t] [32][32][3][3] +=;
 *   total[e] = small[e] + sum_r rows[r][e] (e < ncols; rows zeroed; rows null = none)
 *                       + sum_r head_rows[r][e - head_off] (head_off <= e < head_off + nhcols; null = none);
 *   seg_dst[k][i] += total[seg_off[k] + i], i < seg_n[k] (nseg <= 32); clear_small != 0: small[e] = 0 afterwards.
 * Every output element has ONE writer that sums in a fixed order: the result does not depend on the block schedule (segments
 * must not overlap).  Only the columns of `rows` that belong to a segment are read and zeroed.  seg_rows (may be NULL: all):
 * seg_rows[k] = the number of leading rows that can hold something for segment k -- the writers of a segment's columns are
 * launches of at most that many blocks, the rows behind are zero and are skipped. */
int evf_grads_finalize(const void* const* slabs, void* const* slab_dst, int nslabs, int nslab, float* small, int clear_small,
                       float* rows, int nrows, int ncols, const float* head_rows, int nhrows, int nhcols, int head_off,
                       void* const* seg_dst, const int* seg_off, const int* seg_n, const int* seg_rows, int nseg,
                       void* stream);

/* ---- data parallelism: the ONE collective of an optimizer step (SURVEY.md 8(b) `evf_allreduce_sum`, 8(e)) ----------------
 * The reference is single-process (/root/reference/configs/parser.py:83-86, train_flow.py:98-171); its loss SUMS over the batch
 * (loss/flow.py:226,259,289), so N ranks holding N shards of the batch need exactly one in-place SUM all-reduce of the flat
 * gradient buffer [gradient | loss | new_seq flag] per step, before clip + Adam (evf_clip_adam_*), over RCCL / xGMI.
 * evf_comm_*: a communicator of this library's own (RCCL bound at run time by dlopen; EVF_ENOTSUP when absent).
 *   evf_comm_load(path)      bind RCCL from `path` (NULL / "": a librccl this process has loaded already, else the system's);
 *   evf_comm_unique_id(id)   rank 0: 128 bytes (EVF_COMM_ID_BYTES) to hand to every rank out of band (torch's store, a file);
 *   evf_comm_init(id, rank, world, &comm)   collective over the ranks (blocks until all have called), current HIP device;
 *   evf_allreduce_sum / _max(comm, buf, n, stream)   in place, fp32, enqueued on `stream`; CAPTURABLE into a hipGraph (one
 *                            kernel node), which is what makes the N-rank step ONE graph;
 *   evf_comm_destroy(comm); evf_comm_version(&v) (RCCL's version code); evf_comm_last_error() (text of the last RCCL error).
 * Status: 0, EVF_EINVAL, EVF_ENOTSUP, or -(2000 + ncclResult_t). */
#define EVF_COMM_ID_BYTES 128
int evf_comm_load(const char* librccl_path);
const char* evf_comm_last_error(void);
int evf_comm_version(int* version);
int evf_comm_unique_id(void* id128);
int evf_comm_init(const void* id128, int rank, int world, void** comm);
int evf_comm_count(void* comm, int* ranks); /* ncclCommCount: the ranks RCCL sees on `comm` */
int evf_comm_destroy(void* comm);
int evf_allreduce_sum(void* comm, float* buf, int64_t n, void* stream);
int evf_allreduce_max(void* comm, float* buf, int64_t n, void* stream);

/* memset as a kernel launch: `bytes` bytes at `dst` (4-byte aligned) set to the byte `value`.  Inside a captured step
 * (hipGraph) a hipMemsetAsync becomes a memset node; on ROCm 7.2 graphs holding such nodes between kernel nodes replayed with
 * corrupted results after a hipDeviceSynchronize (packet-captured kernel nodes; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 hides it).
 * The library and its Python host clear every buffer with this kernel instead. */
int evf_memset(void* dst, int value, size_t bytes, void* stream);

/* Debugging aid (csrc/evf_debug.hip): fills the LDS of every CU with `pattern` (0: a quiet NaN), so that a kernel reading an LDS
 * word it never wrote shows it in its output.  EVF_DEBUG_POISON_LDS=1 makes the Python host call it in front of every entry point. */
int evf_debug_poison_lds(uint32_t pattern, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EVFLOW_H */

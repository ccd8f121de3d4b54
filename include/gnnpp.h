/*
 * gnnpp.h -- C ABI of libgnnpp.so: the MI355X (gfx950) implementation of the GNN policy forward
 * pass of proroklab/gnn_pathplanning (DecentralPlannerNet + GraphFilter / GraphFilterBatch).
 *
 * The reference has no FFI for this path (it is 100 % stock aten calls); the entry points below
 * are what a binding of the reference's Python call sites needs.  Every function cites the
 * reference interface it replaces (paths relative to the upstream repo root).
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless marked "host"; the caller owns every buffer;
 *     inputs are never written;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); every call only
 *     enqueues work on that stream and returns without synchronising;
 *   - return value 0 = success, negative = error (gnnpp_error_string()); a failed call has
 *     enqueued nothing;
 *   - fp32 everywhere; a GSO may be given as fp64 and is rounded to fp32 on load exactly like the
 *     reference's `S.float()` (utils/graphUtils/graphML.py:2350);
 *   - every forward entry point takes `precision` (GNNPP_PREC_*): the arithmetic of its matrix-pipe
 *     contractions, chosen PER CALL (no process-wide state: two streams / threads cannot change each other's
 *     schedule).  0 = GNNPP_PREC_FP32 is what an unchanged caller of the reference gets.
 */
#ifndef GNNPP_H_
#define GNNPP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNNPP_OK               0
#define GNNPP_ERR_ARG         (-1)   /* null pointer / non-positive size / inconsistent flags   */
#define GNNPP_ERR_UNSUPPORTED (-2)   /* shape outside what the kernels cover (see each call)     */
#define GNNPP_ERR_LAUNCH      (-3)   /* HIP launch error                                         */
#define GNNPP_ERR_RANGE       (-4)   /* GNNPP_PREC_SPLIT_F16 only: an activation left the f16 range  */

/* Arithmetic of the matrix-pipe contractions (encoder convolutions + compressMLP, the filter's tap
 * contraction).  Graph shifts, BatchNorm / bias / ReLU / pooling epilogues and the action head are exact fp32
 * in every mode; accumulation is fp32 in every mode.
 *   GNNPP_PREC_FP32 (default)  fp32-equivalent, NO input domain: every fp32 operand is represented exactly as
 *       three bf16 planes (x = h + m + l, fp32's exponent range) and a product keeps six of the nine plane
 *       products on v_mfma_f32_16x16x32_bf16 (dropped terms <= 2^-23 |w x|, below one fp32 rounding).  Where
 *       a kernel has no LDS room for the planes -- the general graph-filter kernel (lsigf_kernel) and the
 *       policy filter of teams whose rows + planes exceed the 160 KB LDS (N > ~64 agents, e.g. the 100-agent
 *       configuration) -- the contraction runs on the exact fp32 MFMA instead: exact as well, only slower.
 *   GNNPP_PREC_FP32_MFMA       v_mfma_f32_16x16x4_f32 everywhere: bitwise an fmaf chain, 2.7x more pipe time.
 *   GNNPP_PREC_SPLIT_F16       fast, NARROWER than fp32: operands as f16 hi + lo halves (22 significand bits)
 *       on v_mfma_f32_16x16x32_f16, valid for |activation| < 65504 only -- see "Range guard"; opt-in. */
#define GNNPP_PREC_FP32        0
#define GNNPP_PREC_FP32_MFMA   1
#define GNNPP_PREC_SPLIT_F16   2

#define GNNPP_OBS_C        3         /* observation channels      (decentralplanner.py:89)       */
#define GNNPP_OBS_HW       11        /* observation height=width  (decentralplanner.py:22-23)    */
#define GNNPP_FEAT         128       /* numFeatures2Share         (decentralplanner.py:93,197)   */
#define GNNPP_ACTIONS      5         /* numAction                 (decentralplanner.py:27)       */
#define GNNPP_MAX_NODES    100       /* N <= 100 nodes per graph always fits (G, F <= 128)       */
#define GNNPP_MAX_ROWS     112       /* hard limit: a workgroup keeps <= 112 node rows in LDS; N in
                                        101..112 works when the 160 KB LDS budget allows (narrower
                                        G / F), else GNNPP_ERR_UNSUPPORTED.  Larger graphs: the same
                                        filter as dense GEMMs, one gnnpp_gemm_kmajor call per shift
                                        (A = S with strides (1, N), B = z_{k-1}) and one for the tap
                                        contraction -- what graphML._lsigf_large does              */

int         gnnpp_version(void);
const char* gnnpp_error_string(int code);

/* Process-wide tuning knobs (atomic; a concurrent call sees the old or the new value).  They choose between
 * SCHEDULES of the same arithmetic (launch geometry, kernel fusion): every setting computes the same function
 * to the last bit or two.  The arithmetic itself is the per-call `precision` argument, never a knob.  Knobs
 * that skip kernel phases for profiling are not part of this ABI (csrc/gnnpp_measure.h, -DGNNPP_MEASURE
 * builds only). */
#define GNNPP_TUNE_FILTER_GPW      1  /* graphs per workgroup of the filter kernel; 0 = heuristic */
#define GNNPP_TUNE_FILTER_WAVES    2  /* waves per workgroup of the filter kernel: 8, 16; 0 = auto */
#define GNNPP_TUNE_FILTER_SPLIT    7  /* 0 = heuristic (when one-graph workgroups fill at most half of the
                                         256 CUs: 2 .. 7 workgroups per graph, as many as keep one workgroup
                                         per CU, at most one per 16-row tile -- v320; v310: two at most);
                                         1 = never; n = 2 .. 7: n parts (clamped to the row tiles) whenever
                                         a workgroup holds one graph with >= 2 row tiles            */
#define GNNPP_TUNE_FUSED_POLICY     6  /* 1 (default): for teams of N <= 16 agents and K = 2, 3 or 4 taps
                                         (GNNPP_PREC_FP32 or GNNPP_PREC_SPLIT_F16), when B <= 512 graphs
                                         or N >= 13, gnnpp_policy_fwd is ONE kernel -- a workgroup
                                         encodes one graph's agents, then runs that graph's filter and
                                         action head on chip (identical logits); 0: always the encoder
                                         kernel followed by the filter kernel; 2: the one kernel for
                                         every batch size (measurement: tools/cp_ab.py)              */
#define GNNPP_TUNE_POLICY_FILTER    9  /* 1 (default): the filter + action head of gnnpp_policy_fwd / the rollout step
                                         for teams of 17 .. 100 agents (one graph per workgroup,
                                         FILTER_WAVES != 8) runs on the latency-scheduled policy_filter_kernel;
                                         0: on the general filter kernel (same logits to the last bit or two).
                                         (v310's value 2 -- bf16x3 planes beside compact neighbour lists for 65 ..
                                         100 agents, measured no faster -- was removed in v320: GNNPP_ERR_ARG)   */
#define GNNPP_TUNE_FILTER_SMALL     10  /* 1 (default): graph filters over many small graphs (GNNPP_PREC_FP32, N <= 16,
                                         G = F = 128, node-major rows, >= 64 workgroups) run on the
                                         throughput kernel lsigf_small_b3_kernel (bf16x3 planes, two workgroups per
                                         CU); 0: on the general filter kernel; 2: whenever the shape fits,
                                         however few graphs (tests); 3: the pipeline kernel whenever it fits */
#define GNNPP_TUNE_FILTER_SMALL_ROWS 11 /* rows per workgroup of that kernel: 0 = heuristic, 32 or 48 (fewer tap bytes per
                                         agent-step); 64 = the producer / consumer pipeline kernel
                                         (lsigf_pipe_b3_kernel: one persistent 8-wave workgroup per CU -- the heuristic
                                         takes it from 4096 groups of 64 rows on; FILTER_SMALL = 3 forces it too)  */
#define GNNPP_TUNE_FILTER_PIPE_GRID 12  /* persistent workgroups of the pipeline kernel: 0 (default) = one per CU */
#define GNNPP_TUNE_POLICY_CP        13  /* 1 (default): the one-launch policy kernel of teams of <= 12 agents puts (agent,
                                         position) pairs on the MFMA columns in its 5x5 layers (a 10-agent graph then
                                         fills 15/16 of those tiles instead of 10/16); 0 = agents on the columns for
                                         every team size.  Same logits to the bit either way (v310)                     */
#define GNNPP_TUNE_ENCODER_CP_TILE   14  /* agents per tile of the (unfused) encoder kernel's column-packed form: 0 (default)
                                         = heuristic -- M <= 2048 agents: tiles of ceil(M / 256) agents, one per CU
                                         (latency regime), else 16-agent tiles --, 1 .. 12 = that tile size for every
                                         M, 16 = always 16-agent tiles.  Same features to the bit (v310)                */
#define GNNPP_TUNE_TRAIN_FORK        15  /* gnnpp_encoder_train_bwd can run the weight-gradient kernels of every layer
                                         on a second HIP stream that forks behind the layer's BatchNorm backward and
                                         joins before the call hands the stream back (graph edges under capture):
                                         1 (default) = from 4096 agent-samples (N x B) on -- measured + 7 % eager at
                                         512 x 10, - 16 % at 64 x 10 --, 0 = never, 2 = always.  Same gradients to the
                                         bit (v320)                                                                  */
#define GNNPP_TUNE_FILTER_PLANE_ALIAS 16 /* 1 (default): the filter + head launch of teams whose bf16x3 operand planes do
                                         not fit the LDS beside the graph (65 .. 100 agents) keeps the planes in the z buffer
                                         that is dead while a tap is contracted -- possible whenever the graph is split
                                         over >= 2 workgroups -- instead of contracting on the exact fp32 MFMA (2.7x the
                                         matrix-pipe time); 0 = r05's behaviour.  Same accuracy class (v330)              */
#define GNNPP_TUNE_TRAIN_WGRAD_WGS   17  /* workgroups per layer of the training step's weight-gradient kernel (image splits x
                                         output-channel tiles): 0 (default) = by the batch (128 up to 1 280 agent-samples, 256
                                         beyond), or 16 .. 2048: more splits = shorter workgroups, more partial sums to add.
                                         Set before gnnpp_encoder_train_workspace_floats / _train_fwd: the workspace size
                                         depends on it.  Same gradients to rounding (v330)                              */
#define GNNPP_TUNE_TRAIN_WGRAD_MERGED 18 /* 1 (default): gnnpp_encoder_train_bwd computes the weight gradients of all five
                                         layers in ONE launch behind the backward chain (nothing downstream needs them before
                                         the optimizer; at 64 x 10 the five per-layer launches are 65 us of latency, their
                                         MFMAs 14 us); 0: one launch per layer inside the chain, where GNNPP_TUNE_TRAIN_FORK
                                         can move them to a second stream.  Same gradients to the bit (v330)               */
#define GNNPP_TUNE_TRAIN_RUNNING_FUSED 19 /* 1 (default): gnnpp_encoder_train_fwd updates the BatchNorm running statistics inside
                                         its last BatchNorm launch (the workgroups of a channel tile publish their statistics
                                         with an agent-scope release / ticket / acquire hand-off and the last one to arrive
                                         runs the N updates); 0: a launch of its own (r05).  Same statistics to the bit (v330) */
int         gnnpp_set_tuning(int key, int value);
int         gnnpp_get_tuning(int key);   /* current value of a knob; GNNPP_ERR_ARG for an unknown key */

/* ------------------------------------------------------------------------------------------
 * Graph filter (LSIGF):  y = bias + sum_e sum_k W[:,e,k,:] . (x S_e^k),   z_k = z_{k-1} S (right
 * multiplication: node n gathers from the non-zeros of COLUMN n of S).
 * Replaces LSIGF (utils/graphUtils/graphML.py:48-141) and BatchLSIGF (:2273-2367), and with them
 * GraphFilter.forward (:1200-1219) / GraphFilterBatch.forward (:2458-2477).
 * ------------------------------------------------------------------------------------------ */

/* Number of floats of the MFMA-fragment-ordered copy of the filter taps h[F,E,K,G]. */
size_t gnnpp_filter_packed_floats(int G, int F, int K, int E);

/* Re-order h[F,E,K,G] (the nn.Parameter `weight`, graphML.py:1175 / :2434) into `packed`.
 * Call again whenever the parameter changes (load_state_dict, optimizer step). */
int gnnpp_filter_pack(const float* h, float* packed, int G, int F, int K, int E, void* stream);

/*
 * x        [B,G,Nin] feature-major (the module API, graphML.py:2459) when x_node_major == 0,
 *          [B,N,G]   node-major when x_node_major == 1 (requires Nin == N);
 * S        [B,E,N,N] when s_batched == 1 (BatchLSIGF), [E,N,N] when 0 (LSIGF, shared by the batch);
 *          element type double when s_is_f64 != 0, else float;
 * packed   from gnnpp_filter_pack;  bias [F] (bias_per_node == 0), [F,N] one value per feature and
 *          node (bias_per_node != 0; the reference's `b` is F x N or F x 1, graphML.py:2300-2302,
 *          :2365-2366), or NULL;
 * y        [B,F,Nin] when y_node_major == 0, [B,N,F] when 1;
 * Nin <= N: nodes Nin..N-1 of x are zero and the corresponding outputs are dropped
 *          (zero padding + index_select of graphML.py:1206-1218 / :2464-2476);
 * relu     apply max(.,0) to y (GFL[1], decentralplanner.py:221).
 * Limits:  1 <= N <= GNNPP_MAX_NODES at G,F <= 128 (LDS footprint); K >= 1; any F
 *          (more than 128 output features run as ceil(F/128) launches inside the call).
 * precision   GNNPP_PREC_*; range_flag  optional DEVICE int (NULL = no check), see "Range guard" below.
 */
int gnnpp_lsigf_fwd(const float* x, const void* S, const float* packed, const float* bias,
                    float* y, int B, int N, int Nin, int G, int F, int K, int E,
                    int s_is_f64, int s_batched, int x_node_major, int y_node_major, int relu,
                    int bias_per_node, int precision, int* range_flag, void* stream);

/* Range guard (GNNPP_PREC_SPLIT_F16 only; the other modes have no input domain and never touch the flag).
 * The split-f16 schedules feed the f16 matrix pipe with fp32 operands split in hi + lo halves (22 significand
 * bits), which is exact to ~2^-22 as long as every activation satisfies |x| < 65504.  A call
 * that hands a larger value to that pipe stores 1 to *range_flag (never cleared by the library; plain store, no
 * synchronisation added): the results of that call are then NOT trustworthy -- re-run it with
 * GNNPP_PREC_FP32.  gnnpp_error_string(GNNPP_ERR_RANGE) is the message the Python layer raises. */

/*
 * Training variant of gnnpp_lsigf_fwd (loss.backward() at agents/decentralplannerlocal.py:314):
 *   zs            optional out [E*K, B*N, G] node-major: every tap signal z_{e,k} = x S_e^k, so that
 *                 dW[f,e,k,g] = sum_{b,n} dy[b,f,n] z_{e,k}[b,n,g] is one library GEMM per tap;
 *   s_transposed  use S^T.  The input gradient of the filter is itself a filter,
 *                 dx = sum_k W_k^T . dy . (S^T)^k, i.e. this call with x := dy, taps packed from
 *                 h.permute(3,1,2,0) and s_transposed = 1.  This form always contracts on the exact
 *                 fp32 MFMA, whatever `precision` says (cotangents are far below the f16 normal range).
 * LIMIT: this call keeps a graph's rows in one workgroup's LDS: N <= GNNPP_MAX_ROWS (112; 100 guaranteed at
 * G = F = 128), else GNNPP_ERR_UNSUPPORTED.  Larger graphs train through the dense form described at GNNPP_MAX_ROWS,
 * forward AND backward as gnnpp_gemm_kmajor calls (shifts, tap contraction, dh = dy^T Z, dZ = dy h, the adjoint shift
 * chain dz_{k-1} += S dz_k): graphML._LSIGFFunction does exactly that (the reference trains on 10-agent teams,
 * configs/dcp_*.json, but BatchLSIGF itself has no size limit).
 */
int gnnpp_lsigf_fwd_save(const float* x, const void* S, const float* packed, const float* bias,
                         float* y, float* zs, int B, int N, int Nin, int G, int F, int K, int E,
                         int s_is_f64, int s_batched, int s_transposed, int x_node_major,
                         int y_node_major, int relu, int bias_per_node, int precision, int* range_flag,
                         void* stream);

/* 1 when a graph of N nodes with G input / F output features, K taps and E edge features fits the LDS-resident
 * filter kernels (gnnpp_lsigf_fwd / _save return GNNPP_OK for it), 0 when they answer GNNPP_ERR_UNSUPPORTED for
 * LACK OF LDS (N > GNNPP_MAX_ROWS, or fewer nodes with wide features), negative on invalid arguments.  The
 * reference's BatchLSIGF (utils/graphUtils/graphML.py:2273-2367) has no size limit: a caller uses this to tell
 * "take the dense-GEMM form" (what graphML._lsigf_large does) from any other GNNPP_ERR_UNSUPPORTED, which must
 * stay an error (v310). */
int gnnpp_lsigf_fits(int N, int G, int F, int K, int E);

/* ------------------------------------------------------------------------------------------
 * Per-agent encoder: 5 x (conv3x3 pad 1 -> BatchNorm(eval) -> ReLU [-> MaxPool 2]) -> flatten ->
 * Linear(128,128) -> ReLU.  Replaces ConvLayers + compressMLP as run by
 * DecentralPlannerNet.forward (graphs/models/decentralplanner.py:284-290; layers :155-195).
 * ------------------------------------------------------------------------------------------ */

/* Raw parameters in the reference's state_dict layout. */
typedef struct gnnpp_encoder_params {
    const float* conv_w[5];     /* ConvLayers.{0,4,7,11,14}.weight  [Cout,Cin,3,3]             */
    const float* conv_b[5];     /* ConvLayers.{0,4,7,11,14}.bias    [Cout]                     */
    const float* bn_w[5];       /* ConvLayers.{1,5,8,12,15}.weight  [Cout]                     */
    const float* bn_b[5];       /* ...bias                                                      */
    const float* bn_mean[5];    /* ...running_mean                                              */
    const float* bn_var[5];     /* ...running_var                                               */
    const float* fc_w;          /* compressMLP.0.weight [128,128]                               */
    const float* fc_b;          /* compressMLP.0.bias   [128]                                   */
    float        bn_eps;        /* 1e-5 (torch default, decentralplanner.py:163)                */
} gnnpp_encoder_params;

size_t gnnpp_encoder_packed_floats(void);

/* Fold eval-mode BatchNorm into a per-channel scale/shift and re-order all weights into MFMA
 * fragment order.  `params` is a HOST struct of DEVICE pointers. */
int gnnpp_encoder_pack(const gnnpp_encoder_params* params, float* packed, void* stream);

/* obs [M,3,11,11] (M = B*N agents, agent index b*N+n as in inputTensor[B,N,3,11,11]) ->
 * feat [M,128] node-major.  Any M >= 1. */
int gnnpp_encoder_fwd(const float* obs, const float* packed, float* feat, int M, int precision,
                      int* range_flag, void* stream);

/* ------------------------------------------------------------------------------------------
 * TRAIN-mode encoder, forward and backward (BASELINE config 4: loss.backward() at
 * agents/decentralplannerlocal.py:314 over the per-agent ConvLayers calls of
 * graphs/models/decentralplanner.py:284-287).  The reference runs the CNN once per agent, so BatchNorm
 * uses the statistics of (agent n, channel c) over that call's B samples, and updates its running
 * statistics N times per forward in agent order (momentum, unbiased variance): exactly that.
 *   obs        [B,N,3,11,11] (inputTensor); feat out [N,B,128] = flattened ConvLayers output per agent
 *              call (the input of compressMLP); dfeat [N,B,128] its gradient;
 *   workspace  gnnpp_encoder_train_workspace_floats(N, B) floats: holds the activations between
 *              forward and backward -- pass the SAME buffer, untouched, to the backward call;
 *   update_running != 0: p->bn_mean / p->bn_var are UPDATED in place (the one documented exception
 *              to "inputs are never written"; eval-mode packs must be rebuilt afterwards);
 *   feat_sample_major != 0: feat and dfeat are [B,N,128] instead (node-major rows: what gnnpp_lsigf_fwd takes
 *              with x_node_major, so that no transposing copy sits between encoder and graph filter);
 *   bn_num_batches  NULL, or the five BatchNorm2d.num_batches_tracked counters (int64, device): each is
 *              advanced by N, as the N forward calls of the reference do (with update_running only);
 *   g          where the gradients of the 20 parameter tensors go (overwritten, not accumulated).
 *   train_pack  NULL (the ten MFMA-fragment packs of the convolution weights are built inside the forward call, in
 *              the workspace), or the buffer gnnpp_train_pack filled from the CURRENT weights (v330: one pack launch
 *              per weight version -- shared with the graph filter's taps -- instead of one per forward call); the
 *              backward call must be given what the forward call was given.
 * fp32, deterministic (fixed-order reductions, no atomics).  compressMLP, the graph filter and the
 * action head are separate calls (gnnpp_linear_fwd / gnnpp_lsigf_fwd_save / gnnpp_lsigf_input_grad).
 * ------------------------------------------------------------------------------------------ */
typedef struct gnnpp_encoder_grads {
    float* conv_w[5];           /* d ConvLayers.{0,4,7,11,14}.weight  [Cout,Cin,3,3]             */
    float* conv_b[5];           /* d ...bias [Cout] (exactly zero in exact arithmetic: BatchNorm follows) */
    float* bn_w[5];             /* d ConvLayers.{1,5,8,12,15}.weight [Cout]                      */
    float* bn_b[5];             /* d ...bias                                                      */
} gnnpp_encoder_grads;

size_t gnnpp_encoder_train_workspace_floats(int N, int B);
int gnnpp_encoder_train_fwd(const gnnpp_encoder_params* p, const float* obs, float* workspace, float* feat,
                            int B, int N, float momentum, int update_running,
                            long long* const* bn_num_batches, int feat_sample_major, const float* train_pack,
                            void* stream);
int gnnpp_encoder_train_bwd(const gnnpp_encoder_params* p, const float* obs, float* workspace,
                            const float* dfeat, const gnnpp_encoder_grads* g, int B, int N,
                            int feat_sample_major, const float* train_pack, void* stream);

/* Every weight re-ordering the training step of config 4 needs, in ONE launch (v330; r05: five launches and a
 * transposing copy per step): the ten fragment packs of the convolution weights p->conv_w (-> train_pack,
 * gnnpp_train_pack_floats() floats, 16-byte aligned) and, for the graph filter's taps h [F,E,K,G]
 * (GraphFilterBatch.weight, graphML.py:2434), the fp32 fragments of the forward filter (-> taps_fwd, a buffer of
 * gnnpp_filter_packed_floats(G, F, K, E) floats) and of the input-gradient filter h^T [G,E,K,F] (-> taps_t,
 * gnnpp_filter_packed_floats(F, G, K, E) floats), read straight from h.  ONLY the fp32 region of the two tap buffers is
 * written: they serve gnnpp_lsigf_fwd_save with GNNPP_PREC_FP32 / _FP32_MFMA and gnnpp_lsigf_input_grad (the exact-fp32
 * contractions the training step runs), not the split-f16 or bf16x3 schedules.  p == NULL: taps only; h == NULL:
 * convolution weights only. */
size_t gnnpp_train_pack_floats(void);
int gnnpp_train_pack(const gnnpp_encoder_params* p, float* train_pack, const float* h, float* taps_fwd,
                     float* taps_t, int G, int F, int K, int E, void* stream);

/* Input gradient of the graph filter (graphML.py:2345-2366 run backwards): dx = sum_k (dy W_k) (S^T)^k, i.e. the
 * filter of the transposed taps h^T [G,E,K,F] on S^T applied to dy [.., F] -> dx [.., G]; packed_t = the taps_t of
 * gnnpp_train_pack (or gnnpp_filter_pack of h.permute(3,1,2,0)).  node_major != 0: dy [B,N,F], dx [B,N,G], and an
 * optional mask [B,N,G]: dx is stored as 0 where mask <= 0 -- the ReLU backward of the layer that produced the
 * filter's input (compressMLP's ReLU, decentralplanner.py:190), folded into this launch (v330).  Exact fp32. */
int gnnpp_lsigf_input_grad(const float* dy, const void* S, const float* packed_t, const float* mask, float* dx,
                           int B, int N, int G, int F, int K, int E, int s_is_f64, int s_batched, int node_major,
                           void* stream);

/* y = x W^T + bias (+ ReLU): forward of compressMLP (Linear 128 -> 128 + ReLU, decentralplanner.py:187-195, :289-290)
 * and of actionsMLP (Linear 128 -> 5, :232-243, :304-315) in the training step.  x [R,I], W [O,I] (nn.Linear's layout),
 * bias [O] or NULL, y [R,O].  Exact fp32 MFMA, fixed summation order.  I must be a multiple of 64 and x, W 16-byte
 * aligned (GNNPP_ERR_UNSUPPORTED otherwise).  (v330; r05 ran these as library GEMMs + an aten ReLU.) */
int gnnpp_linear_fwd(const float* x, const float* W, const float* bias, float* y, int R, int I, int O, int relu,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * The rest of the optimisation step of config 4 (agents/decentralplannerlocal.py:287-317): the weight
 * gradients that are tall-contraction GEMMs, the loss, and the Adam update.  fp32, deterministic.
 * ------------------------------------------------------------------------------------------ */
/* C[b][m][n] = sum_k A_b(m,k) * B_b(k,n), b < batch, with element addresses
 *   A_b(m,k) = A + b*a_sb + m*a_sm + k*a_sk,   B_b(k,n) = B + b*b_sb + k*b_sk + n,   C_b(m,n) = C + b*c_sb + m*c_sm + n
 * (strides in floats).  The contraction is split over workgroups (fp32 MFMA) and the partial results are
 * summed in order: meant for small M x N (<= a few hundred) and long K, where a library GEMM runs on one
 * macro tile.  Serves the graph filter's tap gradient dh (graphML.py:2345-2352 backwards: A = dy [F,(b n)],
 * B = the saved shifted signals z_k [(b n),G], batch = E*K) and Linear weight gradients (A = dY^T, B = X).
 * workspace: gnnpp_gemm_workspace_floats(batch, M, N, K) floats (may be 0 -> NULL allowed). */
size_t gnnpp_gemm_workspace_floats(int batch, int M, int N, int K);
int gnnpp_gemm_kmajor(const float* A, long long a_sb, long long a_sm, long long a_sk, const float* B,
                      long long b_sb, long long b_sk, float* C, long long c_sb, long long c_sm, int batch,
                      int M, int N, int K, float* workspace, void* stream);
/* Up to 8 independent products of that kind in ONE launch (plus one for the ordered sums): the three
 * products of a Linear layer's backward pass (dx = dY W, dW = dY^T X, db = 1^T dY) cost two launches
 * instead of six.  workspace: gnnpp_gemm_multi_workspace_floats(d, count) floats. */
typedef struct gnnpp_gemm_desc {
    const float* A; long long a_sb, a_sm, a_sk;
    const float* B; long long b_sb, b_sk;
    float*       C; long long c_sb, c_sm;
    int batch, M, N, K;
    const float* mask;          /* NULL, or addressed like C: C(m,n) is stored as 0 where mask(m,n) <= 0 -- the ReLU
                                   backward of the layer whose output gradient this product is (v330) */
} gnnpp_gemm_desc;
size_t gnnpp_gemm_multi_workspace_floats(const gnnpp_gemm_desc* d, int count);
int gnnpp_gemm_kmajor_multi(const gnnpp_gemm_desc* d, int count, float* workspace, void* stream);

/* The training loop's loss (agents/decentralplannerlocal.py:296-312), forward and backward in one launch:
 *   logits [N,B,C] (agent-major: the list forward() returns, stacked), target [B,N,C] one-hot expert actions;
 *   loss[0] = (1/N) sum_n CrossEntropyLoss(logits[n], argmax_c target[:, n])   (first maximum, like torch.max);
 *   dlogits [N,B,C] = d loss / d logits, or NULL.  C <= 64.
 *   logits_sample_major != 0: logits and dlogits are [B,N,C] (the layout the train-mode forward computes). */
int gnnpp_policy_loss(const float* logits, const float* target, float* loss, float* dlogits, int B, int N,
                      int C, int logits_sample_major, void* stream);

/* torch.optim.Adam's update (amsgrad = False; L2 weight decay added to the gradient; bias correction) of up
 * to 32 tensors in one launch:  p, m (exp_avg), v (exp_avg_sq) are updated in place from g.
 *   state  8 floats on the device, zero before the first step: state[0] = number of steps taken so far, [1], [2] the
 *          bias-correction factors of the last tick, [3] scratch (a workgroup arrival counter, zero between calls),
 *          [4] .. [7] the betas and the factors of the NEXT step (precomputed by the tick's last workgroup);
 *   tick   != 0: this call is the first table of a step: it uses t = state[0] + 1 and its last workgroup stores t
 *          (v330: inside the same launch; pass 0 for further tables of the same step).  The counter lives on the
 *          device so that the call can be captured in a HIP graph. */
typedef struct gnnpp_adam_tensors {
    float* p[32];
    const float* g[32];
    float* m[32];
    float* v[32];
    long long numel[32];
    int count;
} gnnpp_adam_tensors;
int gnnpp_adam_step(const gnnpp_adam_tensors* t, float* state, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int tick, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole policy step: DecentralPlannerNet.addGSO + forward (decentralplanner.py:266-318):
 * encoder -> GraphFilterBatch(128,128,K,E=1) -> ReLU -> actionsMLP Linear(128,5).
 * ------------------------------------------------------------------------------------------ */
/*
 * obs      [B,N,3,11,11];  S [B,E,N,N] float or double (s_is_f64); E = 1 in the reference's
 *          configuration (decentralplanner.py:208; addGSO :266-276 handles E > 1);
 * enc_packed / filt_packed from the two pack calls (taps h[128,E,K,128]); gf_bias [128] (GFL.0.bias),
 * act_w [5,128], act_b [5] (actionsMLP.0.*);
 * feat_ws  workspace [B*N,128] (receives the encoder output, i.e. extractFeatureMap in
 *          node-major order);
 * logits   [N,B,5]: logits + n*B*5 is the contiguous [B,5] tensor of agent n, the n-th element
 *          of the list the reference returns (decentralplanner.py:303-318).
 */
int gnnpp_policy_fwd(const float* obs, const void* S, const float* enc_packed,
                     const float* filt_packed, const float* gf_bias, const float* act_w,
                     const float* act_b, float* feat_ws, float* logits, int B, int N, int K, int E,
                     int s_is_f64, int precision, int* range_flag, void* stream);

/* Last graph-filter layer + ReLU + action head of a planner with SEVERAL graph-filter layers
 * (decentralplanner.py:205-224 builds L layers, :293-315 runs them and the head): the earlier layers
 * are gnnpp_lsigf_fwd calls (node-major in/out, relu = 1); this call runs layer L on
 * x [B,N,G] node-major with taps h[F,E,K,G] (F <= 128), bias [F] or NULL, then Linear(F,5):
 * logits [N,B,5] as gnnpp_policy_fwd. */
int gnnpp_filter_head_fwd(const float* x, const void* S, const float* packed, const float* bias,
                          const float* act_w, const float* act_b, float* logits, int B, int N, int G,
                          int F, int K, int E, int s_is_f64, int precision, int* range_flag, void* stream);

/* Which schedule gnnpp_filter_head_fwd / the second launch of gnnpp_policy_fwd would run for this shape under the
 * current tuning: 0 split-f16 planes (v_mfma_f32_16x16x32_f16), 1 exact fp32 MFMA (v_mfma_f32_16x16x4_f32), 2 bf16x3 planes
 * (v_mfma_f32_16x16x32_bf16) in an LDS buffer of their own, 3 bf16x3 planes aliased onto the dead z buffer;
 * -1: the shape runs on the general filter kernels (lsigf_kernel / the small-graph kernels), GNNPP_ERR_ARG < -1 never.
 * The answer depends on B through the workgroups-per-graph heuristic (GNNPP_TUNE_FILTER_SPLIT): a measurement states
 * the instruction it priced from this call instead of guessing it.  (v330) */
int gnnpp_filter_head_mode(int B, int N, int K, int precision);

/* Action decode used by the rollout loop (utils/multirobotsim_dcenlocal.py:589-591: LogSoftmax
 * then argmax == argmax of the logits, first maximum wins like torch.max).
 * logits [N,B,5] -> actions [B,N] int32. */
int gnnpp_decode_actions(const float* logits, int* actions, int B, int N, void* stream);

/* ------------------------------------------------------------------------------------------
 * Batched rollout step around the forward (B independent episodes resident on the device):
 *   gnnpp_rollout_observe  AgentState.toInputTensor             dataloader/statetransformer.py:82-130
 *   gnnpp_rollout_gso      multiRobotSim.computeAdjacencyMatrix utils/multirobotsim_dcenlocal.py:320-365
 *                          (+ getGSO :367-394: radius bookkeeping)
 *   gnnpp_rollout_move     multiRobotSim.move :562-723 with interRobotCollision :462-555
 * One struct (HOST struct of DEVICE pointers) carries the episode state; each call reads the
 * fields of its section.  Integer / boolean / fp64 work: bit-exact against the simulator.
 * ------------------------------------------------------------------------------------------ */
#define GNNPP_ROLLOUT_MAX_AGENTS 128
#define GNNPP_TIE_LOWEST  0   /* colliding agent with the lowest index keeps its move          */
#define GNNPP_TIE_HASHED  1   /* counter-based hash of (seed, episode, step, call)             */
#define GNNPP_TIE_REPLAY  2   /* replay recorded random.choice outcomes (parity tests)         */
#define GNNPP_TIE_MT19937 3   /* CPython's random.choice itself on a per-episode Mersenne-Twister stream
                                 (rng_words): what the reference does at :489 after random.seed(s)  */

typedef struct gnnpp_rollout {
    /* episode state shared by the three calls */
    const unsigned char* grid;  /* [B,H,W] when grid_batched else [H,W]; 1 = obstacle           */
    int          grid_batched;
    const int*   goal;          /* [B,N,2] (row, col)                                           */
    int*         pos;           /* [B,N,2] current positions; updated by move                   */
    int          B, N, H, W;
    /* observe */
    float*       obs;           /* out [B,N,3,11,11]: obstacle FOV, goal / projected goal, agents */
    /* gso */
    double*      radius;        /* [B] communication radius, in/out (commR at step 0)           */
    float*       S;             /* out [B,N,N] = float(D^-1/2 A D^-1/2)                          */
    int*         connected;     /* out [B] or NULL                                              */
    int          grow;          /* 1 at step 0: radius /= 1.1, then *= 1.1 until connected      */
    /* move */
    const float* logits;        /* [N,B,5] from gnnpp_policy_fwd (argmax decoded here), or NULL */
    const int*   actions;       /* [B,N] action ids when logits == NULL                         */
    int*         reached;       /* [B,N] 0/1                 (count_reachgoal)                   */
    int*         start_step;    /* [B,N], -1 = None          (startStep_action_predict)          */
    int*         end_step;      /* [B,N], -1 = None          (endStep_action_predict)            */
    const int*   maxstep;       /* [B] per-episode step limit (rate_maxstep * makespanTarget)    */
    int*         done;          /* [B] in/out, or NULL.  The reference's loop stops calling move() for
                                   a case after the call that saw allReachGoal at entry, or ran with
                                   currentstep >= maxstep (agents/decentralplannerlocal.py:560-605).
                                   That call sets done[b] = 1; an episode with done[b] != 0 -- or, with
                                   or without this array, one called with currentstep > maxstep[b] -- is
                                   FROZEN: nothing of its state (pos, reached, steps, stats) changes  */
    int*         flags;         /* out [B,3]: allReachGoal at entry, moveCollision, predictCollision */
    int*         stats;         /* out [B,2]: makespan, flowtime (written when the episode ends) */
    int          currentstep;   /* 1-based step index, as the agent passes it (:588)            */
    int          tie_mode;      /* GNNPP_TIE_*: stands in for random.choice (:489)              */
    unsigned     seed;
    const short* choices;       /* [B,max_choices] recorded outcomes for GNNPP_TIE_REPLAY       */
    int*         choice_count;  /* out [B] tie-breaks consumed in this call, or NULL            */
    int          max_choices;
    int*         range_flag;    /* gnnpp_rollout_policy_step only: range guard of the policy, or NULL */
    const unsigned* rng_words;  /* GNNPP_TIE_MT19937: [B,rng_max] successive genrand_uint32() outputs of
                                   each episode's generator (random.Random(seed).getrandbits(32))   */
    int*         rng_cursor;    /* [B] in/out: words consumed so far                              */
    int          rng_max;
} gnnpp_rollout;

int gnnpp_rollout_observe(const gnnpp_rollout* r, void* stream);
int gnnpp_rollout_gso(const gnnpp_rollout* r, void* stream);
int gnnpp_rollout_move(const gnnpp_rollout* r, void* stream);
/* gnnpp_rollout_gso (grow = 0) and gnnpp_rollout_observe of the current positions in ONE launch: the two depend only
 * on the positions, so their workgroups run side by side (large teams, where one workgroup per episode is too
 * little for gnnpp_rollout_step).  Same results as the two calls. */
int gnnpp_rollout_gso_observe(const gnnpp_rollout* r, void* stream);
/* move -> gso (grow = 0) -> observe of the new positions in ONE launch: the simulator work between
 * two policy forwards of a rollout (the loop agents/decentralplannerlocal.py:560-599 runs move,
 * then getCurrentState + getGSO of the next iteration).  Same results as the three calls in
 * sequence; fields as for those calls. */
int gnnpp_rollout_step(const gnnpp_rollout* r, void* stream);
/* A WHOLE rollout step in one launch for teams of N <= 16 agents (K = 2, 3 or 4): policy forward on r->obs /
 * r->S (fp32) with the packed encoder / filter weights, logits to r->logits [N,B,5], then move ->
 * gso -> observe as gnnpp_rollout_step; r->obs and r->S are overwritten with the next step's.
 * GNNPP_ERR_UNSUPPORTED (nothing enqueued) when the shape does not qualify (see GNNPP_TUNE_FUSED_POLICY;
 * precision GNNPP_PREC_FP32 or GNNPP_PREC_SPLIT_F16; H*W must fit the kernel's spare LDS: 10 208 cells under
 * GNNPP_PREC_FP32): use gnnpp_policy_fwd + gnnpp_rollout_step then. */
int gnnpp_rollout_policy_step(const gnnpp_rollout* r, const float* enc_packed, const float* filt_packed,
                              const float* gf_bias, const float* act_w, const float* act_b, int K,
                              int precision, void* stream);
/* nsteps consecutive calls of gnnpp_rollout_policy_step with currentstep = r->currentstep, +1, ...: the
 * inner loop of a rollout (agents/decentralplannerlocal.py:560-599) enqueued back to back, so the host
 * returns to its interpreter once per nsteps launches.  Episodes that end on the way freeze (see `done`);
 * GNNPP_TIE_REPLAY is per-call state and is refused (GNNPP_ERR_ARG). */
int gnnpp_rollout_policy_steps(const gnnpp_rollout* r, const float* enc_packed, const float* filt_packed,
                               const float* gf_bias, const float* act_w, const float* act_b, int K,
                               int nsteps, int precision, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GNNPP_H_ */

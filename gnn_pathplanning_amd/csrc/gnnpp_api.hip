// extern "C" surface of libgnnpp.so -- see include/gnnpp.h for the contract of every entry point.
// Single translation unit: the kernels are included so hipcc builds one code object.
#include "../../include/gnnpp.h"
#ifdef GNNPP_MEASURE
#include "gnnpp_measure.h"
#endif

#include "encoder_pack.hip"
#include "encoder_ring_f32.hip"
#include "encoder_kernel_f32.hip"
#include "rollout_kernels.hip"       // before the fused policy kernel, which can run the simulator step too
#include "encoder_kernel_h2.hip"
#include "encoder_kernel_b3.hip"
#include "lsigf_kernel.hip"
#include "policy_filter_kernel.hip"
#include "lsigf_small_kernel.hip"
#include "train_encoder.hip"
#include "train_ops.hip"

using namespace gnnpp;

static_assert(GNNPP_OK == 0 && GNNPP_ERR_UNSUPPORTED == -2 && GNNPP_ERR_LAUNCH == -3, "codes");

extern "C" {

int gnnpp_version(void) { return 330; }

const char* gnnpp_error_string(int code) {
    switch (code) {
        case GNNPP_OK: return "ok";
        case GNNPP_ERR_ARG: return "invalid argument (null pointer, non-positive size or inconsistent flags)";
        case GNNPP_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels (more than 112 nodes, or the graph's rows exceed the 160 KB LDS budget: N <= 100 is guaranteed at G, F <= 128)";
        case GNNPP_ERR_RANGE: return "an activation left the f16 range (|x| >= 65504) of GNNPP_PREC_SPLIT_F16; results of this call are not trustworthy -- run it with GNNPP_PREC_FP32 (the default), which has no input domain";
        case GNNPP_ERR_LAUNCH: return "HIP kernel launch failed";
        default: return "unknown gnnpp error code";
    }
}

size_t gnnpp_filter_packed_floats(int G, int F, int K, int E) {
    if (G <= 0 || F <= 0 || K <= 0 || E <= 0) return 0;
    return filter_packed_floats(G, F, K, E);
}

int gnnpp_filter_pack(const float* h, float* packed, int G, int F, int K, int E, void* stream) {
    if (!h || !packed || G <= 0 || F <= 0 || K <= 0 || E <= 0) return GNNPP_ERR_ARG;
    return filter_pack_launch(h, packed, G, F, K, E, static_cast<hipStream_t>(stream));
}

// One launch covers F <= 128 output features; wider filters run as several launches over
// output-feature chunks inside lsigf_launch (each chunk recomputes the cheap shifts).
int gnnpp_lsigf_fwd(const float* x, const void* S, const float* packed, const float* bias,
                    float* y, int B, int N, int Nin, int G, int F, int K, int E, int s_is_f64,
                    int s_batched, int x_node_major, int y_node_major, int relu, int bias_per_node,
                    int precision, int* range_flag, void* stream) {
    if (!x || !packed || !y || B <= 0 || N <= 0 || Nin <= 0 || Nin > N || G <= 0 || F <= 0 ||
        K <= 0 || E <= 0 || precision < 0 || precision > 2)
        return GNNPP_ERR_ARG;
    if (K > 1 && !S) return GNNPP_ERR_ARG;
    if ((x_node_major || y_node_major) && Nin != N) return GNNPP_ERR_ARG;
    if (N > GNNPP_MAX_ROWS) return GNNPP_ERR_UNSUPPORTED;
    LsigfArgs a = {};
    a.x = x; a.S = S; a.wpk = packed; a.bias = bias; a.y = y;
    a.B = B; a.N = N; a.Nin = Nin; a.G = G; a.F = F; a.K = K; a.E = E;
    a.s_is_f64 = s_is_f64; a.s_batched = s_batched;
    a.x_node_major = x_node_major; a.y_node_major = y_node_major; a.relu = relu;
    a.bias_per_node = bias && bias_per_node; a.range_flag = range_flag; a.prec = precision;
    return lsigf_launch(a, static_cast<hipStream_t>(stream));
}

int gnnpp_lsigf_fwd_save(const float* x, const void* S, const float* packed, const float* bias,
                         float* y, float* zs, int B, int N, int Nin, int G, int F, int K, int E,
                         int s_is_f64, int s_batched, int s_transposed, int x_node_major,
                         int y_node_major, int relu, int bias_per_node, int precision, int* range_flag,
                         void* stream) {
    if (!x || !packed || !y || B <= 0 || N <= 0 || Nin <= 0 || Nin > N || G <= 0 || F <= 0 ||
        K <= 0 || E <= 0 || precision < 0 || precision > 2)
        return GNNPP_ERR_ARG;
    if (K > 1 && !S) return GNNPP_ERR_ARG;
    if ((x_node_major || y_node_major) && Nin != N) return GNNPP_ERR_ARG;
    if (N > GNNPP_MAX_ROWS) return GNNPP_ERR_UNSUPPORTED;
    LsigfArgs a = {};
    a.x = x; a.S = S; a.wpk = packed; a.bias = bias; a.y = y; a.zs = zs;
    a.B = B; a.N = N; a.Nin = Nin; a.G = G; a.F = F; a.K = K; a.E = E;
    a.s_is_f64 = s_is_f64; a.s_batched = s_batched; a.s_transposed = s_transposed;
    a.x_node_major = x_node_major; a.y_node_major = y_node_major; a.relu = relu;
    a.bias_per_node = bias && bias_per_node; a.range_flag = range_flag; a.prec = precision;
    return lsigf_launch(a, static_cast<hipStream_t>(stream));
}

int gnnpp_lsigf_input_grad(const float* dy, const void* S, const float* packed_t, const float* mask, float* dx,
                           int B, int N, int G, int F, int K, int E, int s_is_f64, int s_batched, int node_major,
                           void* stream) {
    if (!dy || !packed_t || !dx || B <= 0 || N <= 0 || G <= 0 || F <= 0 || K <= 0 || E <= 0) return GNNPP_ERR_ARG;
    if (K > 1 && !S) return GNNPP_ERR_ARG;
    if (mask && !node_major) return GNNPP_ERR_ARG;     // (the mask is applied where the node-major rows are stored)
    if (N > GNNPP_MAX_ROWS) return GNNPP_ERR_UNSUPPORTED;
    LsigfArgs a = {};
    // the filter of h^T [G,E,K,F] on S^T: F input features (dy), G output features (dx); exact fp32 contraction
    a.x = dy; a.S = S; a.wpk = packed_t; a.bias = nullptr; a.y = dx; a.y_mask = mask;
    a.B = B; a.N = N; a.Nin = N; a.G = F; a.F = G; a.K = K; a.E = E;
    a.s_is_f64 = s_is_f64; a.s_batched = s_batched; a.s_transposed = 1;
    a.x_node_major = node_major; a.y_node_major = node_major; a.relu = 0;
    a.prec = GNNPP_PREC_FP32_MFMA;
    return lsigf_launch(a, static_cast<hipStream_t>(stream));
}

int gnnpp_lsigf_fits(int N, int G, int F, int K, int E) {
    if (N <= 0 || G <= 0 || F <= 0 || K <= 0 || E <= 0) return GNNPP_ERR_ARG;
    if (N > GNNPP_MAX_ROWS) return 0;
    if ((F + 127) / 128 > 64) return 0;                // lsigf_launch plans at most 64 chunks of 128 output features:
                                                       // F > 8192 takes the dense form too (the reference
                                                       // BatchLSIGF has no such limit)
    for (int f0 = 0; f0 < F; f0 += 128) {              // (lsigf_launch's chunks of <= 128 output features)
        LsigfArgs a = {};
        a.B = 1; a.N = N; a.Nin = N; a.G = G; a.K = K; a.E = E;
        a.F_all = F; a.f0 = f0; a.F = F - f0 < 128 ? F - f0 : 128;
        LsigfPlan plan;
        if (lsigf_plan(a, plan) != 0) return 0;
    }
    return 1;
}

size_t gnnpp_encoder_packed_floats(void) { return EncLayout::kTotal; }

int gnnpp_encoder_pack(const gnnpp_encoder_params* p, float* packed, void* stream) {
    if (!p || !packed || !p->fc_w || !p->fc_b) return GNNPP_ERR_ARG;
    EncRawParams rp;
    for (int i = 0; i < 5; ++i) {
        if (!p->conv_w[i] || !p->conv_b[i] || !p->bn_w[i] || !p->bn_b[i] || !p->bn_mean[i] ||
            !p->bn_var[i])
            return GNNPP_ERR_ARG;
        rp.conv_w[i] = p->conv_w[i]; rp.conv_b[i] = p->conv_b[i];
        rp.bn_w[i] = p->bn_w[i]; rp.bn_b[i] = p->bn_b[i];
        rp.bn_mean[i] = p->bn_mean[i]; rp.bn_var[i] = p->bn_var[i];
    }
    rp.fc_w = p->fc_w; rp.fc_b = p->fc_b; rp.bn_eps = p->bn_eps;
    return encoder_pack_launch(rp, packed, static_cast<hipStream_t>(stream));
}

int gnnpp_encoder_fwd(const float* obs, const float* packed, float* feat, int M, int precision,
                      int* range_flag, void* stream) {
    if (!obs || !packed || !feat || M <= 0 || precision < 0 || precision > 2) return GNNPP_ERR_ARG;
    return encoder_launch(obs, packed, feat, M, range_flag, precision, static_cast<hipStream_t>(stream));
}

std::atomic<int> g_fused_policy{1};   // GNNPP_TUNE_FUSED_POLICY

// Does the one-launch policy kernel apply?  (bf16x3 or split-f16 arithmetic -- the exact-fp32 MFMA schedule has no
// fused form --, N <= 16, K = 2..4, and a batch for which one workgroup per graph pays: measured -8 % at B = 512,
// -17 % at B <= 64 (N = 10), but +40 % at B = 2048 -- or N nearly fills the 16-lane tile.)
static bool fused_policy_applies(int B, int N, int K, int prec) {
#ifdef GNNPP_MEASURE
    if (g_filter_ablate.load(std::memory_order_relaxed) || g_encoder_stop.load(std::memory_order_relaxed))
        return false;
#endif
    const int knob = g_fused_policy.load(std::memory_order_relaxed);            // 2: whatever the batch size
    const bool fused_pays = B <= 2 * 256 || N >= 13 || knob == 2;
    return knob && fused_pays && prec != kPrecFp32Mfma &&
           N <= kTileAgents && K >= kPolicyTapsMin && K <= kPolicyTapsMax;
}

int gnnpp_policy_fwd(const float* obs, const void* S, const float* enc_packed,
                     const float* filt_packed, const float* gf_bias, const float* act_w,
                     const float* act_b, float* feat_ws, float* logits, int B, int N, int K, int E,
                     int s_is_f64, int precision, int* range_flag, void* stream) {
    if (!obs || !enc_packed || !filt_packed || !act_w || !act_b || !feat_ws || !logits || B <= 0 ||
        N <= 0 || K <= 0 || E <= 0 || precision < 0 || precision > 2)
        return GNNPP_ERR_ARG;
    if (K > 1 && !S) return GNNPP_ERR_ARG;
    if (N > GNNPP_MAX_ROWS) return GNNPP_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    LsigfArgs a = {};
    a.x = feat_ws; a.S = S; a.wpk = filt_packed; a.bias = gf_bias; a.y = nullptr;
    a.act_w = act_w; a.act_b = act_b; a.logits = logits;
    a.B = B; a.N = N; a.Nin = N; a.G = GNNPP_FEAT; a.F = GNNPP_FEAT; a.K = K; a.E = E;
    a.s_is_f64 = s_is_f64; a.s_batched = 1; a.x_node_major = 1; a.y_node_major = 1; a.relu = 1;
    a.range_flag = range_flag; a.prec = precision;
    LsigfPlan plan;
    int rc = lsigf_plan(a, plan);
    if (rc) return rc;
    // Fused path: a 16-lane tile per graph wastes (16 - N) / 16 of the encoder's lanes, which is free
    // while the graphs fit the chip in one round (2 workgroups per CU).
    if (E == 1 && fused_policy_applies(B, N, K, precision)) {
        // one launch: a workgroup encodes one graph's agents and runs its filter + action head
        PolicyTail pt;
        pt.S = S; pt.filt_h2 = precision == kPrecFp32 ? a.wpk_b : a.wpk_h;
        pt.gf_bias = gf_bias; pt.act_w = act_w; pt.act_b = act_b;
        pt.logits = logits; pt.B = B; pt.N = N; pt.s_is_f64 = s_is_f64;
        pt.range_flag = range_flag;
        pt.with_sim = 0;
        return precision == kPrecFp32 ? policy_launch_fused_b3(obs, enc_packed, pt, K, st)
                                      : policy_launch_fused(obs, enc_packed, pt, K, st);
    }
    rc = encoder_launch(obs, enc_packed, feat_ws, B * N, range_flag, precision, st);
    return rc ? rc : lsigf_dispatch(a, plan, st);
}

int gnnpp_filter_head_fwd(const float* x, const void* S, const float* packed, const float* bias,
                          const float* act_w, const float* act_b, float* logits, int B, int N, int G,
                          int F, int K, int E, int s_is_f64, int precision, int* range_flag, void* stream) {
    if (!x || !packed || !act_w || !act_b || !logits || B <= 0 || N <= 0 || G <= 0 || F <= 0 || K <= 0 ||
        E <= 0 || precision < 0 || precision > 2)
        return GNNPP_ERR_ARG;
    if (K > 1 && !S) return GNNPP_ERR_ARG;
    if (N > GNNPP_MAX_ROWS || F > 128) return GNNPP_ERR_UNSUPPORTED;   // the head needs all features at once
    LsigfArgs a = {};
    a.x = x; a.S = S; a.wpk = packed; a.bias = bias; a.y = nullptr;
    a.act_w = act_w; a.act_b = act_b; a.logits = logits;
    a.B = B; a.N = N; a.Nin = N; a.G = G; a.F = F; a.K = K; a.E = E;
    a.s_is_f64 = s_is_f64; a.s_batched = 1; a.x_node_major = 1; a.y_node_major = 1; a.relu = 1;
    a.range_flag = range_flag; a.prec = precision;
    return lsigf_launch(a, static_cast<hipStream_t>(stream));
}

int gnnpp_filter_head_mode(int B, int N, int K, int precision) {
    if (B <= 0 || N <= 0 || K <= 0 || precision < 0 || precision > 2) return -1;
    if (N > GNNPP_MAX_ROWS) return -1;
    alignas(16) static const float dummy[4] = {0.f, 0.f, 0.f, 0.f};   // (never dereferenced: the plan only tests pointers)
    LsigfArgs a = {};
    a.x = dummy; a.S = dummy; a.wpk = dummy; a.act_w = dummy; a.act_b = dummy;
    a.logits = const_cast<float*>(dummy);
    a.B = B; a.N = N; a.Nin = N; a.G = 128; a.F = 128; a.K = K; a.E = 1;
    a.s_batched = 1; a.x_node_major = 1; a.y_node_major = 1; a.relu = 1; a.prec = precision;
    a.F_all = 128;
    LsigfPlan plan;
    if (lsigf_plan(a, plan) != 0) return -1;
    PfLaunch L;
    return policy_filter_plan(a, L) ? L.mode : -1;
}

size_t gnnpp_encoder_train_workspace_floats(int N, int B) {
    if (N <= 0 || B <= 0) return 0;
    return train_ws_layout(N, B).total;
}

static int train_params_ok(const gnnpp_encoder_params* p, EncRawParams& rp) {
    if (!p) return 0;
    for (int i = 0; i < 5; ++i) {
        if (!p->conv_w[i] || !p->conv_b[i] || !p->bn_w[i] || !p->bn_b[i]) return 0;
        rp.conv_w[i] = p->conv_w[i]; rp.conv_b[i] = p->conv_b[i];
        rp.bn_w[i] = p->bn_w[i]; rp.bn_b[i] = p->bn_b[i];
        rp.bn_mean[i] = p->bn_mean[i]; rp.bn_var[i] = p->bn_var[i];
    }
    rp.fc_w = p->fc_w; rp.fc_b = p->fc_b; rp.bn_eps = p->bn_eps;
    return 1;
}

size_t gnnpp_train_pack_floats(void) { return train_pack_layout().total; }

int gnnpp_train_pack(const gnnpp_encoder_params* p, float* train_pack, const float* h, float* taps_fwd,
                     float* taps_t, int G, int F, int K, int E, void* stream) {
    if (!p && !h) return GNNPP_ERR_ARG;
    const float* cw[5] = {};
    if (p) {
        if (!train_pack || (reinterpret_cast<size_t>(train_pack) & 15)) return GNNPP_ERR_ARG;
        for (int i = 0; i < 5; ++i) {
            if (!p->conv_w[i]) return GNNPP_ERR_ARG;
            cw[i] = p->conv_w[i];
        }
    }
    TrainFilterPack fp = {};
    if (h) {
        if (!taps_fwd || !taps_t || G <= 0 || F <= 0 || K <= 0 || E <= 0) return GNNPP_ERR_ARG;
        fp.h = h; fp.fwd = taps_fwd; fp.tr = taps_t; fp.G = G; fp.F = F; fp.K = K; fp.E = E;
    }
    return train_pack_launch(p ? cw : nullptr, p ? train_pack : nullptr, fp, static_cast<hipStream_t>(stream));
}

int gnnpp_encoder_train_fwd(const gnnpp_encoder_params* p, const float* obs, float* workspace, float* feat,
                            int B, int N, float momentum, int update_running,
                            long long* const* bn_num_batches, int feat_sample_major, const float* train_pack,
                            void* stream) {
    EncRawParams rp;
    if (train_pack && (reinterpret_cast<size_t>(train_pack) & 15)) return GNNPP_ERR_ARG;
    if (!train_params_ok(p, rp) || !obs || !workspace || !feat || B <= 0 || N <= 0) return GNNPP_ERR_ARG;
    float* rm[5];
    float* rv[5];
    for (int i = 0; i < 5; ++i) {
        if (update_running && (!p->bn_mean[i] || !p->bn_var[i])) return GNNPP_ERR_ARG;
        rm[i] = update_running ? const_cast<float*>(p->bn_mean[i]) : nullptr;
        rv[i] = update_running ? const_cast<float*>(p->bn_var[i]) : nullptr;
    }
    if (reinterpret_cast<size_t>(workspace) & 15) return GNNPP_ERR_ARG;      // (16-byte loads on its regions)
    return train_encoder_fwd(rp, rm, rv, update_running ? bn_num_batches : nullptr, momentum, obs, workspace, feat,
                             N, B, feat_sample_major, static_cast<hipStream_t>(stream), train_pack);
}

int gnnpp_encoder_train_bwd(const gnnpp_encoder_params* p, const float* obs, float* workspace,
                            const float* dfeat, const gnnpp_encoder_grads* g, int B, int N,
                            int feat_sample_major, const float* train_pack, void* stream) {
    EncRawParams rp;
    if (train_pack && (reinterpret_cast<size_t>(train_pack) & 15)) return GNNPP_ERR_ARG;
    if (!train_params_ok(p, rp) || !obs || !workspace || !dfeat || !g || B <= 0 || N <= 0) return GNNPP_ERR_ARG;
    for (int i = 0; i < 5; ++i)
        if (!g->conv_w[i] || !g->conv_b[i] || !g->bn_w[i] || !g->bn_b[i]) return GNNPP_ERR_ARG;
    return train_encoder_bwd(rp, obs, workspace, dfeat, g->conv_w, g->conv_b, g->bn_w, g->bn_b, N, B,
                             feat_sample_major, static_cast<hipStream_t>(stream), train_pack);
}

size_t gnnpp_gemm_workspace_floats(int batch, int M, int N, int K) {
    if (batch <= 0 || M <= 0 || N <= 0 || K <= 0) return 0;
    return gemm_workspace_floats(batch, M, N, K);
}

int gnnpp_gemm_kmajor(const float* A, long long a_sb, long long a_sm, long long a_sk, const float* B,
                      long long b_sb, long long b_sk, float* C, long long c_sb, long long c_sm, int batch,
                      int M, int N, int K, float* workspace, void* stream) {
    if (!A || !B || !C || batch <= 0 || M <= 0 || N <= 0 || K <= 0) return GNNPP_ERR_ARG;
    const gnnpp_gemm_desc d = {A, a_sb, a_sm, a_sk, B, b_sb, b_sk, C, c_sb, c_sm, batch, M, N, K, nullptr};
    return gnnpp_gemm_kmajor_multi(&d, 1, workspace, stream);
}

size_t gnnpp_gemm_multi_workspace_floats(const gnnpp_gemm_desc* d, int count) {
    if (!d || count <= 0 || count > kGemmMax) return 0;
    size_t n = 0;
    for (int i = 0; i < count; ++i)
        if (d[i].batch > 0 && d[i].M > 0 && d[i].N > 0 && d[i].K > 0)
            n += gemm_workspace_floats(d[i].batch, d[i].M, d[i].N, d[i].K);
    return n;
}

int gnnpp_gemm_kmajor_multi(const gnnpp_gemm_desc* d, int count, float* workspace, void* stream) {
    if (!d || count <= 0 || count > kGemmMax) return GNNPP_ERR_ARG;
    GemmTable tb = {};
    for (int i = 0; i < count; ++i) {
        if (!d[i].A || !d[i].B || !d[i].C || d[i].batch <= 0 || d[i].M <= 0 || d[i].N <= 0 || d[i].K <= 0)
            return GNNPP_ERR_ARG;
        GemmOne& g = tb.g[i];
        g.A = d[i].A; g.a_sb = (long)d[i].a_sb; g.a_sm = (long)d[i].a_sm; g.a_sk = (long)d[i].a_sk;
        g.B = d[i].B; g.b_sb = (long)d[i].b_sb; g.b_sk = (long)d[i].b_sk;
        g.C = d[i].C; g.c_sb = (long)d[i].c_sb; g.c_sm = (long)d[i].c_sm;
        g.batch = d[i].batch; g.M = d[i].M; g.N = d[i].N; g.K = d[i].K; g.mask = d[i].mask;
    }
    tb.count = count;
    if (gnnpp_gemm_multi_workspace_floats(d, count) > 0 && !workspace) return GNNPP_ERR_ARG;
    return gemm_multi_launch(tb, workspace, static_cast<hipStream_t>(stream));
}

int gnnpp_linear_fwd(const float* x, const float* W, const float* bias, float* y, int R, int I, int O, int relu,
                     void* stream) {
    if (!x || !W || !y || R <= 0 || I <= 0 || O <= 0) return GNNPP_ERR_ARG;
    if (I % 64 != 0 || ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(W)) & 15)) return GNNPP_ERR_UNSUPPORTED;
    const int tiles = ((O + 15) / 16) * ((R + 15) / 16);
    hipLaunchKernelGGL(linear_fwd_kernel, dim3((tiles + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x, W,
                       bias, y, R, I, O, relu);
    return hipGetLastError() == hipSuccess ? GNNPP_OK : GNNPP_ERR_LAUNCH;
}

int gnnpp_policy_loss(const float* logits, const float* target, float* loss, float* dlogits, int B, int N,
                      int C, int logits_sample_major, void* stream) {
    if (!logits || !target || !loss || B <= 0 || N <= 0 || C <= 0 || C > 64) return GNNPP_ERR_ARG;
    hipLaunchKernelGGL(policy_loss_kernel, dim3(1), dim3(1024), 1024 * sizeof(double),
                       static_cast<hipStream_t>(stream), logits, target, loss, dlogits, B, N, C, logits_sample_major);
    return hipGetLastError() == hipSuccess ? GNNPP_OK : GNNPP_ERR_LAUNCH;
}

int gnnpp_adam_step(const gnnpp_adam_tensors* t, float* state, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int tick, void* stream) {
    if (!t || !state || t->count <= 0 || t->count > kAdamTensors) return GNNPP_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    AdamTable tb = {};
    int blocks = 0;
    for (int i = 0; i < t->count; ++i) {
        if (!t->p[i] || !t->g[i] || !t->m[i] || !t->v[i] || t->numel[i] <= 0) return GNNPP_ERR_ARG;
        tb.p[i] = t->p[i]; tb.g[i] = t->g[i]; tb.m[i] = t->m[i]; tb.v[i] = t->v[i];
        tb.numel[i] = (long)t->numel[i];
        tb.first[i] = blocks;
        blocks += (int)((t->numel[i] + 1023) / 1024);
    }
    tb.first[t->count] = blocks;
    tb.count = t->count;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 4 * sizeof(float), st, tb, state, lr, beta1, beta2, eps,
                       weight_decay, tick ? 1 : 0);
    return hipGetLastError() == hipSuccess ? GNNPP_OK : GNNPP_ERR_LAUNCH;
}

#ifdef GNNPP_MEASURE
// libgnnpp_measure.so only: copy the kernels' phase time stamps ([1024 workgroups][16 wall + 16 cycle slots], 100 MHz
// ticks) to a HOST buffer of n entries (synchronises the device).
int gnnpp_measure_read_stamps(unsigned long long* host, int n) {
    if (!host || n <= 0 || n > 1024 * 32) return GNNPP_ERR_ARG;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), (size_t)n * sizeof(unsigned long long)) == hipSuccess
               ? GNNPP_OK : GNNPP_ERR_LAUNCH;
}
#endif

int gnnpp_get_tuning(int key) {
    switch (key) {
        case GNNPP_TUNE_FILTER_GPW: return g_filter_gpw.load();
        case GNNPP_TUNE_FILTER_WAVES: return g_filter_waves.load();
        case GNNPP_TUNE_FUSED_POLICY: return g_fused_policy.load();
        case GNNPP_TUNE_FILTER_SPLIT: return g_filter_split.load();
        case GNNPP_TUNE_POLICY_FILTER: return g_filter_policy_kernel.load();
        case GNNPP_TUNE_FILTER_SMALL: return g_filter_small_kernel.load();
        case GNNPP_TUNE_FILTER_SMALL_ROWS: return g_filter_small_rows.load();
        case GNNPP_TUNE_FILTER_PIPE_GRID: return g_filter_pipe_grid.load();
        case GNNPP_TUNE_POLICY_CP: return g_policy_column_packing.load();
        case GNNPP_TUNE_ENCODER_CP_TILE: return g_encoder_cp_tile.load();
        case GNNPP_TUNE_TRAIN_FORK: return g_train_fork.load();
        case GNNPP_TUNE_FILTER_PLANE_ALIAS: return g_filter_plane_alias.load();
        case GNNPP_TUNE_TRAIN_WGRAD_WGS: return g_train_wgrad_wgs.load();
        case GNNPP_TUNE_TRAIN_WGRAD_MERGED: return g_train_wgrad_merged.load();
        case GNNPP_TUNE_TRAIN_RUNNING_FUSED: return g_train_running_fused.load();
#ifdef GNNPP_MEASURE
        case GNNPP_TUNE_FILTER_ABLATE: return g_filter_ablate.load();
        case GNNPP_TUNE_ENCODER_STOP: return g_encoder_stop.load();
#endif
        default: return GNNPP_ERR_ARG;
    }
}

int gnnpp_set_tuning(int key, int value) {
    switch (key) {
        case GNNPP_TUNE_FILTER_GPW:
            if (value < 0) return GNNPP_ERR_ARG;
            g_filter_gpw.store(value);
            return GNNPP_OK;
        case GNNPP_TUNE_FUSED_POLICY:
            if (value < 0 || value > 2) return GNNPP_ERR_ARG;
            g_fused_policy.store(value);
            return GNNPP_OK;
        case GNNPP_TUNE_FILTER_WAVES:
            if (value != 0 && value != 8 && value != 16) return GNNPP_ERR_ARG;
            g_filter_waves.store(value);
            return GNNPP_OK;
        case GNNPP_TUNE_FILTER_SPLIT:
            if (value < 0 || value > 7) return GNNPP_ERR_ARG;
            g_filter_split.store(value);
            return GNNPP_OK;
        case GNNPP_TUNE_POLICY_FILTER:
            if (value < 0 || value > 1) return GNNPP_ERR_ARG;
            g_filter_policy_kernel.store(value);
            return GNNPP_OK;
        case GNNPP_TUNE_FILTER_SMALL_ROWS:
            if (value != 0 && value != 32 && value != 48 && value != 64) return GNNPP_ERR_ARG;
            g_filter_small_rows.store(value);
            return GNNPP_OK;
        case GNNPP_TUNE_FILTER_PIPE_GRID:
            if (value < 0 || value > 4096) return GNNPP_ERR_ARG;
            g_filter_pipe_grid.store(value);
            return GNNPP_OK;
        case GNNPP_TUNE_POLICY_CP:
            if (value != 0 && value != 1) return GNNPP_ERR_ARG;
            g_policy_column_packing.store(value);
            return GNNPP_OK;
        case GNNPP_TUNE_ENCODER_CP_TILE:
            if (value != 0 && value != 16 && (value < 1 || value > 12)) return GNNPP_ERR_ARG;
            g_encoder_cp_tile.store(value);
            return GNNPP_OK;
        case GNNPP_TUNE_TRAIN_FORK:
            if (value < 0 || value > 2) return GNNPP_ERR_ARG;
            g_train_fork.store(value);
            return GNNPP_OK;
        case GNNPP_TUNE_FILTER_PLANE_ALIAS:
            if (value < 0 || value > 1) return GNNPP_ERR_ARG;
            g_filter_plane_alias.store(value);
            return GNNPP_OK;
        case GNNPP_TUNE_TRAIN_WGRAD_MERGED:
            if (value < 0 || value > 1) return GNNPP_ERR_ARG;
            g_train_wgrad_merged.store(value);
            return GNNPP_OK;
        case GNNPP_TUNE_TRAIN_RUNNING_FUSED:
            if (value < 0 || value > 1) return GNNPP_ERR_ARG;
            g_train_running_fused.store(value);
            return GNNPP_OK;
        case GNNPP_TUNE_TRAIN_WGRAD_WGS:
            if (value != 0 && (value < 16 || value > 2048)) return GNNPP_ERR_ARG;
            g_train_wgrad_wgs.store(value);
            return GNNPP_OK;
        case GNNPP_TUNE_FILTER_SMALL:
            if (value < 0 || value > 3) return GNNPP_ERR_ARG;
            g_filter_small_kernel.store(value);
            return GNNPP_OK;
#ifdef GNNPP_MEASURE
        case GNNPP_TUNE_ENCODER_STOP:
            if (value < 0 || value > 6) return GNNPP_ERR_ARG;
            g_encoder_stop.store(value);
            return GNNPP_OK;
        case GNNPP_TUNE_FILTER_ABLATE:
            g_filter_ablate.store(value);
            return GNNPP_OK;
#endif
        default:
            return GNNPP_ERR_ARG;
    }
}

int gnnpp_decode_actions(const float* logits, int* actions, int B, int N, void* stream) {
    if (!logits || !actions || B <= 0 || N <= 0) return GNNPP_ERR_ARG;
    return decode_actions_launch(logits, actions, B, N, static_cast<hipStream_t>(stream));
}

static int rollout_common_ok(const gnnpp_rollout* r) {
    return r && r->pos && r->B > 0 && r->N > 0 && r->N <= GNNPP_ROLLOUT_MAX_AGENTS;
}

int gnnpp_rollout_observe(const gnnpp_rollout* r, void* stream) {
    if (!rollout_common_ok(r) || !r->grid || !r->goal || !r->obs || r->H <= 0 || r->W <= 0)
        return GNNPP_ERR_ARG;
    return rollout_observe_launch(*r, static_cast<hipStream_t>(stream));
}

int gnnpp_rollout_gso(const gnnpp_rollout* r, void* stream) {
    if (!rollout_common_ok(r) || !r->radius || !r->S) return GNNPP_ERR_ARG;
    return rollout_gso_launch(*r, static_cast<hipStream_t>(stream));
}

int gnnpp_rollout_gso_observe(const gnnpp_rollout* r, void* stream) {
    if (!rollout_common_ok(r) || !r->radius || !r->S || !r->grid || !r->goal || !r->obs || r->H <= 0 || r->W <= 0)
        return GNNPP_ERR_ARG;
    return rollout_gso_observe_launch(*r, static_cast<hipStream_t>(stream));
}

int gnnpp_rollout_move(const gnnpp_rollout* r, void* stream) {
    if (!rollout_common_ok(r) || !r->grid || !r->goal || (!r->logits && !r->actions) ||
        !r->reached || !r->start_step || !r->end_step || !r->maxstep || !r->flags || !r->stats ||
        r->H <= 0 || r->W <= 0)
        return GNNPP_ERR_ARG;
    if (r->tie_mode == GNNPP_TIE_REPLAY && (!r->choices || r->max_choices <= 0)) return GNNPP_ERR_ARG;
    if (r->tie_mode < 0 || r->tie_mode > 3) return GNNPP_ERR_ARG;
    if (r->tie_mode == GNNPP_TIE_MT19937 && (!r->rng_words || !r->rng_cursor || r->rng_max <= 0)) return GNNPP_ERR_ARG;
    return rollout_move_launch(*r, static_cast<hipStream_t>(stream));
}

int gnnpp_rollout_step(const gnnpp_rollout* r, void* stream) {
    if (!rollout_common_ok(r) || !r->grid || !r->goal || !r->obs || !r->radius || !r->S ||
        (!r->logits && !r->actions) || !r->reached || !r->start_step || !r->end_step || !r->maxstep ||
        !r->flags || !r->stats || r->H <= 0 || r->W <= 0)
        return GNNPP_ERR_ARG;
    if (r->tie_mode == GNNPP_TIE_REPLAY && (!r->choices || r->max_choices <= 0)) return GNNPP_ERR_ARG;
    if (r->tie_mode < 0 || r->tie_mode > 3) return GNNPP_ERR_ARG;
    if (r->tie_mode == GNNPP_TIE_MT19937 && (!r->rng_words || !r->rng_cursor || r->rng_max <= 0)) return GNNPP_ERR_ARG;
    return rollout_step_launch(*r, static_cast<hipStream_t>(stream));
}

int gnnpp_rollout_policy_step(const gnnpp_rollout* r, const float* enc_packed, const float* filt_packed,
                              const float* gf_bias, const float* act_w, const float* act_b, int K,
                              int precision, void* stream) {
    if (!rollout_common_ok(r) || !r->grid || !r->goal || !r->obs || !r->radius || !r->S || !r->logits ||
        !r->reached || !r->start_step || !r->end_step || !r->maxstep || !r->flags || !r->stats ||
        r->H <= 0 || r->W <= 0 || !enc_packed || !filt_packed || !act_w || !act_b)
        return GNNPP_ERR_ARG;
    if (r->tie_mode == GNNPP_TIE_REPLAY && (!r->choices || r->max_choices <= 0)) return GNNPP_ERR_ARG;
    if (r->tie_mode < 0 || r->tie_mode > 3) return GNNPP_ERR_ARG;
    if (r->tie_mode == GNNPP_TIE_MT19937 && (!r->rng_words || !r->rng_cursor || r->rng_max <= 0)) return GNNPP_ERR_ARG;
    if (precision < 0 || precision > 2) return GNNPP_ERR_ARG;
    // same conditions as the fused policy kernel of gnnpp_policy_fwd, plus room for the occupancy grid
    const size_t occ_room = precision == kPrecFp32 ? policy_sim_occ_bytes_b3() : policy_sim_occ_bytes(K);
    if (!(fused_policy_applies(r->B, r->N, K, precision) && (size_t)r->H * r->W <= occ_room))
        return GNNPP_ERR_UNSUPPORTED;
    PolicyTail pt;
    pt.S = r->S;
    pt.filt_h2 = filt_packed + (precision == kPrecFp32 ? filter_packed_b3_offset(GNNPP_FEAT, GNNPP_FEAT, K, 1)
                                                       : filter_packed_f32_floats(GNNPP_FEAT, GNNPP_FEAT, K, 1));
    pt.gf_bias = gf_bias; pt.act_w = act_w; pt.act_b = act_b; pt.logits = const_cast<float*>(r->logits);
    pt.B = r->B; pt.N = r->N; pt.s_is_f64 = 0;
    pt.range_flag = r->range_flag;
    pt.with_sim = 1;
    pt.sim = *r;
    pt.sim.grow = 0;
    pt.sim.actions = nullptr;
    return precision == kPrecFp32 ? policy_launch_fused_b3(r->obs, enc_packed, pt, K, static_cast<hipStream_t>(stream))
                                  : policy_launch_fused(r->obs, enc_packed, pt, K, static_cast<hipStream_t>(stream));
}

int gnnpp_rollout_policy_steps(const gnnpp_rollout* r, const float* enc_packed, const float* filt_packed,
                               const float* gf_bias, const float* act_w, const float* act_b, int K,
                               int nsteps, int precision, void* stream) {
    if (!r || nsteps <= 0 || r->tie_mode == GNNPP_TIE_REPLAY) return GNNPP_ERR_ARG;
    gnnpp_rollout rs = *r;
    for (int s = 0; s < nsteps; ++s) {
        rs.currentstep = r->currentstep + s;
        const int rc = gnnpp_rollout_policy_step(&rs, enc_packed, filt_packed, gf_bias, act_w, act_b, K, precision,
                                                 stream);
        if (rc != GNNPP_OK) return rc;                   // (argument / shape errors surface at s = 0)
    }
    return GNNPP_OK;
}

}  // extern "C"

// Parakeet-TDT (FastConformer encoder + TDT greedy decode) as a native kernel graph.
//
// Replaces what the reference obtains from nano_parakeet's model.transcribe(audio)
// (STT/parakeet_tdt_handler.py:621-623; smart_progressive_streaming.py:82-85).  Arithmetic
// restated from transformers/models/parakeet/modeling_parakeet.py (:68-106, :124-193, :267-362,
// :365-431, :434-479, :831-895, :1035-1052) and generation_parakeet.py:113-122, 271-294.
//
// HBM layout for a batch of B sessions with T feature frames (T' = ceil(ceil(ceil(T/2)/2)/2)):
//   feats [B][T][n_mels]                      s1 [B][T/2][n_mels/2][C]  s2a/s2b [B][T/4][n_mels/4][C]
//   s3a/s3b [B][T'][n_mels/8][C] (= GEMM rows [B*T'][F*C] for the flatten+Linear)
//   x [B*T'][d] f32 residual stream;  R[l] [2*Tm-1][d] relative-position keys (pre-projected,
//   pre-scaled, shared by every session and length);  bd [B][H][T'][2T'-1] f32 position scores
#include "ops.h"

#include <map>
#include <string>
#include <vector>

namespace s2s {

struct PTensor {
    const void* ptr;
    long numel;
};

struct Parakeet {
    s2s_parakeet_config cfg;
    int max_batch, esz;
    int tmax_feat, tmax_enc;   // max feature frames / encoder frames
    std::map<std::string, PTensor> w;
    bool finalized = false;
    char* arena = nullptr;
    size_t arena_bytes = 0;
    // encoder workspace
    void *s1, *s2a, *s2b, *s3a, *s3b, *h, *qkv, *qv, *vt, *att, *ffn, *cy, *cg;
    float *x, *bd, *encp;
    int* enc_len;
    int cur_B = 0, cur_T = 0;   // of the last encode
    // decoder state
    float *dec, *dec_new, *gates, *logits;
    float *hst[4], *cst[4], *hnew[4], *cnew[4];
    void *jin, *lx;
    int *frame, *finished, *tok, *adv, *n_steps, *n_active, *o_tok, *o_dur, *o_frame;
    float** d_dst; const float** d_src;   // device pointer tables for the commit kernel
    int max_steps_alloc;
    int* host_flag = nullptr;
    // S2S_TDT_GRAPH=1 (round 4, unmeasured, off by default): one captured TDT step per batch size; the utterance's frame count and the
    // step budget are read from tdt_dyn {T, max_steps} in device memory, so the graph does not depend on the utterance
    int* tdt_dyn = nullptr;
    hipStream_t cap_stream = nullptr;
    std::map<int, hipGraphExec_t> tdt_graphs;
};

static const PTensor* pfind(const Parakeet* m, const std::string& n) {
    auto it = m->w.find(n);
    return it == m->w.end() ? nullptr : &it->second;
}
static const void* PP(const Parakeet* m, const std::string& n) { return pfind(m, n)->ptr; }
static const float* PPF(const Parakeet* m, const std::string& n) { return (const float*)pfind(m, n)->ptr; }
static std::string L(int l, const char* r) { return "enc." + std::to_string(l) + "." + r; }

static int need(const Parakeet* m, const std::string& n, long numel) {
    const PTensor* t = pfind(m, n);
    if (!t) return fail(S2S_ERR_STATE, "parakeet: missing tensor %s", n.c_str());
    if (t->numel != numel) return fail(S2S_ERR_STATE, "parakeet: tensor %s has %ld elements, expected %ld", n.c_str(), t->numel, numel);
    return S2S_OK;
}

static int check_weights(const Parakeet* m) {
    const auto& c = m->cfg;
    const long d = c.d_model, C = c.sub_channels, f = c.ffn, Hd = c.dec_hidden, F3 = c.n_mels / 8;
    int rc;
#define NEED(n, k) if ((rc = need(m, n, k))) return rc
    NEED("sub.conv1.w", C * 9); NEED("sub.conv1.b", C);
    NEED("sub.dw2.w", C * 9); NEED("sub.dw2.b", C); NEED("sub.pw2.w", C * C); NEED("sub.pw2.b", C);
    NEED("sub.dw3.w", C * 9); NEED("sub.dw3.b", C); NEED("sub.pw3.w", C * C); NEED("sub.pw3.b", C);
    NEED("sub.lin.w", d * F3 * C); NEED("sub.lin.b", d);
    for (int l = 0; l < c.layers; ++l) {
        for (const char* n : {"ln_ff1", "ln_att", "ln_conv", "ln_ff2", "ln_out"}) {
            NEED(L(l, n) + ".g", d); NEED(L(l, n) + ".b", d);
        }
        for (const char* n : {"ff1", "ff2"}) {
            NEED(L(l, n) + ".fc1.w", f * d); NEED(L(l, n) + ".fc1.b", f);
            NEED(L(l, n) + ".fc2.w", f * d); NEED(L(l, n) + ".fc2.b", d);
        }
        NEED(L(l, "qkv.w"), 3 * d * d); NEED(L(l, "qkv.b"), 3 * d); NEED(L(l, "vmu"), d);
        NEED(L(l, "R"), (long)(2 * m->tmax_enc - 1) * d);
        NEED(L(l, "o.w"), d * d); NEED(L(l, "o.b"), d);
        NEED(L(l, "pw1.w"), 2 * d * d); NEED(L(l, "pw1.b"), 2 * d);
        NEED(L(l, "dw.w"), d * c.conv_kernel); NEED(L(l, "dw.b"), d);
        NEED(L(l, "pw2.w"), d * d); NEED(L(l, "pw2.b"), d);
    }
    NEED("dec.encp.w", Hd * d); NEED("dec.encp.b", Hd);
    NEED("dec.emb", (long)c.vocab * Hd);
    for (int l = 0; l < c.dec_layers; ++l) {
        NEED("dec.lstm." + std::to_string(l) + ".w", 4 * Hd * 2 * Hd);
        NEED("dec.lstm." + std::to_string(l) + ".b", 4 * Hd);
    }
    NEED("dec.dproj.w", Hd * Hd); NEED("dec.dproj.b", Hd);
    NEED("dec.joint.w", (long)(c.vocab + c.n_durations) * Hd); NEED("dec.joint.b", c.vocab + c.n_durations);
#undef NEED
    return S2S_OK;
}

static inline int half_up(int n) { return (n - 1) / 2 + 1; }

using Plan = ArenaPlan;      // (common.h: the shared carve-up, with poisoned gaps under S2S_GUARD)

static int finalize(Parakeet* m) {
    int rc = check_weights(m);
    if (rc) return rc;
    const auto& c = m->cfg;
    const size_t B = m->max_batch, e = m->esz, d = c.d_model, C = c.sub_channels;
    const size_t T = m->tmax_feat, T1 = half_up(T), T2 = half_up(T1), T3 = half_up(T2);
    const size_t F = c.n_mels, F1 = F / 2, F2 = F / 4, F3 = F / 8;
    const size_t Tp = (T3 + 63) / 64 * 64, Hd = c.dec_hidden, NL = c.dec_layers;
    const size_t V = c.vocab + c.n_durations;
    m->max_steps_alloc = 10 * (int)T3;
    Plan a;
    size_t o[64]; int k = 0;
    o[k++] = a.take(B * T1 * F1 * C * e);      // 0 s1
    o[k++] = a.take(B * T2 * F2 * C * e);      // 1 s2a
    o[k++] = a.take(B * T2 * F2 * C * e);      // 2 s2b
    o[k++] = a.take(B * T3 * F3 * C * e);      // 3 s3a
    o[k++] = a.take(B * T3 * F3 * C * e);      // 4 s3b
    o[k++] = a.take(B * T3 * d * 4);           // 5 x
    o[k++] = a.take(B * T3 * d * e);           // 6 h
    o[k++] = a.take(B * T3 * 3 * d * e);       // 7 qkv
    o[k++] = a.take(B * T3 * d * e);           // 8 qv
    o[k++] = a.take(B * d * Tp * e);           // 9 vt
    o[k++] = a.take(B * T3 * d * e);           // 10 att
    o[k++] = a.take(B * T3 * (size_t)c.ffn * e);  // 11 ffn
    o[k++] = a.take(B * T3 * 2 * d * e);       // 12 cy
    o[k++] = a.take(B * T3 * d * e);           // 13 cg
    o[k++] = a.take(B * c.n_heads * T3 * (2 * T3 - 1) * 4);   // 14 bd
    o[k++] = a.take(B * T3 * Hd * 4);          // 15 encp
    o[k++] = a.take(B * 4);                    // 16 enc_len
    o[k++] = a.take(B * Hd * 4);               // 17 dec
    o[k++] = a.take(B * Hd * 4);               // 18 dec_new
    o[k++] = a.take(B * 4 * Hd * 4);           // 19 gates
    o[k++] = a.take(B * V * 4);                // 20 logits
    o[k++] = a.take(B * Hd * e);               // 21 jin
    o[k++] = a.take(B * 2 * Hd * e);           // 22 lx
    o[k++] = a.take(NL * 4 * B * Hd * 4);      // 23 lstm state: h, c, hnew, cnew per layer
    o[k++] = a.take((6 * B + 8) * 4);          // 24 ints
    o[k++] = a.take(3 * B * (size_t)m->max_steps_alloc * 4);   // 25 outputs
    o[k++] = a.take(64 * sizeof(void*));       // 26 pointer tables
    m->arena_bytes = a.off;
    if ((rc = a.commit(&m->arena, "parakeet"))) return rc;
    char* b0 = m->arena;
    m->s1 = b0 + o[0]; m->s2a = b0 + o[1]; m->s2b = b0 + o[2]; m->s3a = b0 + o[3]; m->s3b = b0 + o[4];
    m->x = (float*)(b0 + o[5]); m->h = b0 + o[6]; m->qkv = b0 + o[7]; m->qv = b0 + o[8]; m->vt = b0 + o[9];
    m->att = b0 + o[10]; m->ffn = b0 + o[11]; m->cy = b0 + o[12]; m->cg = b0 + o[13];
    m->bd = (float*)(b0 + o[14]); m->encp = (float*)(b0 + o[15]); m->enc_len = (int*)(b0 + o[16]);
    m->dec = (float*)(b0 + o[17]); m->dec_new = (float*)(b0 + o[18]); m->gates = (float*)(b0 + o[19]);
    m->logits = (float*)(b0 + o[20]); m->jin = b0 + o[21]; m->lx = b0 + o[22];
    float* ls = (float*)(b0 + o[23]);
    for (size_t l = 0; l < NL && l < 4; ++l) {
        m->hst[l] = ls + (l * 4 + 0) * B * Hd; m->cst[l] = ls + (l * 4 + 1) * B * Hd;
        m->hnew[l] = ls + (l * 4 + 2) * B * Hd; m->cnew[l] = ls + (l * 4 + 3) * B * Hd;
    }
    int* ints = (int*)(b0 + o[24]);
    m->frame = ints; m->finished = ints + B; m->tok = ints + 2 * B; m->adv = ints + 3 * B; m->n_steps = ints + 4 * B;
    m->n_active = ints + 5 * B;
    m->tdt_dyn = ints + 5 * B + 4;
    int* outs = (int*)(b0 + o[25]);
    m->o_tok = outs; m->o_dur = outs + B * m->max_steps_alloc; m->o_frame = outs + 2 * B * m->max_steps_alloc;
    // pointer tables for the commit kernel: dst = {h_l, c_l ..., dec}, src = {hnew_l, cnew_l ..., dec_new}
    std::vector<void*> tab;
    for (size_t l = 0; l < NL; ++l) { tab.push_back(m->hst[l]); tab.push_back(m->cst[l]); }
    tab.push_back(m->dec);
    for (size_t l = 0; l < NL; ++l) { tab.push_back(m->hnew[l]); tab.push_back(m->cnew[l]); }
    tab.push_back(m->dec_new);
    m->d_dst = (float**)(b0 + o[26]);
    m->d_src = (const float**)(m->d_dst + (2 * NL + 1));
    S2S_HIP_CHECK(hipMemcpy(m->d_dst, tab.data(), tab.size() * sizeof(void*), hipMemcpyHostToDevice));
    S2S_HIP_CHECK(hipHostMalloc((void**)&m->host_flag, 64, 0));
    m->finalized = true;
    return S2S_OK;
}

static inline void* at(void* p, size_t bytes) { return (char*)p + bytes; }

static int ffn_half(Parakeet* m, hipStream_t st, int l, const char* ln, const char* ff, int M) {
    const auto& c = m->cfg;
    const int d = c.d_model, dt = c.dtype;
    int rc;
    const std::string lnn = L(l, ln), ffn = L(l, ff);
    if ((rc = layernorm(st, m->x, PPF(m, lnn + ".g"), PPF(m, lnn + ".b"), m->h, M, d, 1e-5f, dt))) return rc;
    if ((rc = gemm(st, m->h, d, PP(m, ffn + ".fc1.w"), PPF(m, ffn + ".fc1.b"), nullptr, m->ffn, c.ffn, M, c.ffn, d, dt, dt, S2S_ACT_SILU, false))) return rc;
    GemmArgs g{};
    g.A = m->ffn; g.lda = c.ffn; g.W = PP(m, ffn + ".fc2.w"); g.bias = PPF(m, ffn + ".fc2.b");
    g.residual = m->x; g.C = m->x; g.ldc = d; g.M = M; g.N = d; g.K = c.ffn;
    g.ab_dtype = dt; g.c_dtype = S2S_F32; g.act = S2S_ACT_NONE; g.alpha = 0.5f; g.alpha_set = true;
    return gemm_ex(st, g);
}

static int encode(Parakeet* m, hipStream_t st, const void* feats, const int* n_frames, int B, int T) {
    const auto& c = m->cfg;
    const int d = c.d_model, H = c.n_heads, D = d / H, dt = c.dtype, e = m->esz, C = c.sub_channels;
    const int T1 = half_up(T), T2 = half_up(T1), T3 = half_up(T2);
    const int F = c.n_mels, F1 = F / 2, F2 = F / 4, F3 = F / 8;
    int rc;
    // ---- dw-striding subsampling x8
    if ((rc = sub_conv1(st, feats, PPF(m, "sub.conv1.w"), PPF(m, "sub.conv1.b"), m->s1, n_frames, B, T, F, C, dt))) return rc;
    if ((rc = dwconv2d(st, m->s1, PPF(m, "sub.dw2.w"), PPF(m, "sub.dw2.b"), m->s2a, n_frames, 1, B, T1, F1, C, dt))) return rc;
    if ((rc = gemm(st, m->s2a, C, PP(m, "sub.pw2.w"), PPF(m, "sub.pw2.b"), nullptr, m->s2b, C, B * T2 * F2, C, C, dt, dt, S2S_ACT_RELU, false))) return rc;
    if ((rc = dwconv2d(st, m->s2b, PPF(m, "sub.dw3.w"), PPF(m, "sub.dw3.b"), m->s3a, n_frames, 2, B, T2, F2, C, dt))) return rc;
    if ((rc = gemm(st, m->s3a, C, PP(m, "sub.pw3.w"), PPF(m, "sub.pw3.b"), nullptr, m->s3b, C, B * T3 * F3, C, C, dt, dt, S2S_ACT_RELU, false))) return rc;
    if ((rc = mask_rows(st, m->s3b, n_frames, 3, B, T3, (long)F3 * C, dt))) return rc;
    if ((rc = sub_lengths(st, n_frames, 3, m->enc_len, B))) return rc;
    const int M = B * T3;
    // flatten [f][c] + Linear (+ sqrt(d) input scale folded into the weights)
    if ((rc = gemm(st, m->s3b, F3 * C, PP(m, "sub.lin.w"), PPF(m, "sub.lin.b"), nullptr, m->x, d, M, d, F3 * C, dt, S2S_F32, S2S_ACT_NONE, false))) return rc;

    const int Tp = (T3 + 63) / 64 * 64;
    const float scale = 1.0f / sqrtf((float)D);
    const long bd_rs = 2L * T3 - 1;
    for (int l = 0; l < c.layers; ++l) {
        if ((rc = ffn_half(m, st, l, "ln_ff1", "ff1", M))) return rc;
        // ---- relative-position self-attention
        if ((rc = layernorm(st, m->x, PPF(m, L(l, "ln_att.g")), PPF(m, L(l, "ln_att.b")), m->h, M, d, 1e-5f, dt))) return rc;
        if ((rc = gemm(st, m->h, d, PP(m, L(l, "qkv.w")), PPF(m, L(l, "qkv.b")), nullptr, m->qkv, 3 * d, M, 3 * d, d, dt, dt, S2S_ACT_NONE, false))) return rc;
        if ((rc = add_rowvec(st, m->qkv, 3 * d, PPF(m, L(l, "vmu")), m->qv, M, d, dt))) return rc;
        {   // bd[b][h][i][n] = (q_i + v) . R[row0 + n]   (R pre-scaled by head_dim^-0.5)
            GemmArgs g{};
            g.A = m->qv; g.lda = d; g.W = at(const_cast<void*>(PP(m, L(l, "R"))), (size_t)(m->tmax_enc - T3) * d * e); g.ldw = d;
            g.C = m->bd; g.ldc = bd_rs; g.M = T3; g.N = 2 * T3 - 1; g.K = D;
            g.nz1 = B; g.nz2 = H; g.a_z1 = (long)T3 * d; g.a_z2 = D; g.w_z1 = 0; g.w_z2 = D;
            g.c_z1 = (long)H * T3 * bd_rs; g.c_z2 = (long)T3 * bd_rs;
            g.ab_dtype = dt; g.c_dtype = S2S_F32; g.act = S2S_ACT_NONE;
            if ((rc = gemm_ex(st, g))) return rc;
        }
        if (dt == S2S_BF16 && (D == 64 || D == 128)) {
            const bool direct_v = attn_mfma_direct_v() && d % 8 == 0;      // V rows straight from the qkv projection (round 6): no transpose pass
            if (!direct_v && (rc = transpose_v(st, at(m->qkv, (size_t)2 * d * e), (long)T3 * 3 * d, 3 * d, m->vt, B, H, D, T3, Tp, dt))) return rc;
            AttnMfmaParams p{};
            p.Q = (const __hip_bfloat16*)m->qkv; p.q_bs = (long)T3 * 3 * d; p.q_rs = 3 * d;
            p.K = (const __hip_bfloat16*)at(m->qkv, (size_t)d * e); p.k_bs = p.q_bs; p.k_rs = 3 * d;
            p.Vt = direct_v ? nullptr : (const __hip_bfloat16*)m->vt;
            p.V = (const __hip_bfloat16*)at(m->qkv, (size_t)2 * d * e); p.v_bs = p.q_bs; p.v_rs = 3 * d;
            p.O = (__hip_bfloat16*)m->att; p.o_bs = (long)T3 * d; p.o_rs = d;
            p.bias = m->bd; p.bias_bs = (long)H * T3 * bd_rs; p.bias_hs = (long)T3 * bd_rs; p.bias_rs = bd_rs; p.bias_shift = T3 - 1;
            p.kv_len = m->enc_len; p.Tk = T3; p.Tp = Tp; p.Tq = T3; p.H = H; p.B = B;
            p.scale_log2e = scale * 1.4426950408889634f;
            if ((rc = attn_mfma(st, p, D))) return rc;
        } else {
            AttnRowsParams p{};
            p.Q = m->qkv; p.q_bs = (long)T3 * 3 * d; p.q_rs = 3 * d;
            p.K = at(m->qkv, (size_t)d * e); p.V = at(m->qkv, (size_t)2 * d * e); p.kv_bs = p.q_bs; p.kv_rs = 3 * d;
            p.O = m->att; p.o_bs = (long)T3 * d; p.o_rs = d;
            p.bias = m->bd; p.bias_bs = (long)H * T3 * bd_rs; p.bias_hs = (long)T3 * bd_rs; p.bias_rs = bd_rs; p.bias_shift = T3 - 1;
            p.kv_len = m->enc_len; p.kv_len_const = T3; p.B = B; p.H = H; p.Hkv = H; p.Tq = T3; p.D = D; p.scale = scale;
            if ((rc = attn_rows(st, p, dt))) return rc;
        }
        if ((rc = gemm(st, m->att, d, PP(m, L(l, "o.w")), PPF(m, L(l, "o.b")), m->x, m->x, d, M, d, d, dt, S2S_F32, S2S_ACT_NONE, false))) return rc;
        // ---- convolution module
        if ((rc = layernorm(st, m->x, PPF(m, L(l, "ln_conv.g")), PPF(m, L(l, "ln_conv.b")), m->h, M, d, 1e-5f, dt))) return rc;
        if ((rc = gemm(st, m->h, d, PP(m, L(l, "pw1.w")), PPF(m, L(l, "pw1.b")), nullptr, m->cy, 2 * d, M, 2 * d, d, dt, dt, S2S_ACT_NONE, false))) return rc;
        if ((rc = glu_dwconv(st, m->cy, PPF(m, L(l, "dw.w")), PPF(m, L(l, "dw.b")), m->cg, m->enc_len, B, T3, d, c.conv_kernel, dt))) return rc;
        if ((rc = gemm(st, m->cg, d, PP(m, L(l, "pw2.w")), PPF(m, L(l, "pw2.b")), m->x, m->x, d, M, d, d, dt, S2S_F32, S2S_ACT_NONE, false))) return rc;
        if ((rc = ffn_half(m, st, l, "ln_ff2", "ff2", M))) return rc;
        if ((rc = layernorm(st, m->x, PPF(m, L(l, "ln_out.g")), PPF(m, L(l, "ln_out.b")), m->x, M, d, 1e-5f, S2S_F32))) return rc;
    }
    // encoder_projector applied once to every frame (joint network input)
    if ((rc = cast_f32(st, m->x, m->h, (long)M * d, dt))) return rc;
    if ((rc = gemm(st, m->h, d, PP(m, "dec.encp.w"), PPF(m, "dec.encp.b"), nullptr, m->encp, c.dec_hidden, M, c.dec_hidden, d, dt, S2S_F32, S2S_ACT_NONE, false))) return rc;
    m->cur_B = B;
    m->cur_T = T3;
    return S2S_OK;
}

// prediction network: (tok -> embedding) through the LSTM stack and decoder_projector into the staging buffers
static int prednet(Parakeet* m, hipStream_t st, int B) {
    const auto& c = m->cfg;
    const int Hd = c.dec_hidden, dt = c.dtype;
    int rc;
    for (int l = 0; l < c.dec_layers; ++l) {
        const std::string n = "dec.lstm." + std::to_string(l);
        if ((rc = lstm_in(st, l == 0 ? PP(m, "dec.emb") : nullptr, m->tok, l == 0 ? nullptr : m->hnew[l - 1], m->hst[l], m->lx, B, Hd, dt))) return rc;
        if ((rc = gemm(st, m->lx, 2 * Hd, PP(m, n + ".w"), PPF(m, n + ".b"), nullptr, m->gates, 4 * Hd, B, 4 * Hd, 2 * Hd, dt, S2S_F32, S2S_ACT_NONE, false))) return rc;
        if ((rc = lstm_cell(st, m->gates, m->cst[l], m->hnew[l], m->cnew[l], B, Hd))) return rc;
    }
    if ((rc = cast_f32(st, m->hnew[c.dec_layers - 1], m->jin, (long)B * Hd, dt))) return rc;
    return gemm(st, m->jin, Hd, PP(m, "dec.dproj.w"), PPF(m, "dec.dproj.b"), nullptr, m->dec_new, Hd, B, Hd, Hd, dt, S2S_F32, S2S_ACT_NONE, false);
}

// the prediction network inside the fused step: lx already holds layer 0's input (tdt_select_in), every cell writes the next GEMM's input
static int prednet_fused(Parakeet* m, hipStream_t st, int B) {
    const auto& c = m->cfg;
    const int Hd = c.dec_hidden, dt = c.dtype;
    int rc;
    for (int l = 0; l < c.dec_layers; ++l) {
        const std::string n = "dec.lstm." + std::to_string(l);
        const bool last = l + 1 == c.dec_layers;
        if ((rc = gemm(st, m->lx, 2 * Hd, PP(m, n + ".w"), PPF(m, n + ".b"), nullptr, m->gates, 4 * Hd, B, 4 * Hd, 2 * Hd, dt, S2S_F32, S2S_ACT_NONE, false))) return rc;
        if ((rc = lstm_cell_out(st, m->gates, m->cst[l], m->hnew[l], m->cnew[l], B, Hd, last ? m->jin : m->lx, last ? nullptr : m->hst[l + 1], dt))) return rc;
    }
    return gemm(st, m->jin, Hd, PP(m, "dec.dproj.w"), PPF(m, "dec.dproj.b"), nullptr, m->dec_new, Hd, B, Hd, Hd, dt, S2S_F32, S2S_ACT_NONE, false);
}

static bool tdt_fused() {       // S2S_TDT_FUSED=0: the 14-launch step of rounds 1 - 5 (A/B)
    static const bool on = getenv("S2S_TDT_FUSED") == nullptr || atoi(getenv("S2S_TDT_FUSED")) != 0;
    return on;
}

__global__ void fill_int_kernel(int* p, int v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void tdt_set_dyn_kernel(int* dyn, int T, int max_steps) {
    dyn[0] = T;
    dyn[1] = max_steps;
}

// one TDT step: joint network on the current frame, token / duration pick, prediction network on the pick, commit where it advanced.
// dyn != null: the captured variant (frame count / step budget from device memory, the active counter reset by a kernel)
static int tdt_step_body(Parakeet* m, hipStream_t st, int B, int T, int max_steps, const TdtState& s, const int* dyn) {
    const auto& c = m->cfg;
    const int Hd = c.dec_hidden, dt = c.dtype, V = c.vocab, ND = c.n_durations, NA = 2 * c.dec_layers + 1;
    int rc;
    if (tdt_fused()) {      // 8 launches: the joint input of this step was written by the previous step's commit (or the prologue's)
        if ((rc = gemm(st, m->jin, Hd, PP(m, "dec.joint.w"), PPF(m, "dec.joint.b"), nullptr, m->logits, V + ND, B, V + ND, Hd, dt, S2S_F32, S2S_ACT_NONE, false))) return rc;
        if (dyn) {      // captured step: the host polls the active-row counter after a whole replay, so it is reset in FRONT of the pick, not behind the commit
            fill_int_kernel<<<1, 64, 0, st>>>(m->n_active, 0, 1);
            S2S_LAUNCH_CHECK();
        }
        if ((rc = tdt_select_in(st, m->logits, V + ND, V, ND, c.vocab - 1, m->enc_len, s, m->o_tok, m->o_dur, m->o_frame, max_steps, B, dyn, PP(m, "dec.emb"), m->hst[0], m->lx, Hd, dt))) return rc;
        if ((rc = prednet_fused(m, st, B))) return rc;
        return tdt_commit_joint(st, m->d_dst, m->d_src, NA, m->adv, 0, B, Hd, m->encp, m->frame, m->jin, T, dyn, dyn ? nullptr : m->n_active, dt);
    }
    if ((rc = tdt_joint_in(st, m->encp, m->dec, m->frame, m->jin, B, T, Hd, dt, dyn))) return rc;
    if ((rc = gemm(st, m->jin, Hd, PP(m, "dec.joint.w"), PPF(m, "dec.joint.b"), nullptr, m->logits, V + ND, B, V + ND, Hd, dt, S2S_F32, S2S_ACT_NONE, false))) return rc;
    if (dyn) {
        fill_int_kernel<<<1, 64, 0, st>>>(m->n_active, 0, 1);
        S2S_LAUNCH_CHECK();
    } else {
        S2S_HIP_CHECK(hipMemsetAsync(m->n_active, 0, sizeof(int), st));
    }
    if ((rc = tdt_select(st, m->logits, V + ND, V, ND, c.vocab - 1, m->enc_len, s, m->o_tok, m->o_dur, m->o_frame, max_steps, B, dyn))) return rc;
    if ((rc = prednet(m, st, B))) return rc;
    return tdt_commit(st, m->d_dst, m->d_src, NA, m->adv, 0, B, Hd);
}

static bool tdt_graph_enabled() {
    static const bool on = getenv("S2S_TDT_GRAPH") != nullptr && atoi(getenv("S2S_TDT_GRAPH")) != 0;
    return on;
}

static int tdt_step_graph(Parakeet* m, int B, const TdtState& s, hipGraphExec_t* out) {
    auto it = m->tdt_graphs.find(B);
    if (it != m->tdt_graphs.end()) {
        *out = it->second;
        return S2S_OK;
    }
    std::lock_guard<std::mutex> capture_lock(capture_mutex());
    if (!m->cap_stream) S2S_HIP_CHECK(hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
    S2S_HIP_CHECK(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = tdt_step_body(m, m->cap_stream, B, /*T=*/1, /*max_steps=*/1, s, m->tdt_dyn);      // (T / max_steps come from tdt_dyn)
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(m->cap_stream, &graph);
    if (rc) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
    }
    S2S_HIP_CHECK(ce);
    hipGraphExec_t exec = nullptr;
    const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    S2S_HIP_CHECK(ie);
    m->tdt_graphs[B] = exec;
    *out = exec;
    return S2S_OK;
}

static int tdt_greedy(Parakeet* m, hipStream_t st, int max_steps, int* out_tok, int* out_dur, int* out_frame, int* out_n) {
    const auto& c = m->cfg;
    const int B = m->cur_B, T = m->cur_T, Hd = c.dec_hidden, dt = c.dtype, V = c.vocab, ND = c.n_durations;
    const int NA = 2 * c.dec_layers + 1;
    int rc;
    if (max_steps <= 0) max_steps = 10 * T - 1;   // generate(): max_length = max_symbols_per_step * T' incl. the start token
    S2S_REQUIRE(max_steps <= m->max_steps_alloc, "tdt_greedy: max_steps %d exceeds %d", max_steps, m->max_steps_alloc);
    // reset state: zero LSTM state, frame 0, start token = blank
    S2S_HIP_CHECK(hipMemsetAsync(m->hst[0], 0, sizeof(float) * (size_t)c.dec_layers * 4 * m->max_batch * Hd, st));
    S2S_HIP_CHECK(hipMemsetAsync(m->frame, 0, sizeof(int) * (6 * (size_t)m->max_batch + 8), st));
    fill_int_kernel<<<cdiv(B, 64), 64, 0, st>>>(m->tok, c.vocab - 1, B);
    S2S_LAUNCH_CHECK();
    if ((rc = prednet(m, st, B))) return rc;
    const bool fused = tdt_fused();
    if (fused) {            // + the joint input of step 0 (frame 0) and a zeroed active-row counter
        if ((rc = tdt_commit_joint(st, m->d_dst, m->d_src, NA, m->adv, /*force=*/1, B, Hd, m->encp, m->frame, m->jin, T, nullptr, m->n_active, dt))) return rc;
    } else if ((rc = tdt_commit(st, m->d_dst, m->d_src, NA, m->adv, /*force=*/1, B, Hd))) return rc;
    TdtState s{m->frame, m->finished, m->tok, m->adv, m->n_steps, m->n_active};
    hipGraphExec_t exec = nullptr;
    bool graph_ok = tdt_graph_enabled() && graphs_enabled() && !prof_active();
    if (graph_ok) {
        tdt_set_dyn_kernel<<<1, 1, 0, st>>>(m->tdt_dyn, T, max_steps);
        S2S_LAUNCH_CHECK();
        if (tdt_step_graph(m, B, s, &exec) != S2S_OK) {      // refused / invalidated capture: this call runs as stream launches
            (void)hipGetLastError();
            graph_ok = false;
        }
    }
    for (int step = 0; step < max_steps; ++step) {
        if (graph_ok) {
            // the whole step is one replay; the prediction network and the commit after the LAST executed pick are harmless (they only
            // touch rows that advanced, and nothing reads the state after the loop)
            S2S_HIP_CHECK(hipGraphLaunch(exec, st));
            if ((step & 15) == 15 || step + 1 == max_steps) {
                S2S_HIP_CHECK(hipMemcpyAsync(m->host_flag, m->n_active, sizeof(int), hipMemcpyDeviceToHost, st));
                S2S_HIP_CHECK(hipStreamSynchronize(st));
                if (*m->host_flag == 0) break;
            }
            continue;
        }
        if (fused) {
            if ((rc = gemm(st, m->jin, Hd, PP(m, "dec.joint.w"), PPF(m, "dec.joint.b"), nullptr, m->logits, V + ND, B, V + ND, Hd, dt, S2S_F32, S2S_ACT_NONE, false))) return rc;
            if ((rc = tdt_select_in(st, m->logits, V + ND, V, ND, c.vocab - 1, m->enc_len, s, m->o_tok, m->o_dur, m->o_frame, max_steps, B, nullptr, PP(m, "dec.emb"), m->hst[0], m->lx, Hd, dt))) return rc;
        } else {
            if ((rc = tdt_joint_in(st, m->encp, m->dec, m->frame, m->jin, B, T, Hd, dt))) return rc;
            if ((rc = gemm(st, m->jin, Hd, PP(m, "dec.joint.w"), PPF(m, "dec.joint.b"), nullptr, m->logits, V + ND, B, V + ND, Hd, dt, S2S_F32, S2S_ACT_NONE, false))) return rc;
            S2S_HIP_CHECK(hipMemsetAsync(m->n_active, 0, sizeof(int), st));
            if ((rc = tdt_select(st, m->logits, V + ND, V, ND, c.vocab - 1, m->enc_len, s, m->o_tok, m->o_dur, m->o_frame, max_steps, B))) return rc;
        }
        if ((step & 15) == 15 || step + 1 == max_steps) {
            S2S_HIP_CHECK(hipMemcpyAsync(m->host_flag, m->n_active, sizeof(int), hipMemcpyDeviceToHost, st));
            S2S_HIP_CHECK(hipStreamSynchronize(st));
            if (*m->host_flag == 0) break;
        }
        if (fused) {
            if ((rc = prednet_fused(m, st, B))) return rc;
            if ((rc = tdt_commit_joint(st, m->d_dst, m->d_src, NA, m->adv, 0, B, Hd, m->encp, m->frame, m->jin, T, nullptr, m->n_active, dt))) return rc;
        } else {
            if ((rc = prednet(m, st, B))) return rc;
            if ((rc = tdt_commit(st, m->d_dst, m->d_src, NA, m->adv, 0, B, Hd))) return rc;
        }
    }
    const size_t ob = sizeof(int) * (size_t)B * max_steps;
    S2S_HIP_CHECK(hipMemcpyAsync(out_tok, m->o_tok, ob, hipMemcpyDeviceToDevice, st));
    S2S_HIP_CHECK(hipMemcpyAsync(out_dur, m->o_dur, ob, hipMemcpyDeviceToDevice, st));
    if (out_frame) S2S_HIP_CHECK(hipMemcpyAsync(out_frame, m->o_frame, ob, hipMemcpyDeviceToDevice, st));
    S2S_HIP_CHECK(hipMemcpyAsync(out_n, m->n_steps, sizeof(int) * B, hipMemcpyDeviceToDevice, st));
    return S2S_OK;
}

}  // namespace s2s

using namespace s2s;

extern "C" int32_t s2s_parakeet_create(const s2s_parakeet_config* cfg, int32_t max_batch, void** out) {
    S2S_REQUIRE(cfg && out && max_batch > 0 && max_batch <= 128, "parakeet_create: bad arguments (max_batch 1..128)");
    S2S_REQUIRE(cfg->dtype == S2S_F32 || cfg->dtype == S2S_BF16, "parakeet_create: bad dtype");
    S2S_REQUIRE(cfg->d_model % cfg->n_heads == 0 && cfg->d_model % 64 == 0 && cfg->n_mels % 8 == 0 && cfg->ffn % 8 == 0 &&
                    cfg->sub_channels % 8 == 0 && cfg->dec_hidden % 8 == 0 && cfg->dec_layers >= 1 && cfg->dec_layers <= 4,
                "parakeet_create: unsupported dimensions");
    S2S_REQUIRE(cfg->max_feat_frames >= 8, "parakeet_create: max_feat_frames too small");
    Parakeet* m = new Parakeet();
    m->cfg = *cfg;
    m->max_batch = max_batch;
    m->esz = cfg->dtype == S2S_F32 ? 4 : 2;
    m->tmax_feat = cfg->max_feat_frames;
    m->tmax_enc = half_up(half_up(half_up(cfg->max_feat_frames)));
    *out = m;
    return S2S_OK;
}

extern "C" int32_t s2s_parakeet_destroy(void* model) {
    if (!model) return S2S_OK;
    Parakeet* m = (Parakeet*)model;
    guard_free(m->arena);
    if (m->host_flag) (void)hipHostFree(m->host_flag);
    for (auto& kv : m->tdt_graphs) (void)hipGraphExecDestroy(kv.second);
    if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
    delete m;
    return S2S_OK;
}

extern "C" int32_t s2s_parakeet_max_encoder_frames(void* model) {
    Parakeet* m = (Parakeet*)model;
    return m ? m->tmax_enc : 0;
}

extern "C" int32_t s2s_parakeet_set_tensor(void* model, const char* name, const void* dev_ptr, int64_t numel) {
    Parakeet* m = (Parakeet*)model;
    S2S_REQUIRE(m && name && dev_ptr, "parakeet_set_tensor: null argument");
    S2S_REQUIRE(((uintptr_t)dev_ptr & 15) == 0, "parakeet_set_tensor: %s is not 16-byte aligned", name);
    m->w[name] = PTensor{dev_ptr, (long)numel};
    return S2S_OK;
}

extern "C" int32_t s2s_parakeet_finalize(void* model) {
    // workspace allocation uses synchronous (legacy-stream) memsets / copies, which HIP refuses - and which poison the other thread's
    // capture - while any stream of the process is capturing: take the capture lock
    std::lock_guard<std::mutex> no_capture(s2s::capture_mutex());
    Parakeet* m = (Parakeet*)model;
    S2S_REQUIRE(m, "parakeet_finalize: null model");
    if (m->finalized) return S2S_OK;
    return finalize(m);
}

extern "C" int64_t s2s_parakeet_workspace_bytes(void* model) {
    Parakeet* m = (Parakeet*)model;
    return m ? (int64_t)m->arena_bytes : 0;
}

extern "C" int32_t s2s_parakeet_encode(void* model, void* stream, const void* feats_dev, const int32_t* n_frames_dev,
                                       int32_t B, int32_t T) {
    Parakeet* m = (Parakeet*)model;
    S2S_REQUIRE(m && m->finalized, "parakeet_encode: model not finalized");
    S2S_REQUIRE(B > 0 && B <= m->max_batch && feats_dev && n_frames_dev, "parakeet_encode: bad batch %d", B);
    S2S_REQUIRE(T >= 8 && T <= m->tmax_feat, "parakeet_encode: T=%d out of range (8..%d)", T, m->tmax_feat);
    return encode(m, (hipStream_t)stream, feats_dev, n_frames_dev, B, T);
}

extern "C" int32_t s2s_parakeet_read_encoder_output(void* model, void* stream, float* dst_dev, int32_t* enc_len_dev) {
    Parakeet* m = (Parakeet*)model;
    S2S_REQUIRE(m && m->finalized && m->cur_B > 0 && dst_dev, "parakeet_read_encoder_output: nothing encoded");
    S2S_HIP_CHECK(hipMemcpyAsync(dst_dev, m->x, sizeof(float) * (size_t)m->cur_B * m->cur_T * m->cfg.d_model, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (enc_len_dev) S2S_HIP_CHECK(hipMemcpyAsync(enc_len_dev, m->enc_len, sizeof(int) * m->cur_B, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return S2S_OK;
}

extern "C" int32_t s2s_parakeet_tdt_greedy(void* model, void* stream, int32_t max_steps, int32_t* tokens_dev,
                                           int32_t* durations_dev, int32_t* frames_dev, int32_t* n_steps_dev) {
    Parakeet* m = (Parakeet*)model;
    S2S_REQUIRE(m && m->finalized && m->cur_B > 0, "parakeet_tdt_greedy: encode first");
    S2S_REQUIRE(tokens_dev && durations_dev && n_steps_dev, "parakeet_tdt_greedy: null output");
    return tdt_greedy(m, (hipStream_t)stream, max_steps, tokens_dev, durations_dev, frames_dev, n_steps_dev);
}

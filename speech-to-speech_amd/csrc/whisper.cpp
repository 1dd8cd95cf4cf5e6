// Whisper encoder / greedy decoder as a native kernel graph (C++ host, HIP kernels).
//
// Replaces, for one batch of sessions, what the reference obtains from
//   model.get_encoder()(input_features), model.detect_language(encoder_outputs) and
//   model.generate(input_features, max_new_tokens=128, num_beams=1, return_timestamps=False,
//                  task="transcribe", language=...)
// (reference STT/whisper_stt_handler.py:166-197,243; arithmetic restated from
//  transformers/models/whisper/modeling_whisper.py and generation_whisper.py:1455-1608,
//  1774-1812, 1610-1675).
//
// Layout in HBM (all per batch of B sessions, T = 1500 encoder frames, d = d_model):
//   feats   [B][3002][n_mels]  time-major log-mel with one zero row either side (conv padding)
//   y1      [B][3002][d]       conv1+GELU output, same padding convention -> conv2 reads it as
//                              overlapping rows (no im2col)
//   x       [B*T][d] f32       residual stream;  h/qkv/att/ffn scratch in the model dtype
//   cross   [L][B*T][2d]       cross-attention K|V, projected once per utterance
//   self    [L][B][Tmax][d]x2  decoder self-attention cache
#include "ops.h"

#include <string.h>

#include <map>
#include <string>
#include <vector>

namespace s2s {

constexpr int ENC_T = 1500;
constexpr int FEAT_ROWS = 3002;

struct Tensor {
    const void* ptr;
    long numel;
};

constexpr int WH_SS_PARTS = 512;      // pair partials per row of a residual projection (2 x d / 16; 160 at d = 1280)
constexpr int WH_SS_STRIDE = 128;     // row stride of the partial arrays (= the most sessions of a decode batch)
constexpr int WH_SLOT_PMAX = 8;       // longest forced decoder prefix of a slot (<sot><lang><task><notimestamps> = 4)

struct Whisper {
    s2s_whisper_config cfg;
    int max_batch;
    int esz;   // bytes per model-dtype element
    std::map<std::string, Tensor> w;
    bool finalized = false;
    // workspace
    char* arena = nullptr;
    size_t arena_bytes = 0;
    void *y1, *h, *qkv, *vt, *att, *ffn, *enc_out, *cross, *self_k, *self_v;
    float *x, *xd, *logits;
    void *hd, *qkvd, *attd, *ffnd, *qd;
    int *cur_ids, *pos, *step, *out_ids, *out_len, *finished, *picked;
    int tmax;
    int* host_flag = nullptr;   // pinned, max_batch ints
    // greedy-decode step as a replayable graph: the step counter, the call's scalars {P, max_new, eos}, the prompt and the
    // three vocabulary masks live in engine-owned device buffers, so one captured step body per batch size serves every call
    int *step_dev = nullptr, *params_dev = nullptr, *prompt_buf = nullptr;
    unsigned char *mask_sup = nullptr, *mask_begin = nullptr, *mask_fill = nullptr;
    hipStream_t cap_stream = nullptr;
    std::map<int, hipGraphExec_t> step_graphs;
    // bf16 decode path on the fragment-major weight-streaming kernel (gemm_dec.hip): own pre-swizzled copies of the decoder matrices
    std::map<std::string, void*> fragw;
    void* dec_ws = nullptr;       // K-slice workspace of the decode projections (gemm_dec2.hip)
    // LayerNorm-free decoder blocks (gemm_dec.hip LayerNorm prologue): the projections behind ln1 / ln2 / ln3 with gamma folded into
    // their columns ("<name>.lf"), beta into their bias ("<name>.lb") and their column sums ("<name>.cs"); pair partials of the residual
    // rows ping-pong between ssA / ssB [WH_SS_PARTS][WH_SS_STRIDE]
    bool ln_fold = false;
    float *ssA = nullptr, *ssB = nullptr;
    // continuous batching of the token loop (s2s_whisper_slots_*): the rows of ONE running greedy loop are slots which a session enters
    // after its encoder pass (s2s_whisper_encode_at wrote its cross K|V into the slot's rows) and leaves at its eos / token budget.
    // Per-slot scalars [max_batch] each: step (index of the token just consumed), P, max_new, active; prompts and model-filled prompt
    // tokens [max_batch][WH_SLOT_PMAX]; generated ids [max_batch][tmax]; the loop's own masks and eos.  cur_ids / pos / out_len /
    // finished are the arrays of the batch API: an engine runs in ONE of the two modes at a time.
    int *slot_step = nullptr, *slot_P = nullptr, *slot_maxnew = nullptr, *slot_active = nullptr, *slot_prompt = nullptr, *slot_fill = nullptr,
        *slot_out = nullptr;
    unsigned char* smask = nullptr;     // [3][vocab]: suppress, begin-suppress, prompt-fill
    int slot_eos = -1;
    float* cand_v = nullptr;            // [max_batch][ARGMAX_SLICES] slice candidates of the vocabulary argmax (value, index) and the per-row tickets
    int* cand_i = nullptr;
    unsigned* pick_ticket = nullptr;
    int slots_live = 0;                 // host-side count of reserved + admitted slots (always == number of non-zero bytes of slot_live_map)
    std::vector<unsigned char> slot_live_map;   // per slot: 0 idle, 1 reserved (s2s_whisper_encode_at wrote its cross K|V), 2 admitted to the token loop
                                                // - the scheduler's contract is checked, not trusted
    std::map<int, hipGraphExec_t> slot_graphs;
    std::vector<void*> owned;
};

static const Tensor* find(const Whisper* m, const std::string& name) {
    auto it = m->w.find(name);
    return it == m->w.end() ? nullptr : &it->second;
}

#define W_GET(var, name, n)                                                                      \
    const Tensor* var##_t = find(m, name);                                                       \
    if (!var##_t) return fail(S2S_ERR_STATE, "whisper: missing tensor %s", std::string(name).c_str()); \
    if (var##_t->numel != (long)(n))                                                             \
        return fail(S2S_ERR_STATE, "whisper: tensor %s has %ld elements, expected %ld",          \
                    std::string(name).c_str(), var##_t->numel, (long)(n));                       \
    const void* var = var##_t->ptr

static std::string lname(const char* side, int l, const char* rest) {
    return std::string(side) + "." + std::to_string(l) + "." + rest;
}

static int check_weights(const Whisper* m) {
    const auto& c = m->cfg;
    const long d = c.d_model;
    { W_GET(t, "enc.conv1.w", d * 3 * c.n_mels); (void)t; }
    { W_GET(t, "enc.conv1.b", d); (void)t; }
    { W_GET(t, "enc.conv2.w", d * 3 * d); (void)t; }
    { W_GET(t, "enc.conv2.b", d); (void)t; }
    { W_GET(t, "enc.pos", (long)ENC_T * d); (void)t; }
    { W_GET(t, "enc.ln_post.g", d); (void)t; }
    { W_GET(t, "enc.ln_post.b", d); (void)t; }
    for (int l = 0; l < c.enc_layers; ++l) {
        const char* names[] = {"ln1.g", "ln1.b", "qkv.b", "o.b", "ln2.g", "ln2.b", "fc2.b"};
        for (const char* nm : names) { W_GET(t, lname("enc", l, nm), std::string(nm) == "qkv.b" ? 3 * d : d); (void)t; }
        { W_GET(t, lname("enc", l, "qkv.w"), 3 * d * d); (void)t; }
        { W_GET(t, lname("enc", l, "o.w"), d * d); (void)t; }
        { W_GET(t, lname("enc", l, "fc1.w"), (long)c.ffn * d); (void)t; }
        { W_GET(t, lname("enc", l, "fc1.b"), c.ffn); (void)t; }
        { W_GET(t, lname("enc", l, "fc2.w"), (long)c.ffn * d); (void)t; }
    }
    { W_GET(t, "dec.tok", (long)c.vocab * d); (void)t; }
    { W_GET(t, "dec.pos", (long)c.max_target * d); (void)t; }
    { W_GET(t, "dec.ln.g", d); (void)t; }
    { W_GET(t, "dec.ln.b", d); (void)t; }
    for (int l = 0; l < c.dec_layers; ++l) {
        const char* dn[] = {"ln1.g", "ln1.b", "o.b", "ln2.g", "ln2.b", "cq.b", "co.b", "ln3.g", "ln3.b", "fc2.b"};
        for (const char* nm : dn) { W_GET(t, lname("dec", l, nm), d); (void)t; }
        { W_GET(t, lname("dec", l, "qkv.w"), 3 * d * d); (void)t; }
        { W_GET(t, lname("dec", l, "qkv.b"), 3 * d); (void)t; }
        { W_GET(t, lname("dec", l, "o.w"), d * d); (void)t; }
        { W_GET(t, lname("dec", l, "cq.w"), d * d); (void)t; }
        { W_GET(t, lname("dec", l, "ckv.w"), 2 * d * d); (void)t; }
        { W_GET(t, lname("dec", l, "ckv.b"), 2 * d); (void)t; }
        { W_GET(t, lname("dec", l, "co.w"), d * d); (void)t; }
        { W_GET(t, lname("dec", l, "fc1.w"), (long)c.ffn * d); (void)t; }
        { W_GET(t, lname("dec", l, "fc1.b"), c.ffn); (void)t; }
        { W_GET(t, lname("dec", l, "fc2.w"), (long)c.ffn * d); (void)t; }
    }
    return S2S_OK;
}

static const void* P(const Whisper* m, const std::string& name) { return find(m, name)->ptr; }
static const float* PF(const Whisper* m, const std::string& name) { return (const float*)find(m, name)->ptr; }

static int finalize(Whisper* m) {
    int rc = check_weights(m);
    if (rc) return rc;
    const auto& c = m->cfg;
    const size_t B = m->max_batch, d = c.d_model, e = m->esz, T = ENC_T;
    m->tmax = c.max_target;
    const size_t Tp = (T + 63) / 64 * 64;
    ArenaPlan a;
    const size_t o_y1 = a.take(B * FEAT_ROWS * d * e);
    const size_t o_x = a.take(B * T * d * 4);
    const size_t o_h = a.take(B * T * d * e);
    const size_t o_qkv = a.take(B * T * 3 * d * e);
    const size_t o_vt = a.take(B * d * Tp * e);
    const size_t o_att = a.take(B * T * d * e);
    const size_t o_ffn = a.take(B * T * (size_t)c.ffn * e);
    const size_t o_enc = a.take(B * T * d * e);
    const size_t o_cross = a.take((size_t)c.dec_layers * B * T * 2 * d * e);
    const size_t o_sk = a.take((size_t)c.dec_layers * B * m->tmax * d * e);
    const size_t o_sv = a.take((size_t)c.dec_layers * B * m->tmax * d * e);
    const size_t B16 = (B + 15) / 16 * 16;            // fragment-major decode activations are addressed in whole 16-row tiles
    const size_t o_xd = a.take(B * d * 4);
    const size_t o_hd = a.take(B16 * d * e);
    const size_t o_qkvd = a.take(B * 3 * d * e);
    const size_t o_attd = a.take(B16 * d * e);
    const size_t o_qd = a.take(B * d * e);
    const size_t o_ffnd = a.take(B16 * (size_t)c.ffn * e);
    const size_t o_logits = a.take(B * (size_t)c.vocab * 4);
    const size_t o_ints = a.take((size_t)(B * (8 + m->tmax)) * 4);
    m->arena_bytes = a.off;
    if ((rc = a.commit(&m->arena, "whisper"))) return rc;
    char* base = m->arena;
    m->y1 = base + o_y1; m->x = (float*)(base + o_x); m->h = base + o_h; m->qkv = base + o_qkv;
    m->vt = base + o_vt; m->att = base + o_att; m->ffn = base + o_ffn; m->enc_out = base + o_enc;
    m->cross = base + o_cross; m->self_k = base + o_sk; m->self_v = base + o_sv;
    m->xd = (float*)(base + o_xd); m->hd = base + o_hd; m->qkvd = base + o_qkvd; m->attd = base + o_attd;
    m->qd = base + o_qd; m->ffnd = base + o_ffnd; m->logits = (float*)(base + o_logits);
    int* ints = (int*)(base + o_ints);
    m->cur_ids = ints; m->pos = ints + B; m->step = ints + 2 * B; m->out_len = ints + 3 * B;
    m->finished = ints + 4 * B; m->picked = ints + 5 * B; m->out_ids = ints + 7 * B;
    S2S_HIP_CHECK(hipHostMalloc((void**)&m->host_flag, sizeof(int) * B, 0));
    {
        int* ib = nullptr;
        S2S_HIP_CHECK(hipMalloc((void**)&ib, sizeof(int) * (8 + (size_t)B * m->tmax)));
        S2S_HIP_CHECK(hipMemset(ib, 0, sizeof(int) * (8 + (size_t)B * m->tmax)));
        m->step_dev = ib; m->params_dev = ib + 4; m->prompt_buf = ib + 8;
        unsigned char* mb = nullptr;
        S2S_HIP_CHECK(hipMalloc((void**)&mb, 3 * (size_t)c.vocab));
        S2S_HIP_CHECK(hipMemset(mb, 0, 3 * (size_t)c.vocab));
        m->mask_sup = mb; m->mask_begin = mb + c.vocab; m->mask_fill = mb + 2 * (size_t)c.vocab;
        char* cb = nullptr;
        const size_t nc = (size_t)B * ARGMAX_SLICES;
        S2S_HIP_CHECK(hipMalloc((void**)&cb, nc * 8 + B * 4));
        S2S_HIP_CHECK(hipMemset(cb, 0, nc * 8 + B * 4));
        m->cand_v = (float*)cb; m->cand_i = (int*)(cb + nc * 4); m->pick_ticket = (unsigned*)(cb + nc * 8);
        m->owned.push_back(cb);
    }
    if (c.dtype == S2S_BF16 && getenv("S2S_NO_DEC_FUSION") == nullptr && d % 32 == 0 && c.ffn % 32 == 0 && d >= 256 && c.ffn >= 256 && c.dec_layers > 0) {
        // fragment-major copies ("<name>.f", s2s_op_to_frag) are made once by the host side and shared by every engine replica
        auto fragw = [&](const std::string& name, long rows, long K) -> bool {
            const Tensor* t = find(m, name + ".f");
            if (!t || t->numel != (rows + 15) / 16 * 16 * K) return false;
            m->fragw[name] = const_cast<void*>(t->ptr);
            return true;
        };
        bool all = true;
        for (int l = 0; l < c.dec_layers; ++l)
            all = all && fragw(lname("dec", l, "qkv.w"), 3 * d, d) && fragw(lname("dec", l, "o.w"), d, d) && fragw(lname("dec", l, "cq.w"), d, d) &&
                  fragw(lname("dec", l, "co.w"), d, d) && fragw(lname("dec", l, "fc1.w"), c.ffn, d) && fragw(lname("dec", l, "fc2.w"), d, c.ffn);
        all = all && fragw("dec.tok", c.vocab, d);
        if (!all) m->fragw.clear();
        else {
            S2S_HIP_CHECK(hipMalloc(&m->dec_ws, DEC_WS_BYTES));
            S2S_HIP_CHECK(hipMemset(m->dec_ws, 0, DEC_WS_BYTES));
            m->owned.push_back(m->dec_ws);
            // LayerNorm folded into the projections that follow it (made by the host side next to the ".f" copies)
            bool ln = getenv("S2S_NO_LN_FOLD") == nullptr && 2 * cdiv(d, 16) <= WH_SS_PARTS && B <= WH_SS_STRIDE;
            auto lnw = [&](const std::string& name, long rows) -> bool {
                const Tensor* w = find(m, name + ".w.lf");
                const Tensor* b = find(m, name + ".lb");
                const Tensor* s = find(m, name + ".cs");
                if (!w || !b || !s || w->numel != (rows + 15) / 16 * 16 * (long)d || b->numel != rows || s->numel != rows) return false;
                m->fragw[name + ".w.lf"] = const_cast<void*>(w->ptr);
                return true;
            };
            for (int l = 0; ln && l < c.dec_layers; ++l) ln = lnw(lname("dec", l, "qkv"), 3 * d) && lnw(lname("dec", l, "cq"), d) && lnw(lname("dec", l, "fc1"), c.ffn);
            if (ln) {
                S2S_HIP_CHECK(hipMalloc((void**)&m->ssA, sizeof(float) * WH_SS_PARTS * WH_SS_STRIDE));
                S2S_HIP_CHECK(hipMalloc((void**)&m->ssB, sizeof(float) * WH_SS_PARTS * WH_SS_STRIDE));
                S2S_HIP_CHECK(hipMemset(m->ssA, 0, sizeof(float) * WH_SS_PARTS * WH_SS_STRIDE));
                S2S_HIP_CHECK(hipMemset(m->ssB, 0, sizeof(float) * WH_SS_PARTS * WH_SS_STRIDE));
                m->owned.push_back(m->ssA);
                m->owned.push_back(m->ssB);
                m->ln_fold = true;
            }
        }
    }
    m->finalized = true;
    return S2S_OK;
}

static inline void* at(void* p, size_t bytes) { return (char*)p + bytes; }

static int encode(Whisper* m, hipStream_t st, const void* feats, int B, int slot0 = 0) {
    const auto& c = m->cfg;
    const int d = c.d_model, H = c.n_heads, D = d / H, dt = c.dtype, e = m->esz;
    const int M = B * ENC_T;
    int rc;
    // conv1 (k3 s1 p1) + GELU -> y1 rows 1..3000 ; conv2 (k3 s2 p1) + GELU + positions -> x
    GemmArgs g{};
    g.A = feats; g.lda = c.n_mels; g.a_bs = (long)FEAT_ROWS * c.n_mels;
    g.W = P(m, "enc.conv1.w"); g.bias = PF(m, "enc.conv1.b");
    g.C = at(m->y1, (size_t)d * e); g.ldc = d; g.c_bs = (long)FEAT_ROWS * d;
    g.M = B * 3000; g.N = d; g.K = 3 * c.n_mels; g.G = 3000;
    g.ab_dtype = dt; g.c_dtype = dt; g.act = S2S_ACT_GELU;
    if ((rc = gemm_ex(st, g))) return rc;
    GemmArgs g2{};
    g2.A = m->y1; g2.lda = 2L * d; g2.a_bs = (long)FEAT_ROWS * d;
    g2.W = P(m, "enc.conv2.w"); g2.bias = PF(m, "enc.conv2.b");
    g2.residual = PF(m, "enc.pos"); g2.ldr = d; g2.r_bs = 0;   // sinusoid table, broadcast over the batch
    g2.C = m->x; g2.ldc = d; g2.c_bs = (long)ENC_T * d;
    g2.M = M; g2.N = d; g2.K = 3 * d; g2.G = ENC_T;
    g2.ab_dtype = dt; g2.c_dtype = S2S_F32; g2.act = S2S_ACT_GELU;
    if ((rc = gemm_ex(st, g2))) return rc;

    const int Tp = (ENC_T + 63) / 64 * 64;
    const float scale = 1.0f / sqrtf((float)D);
    for (int l = 0; l < c.enc_layers; ++l) {
        if ((rc = layernorm(st, m->x, PF(m, lname("enc", l, "ln1.g")), PF(m, lname("enc", l, "ln1.b")), m->h, M, d, 1e-5f, dt))) return rc;
        if ((rc = gemm(st, m->h, d, P(m, lname("enc", l, "qkv.w")), PF(m, lname("enc", l, "qkv.b")), nullptr, m->qkv, 3 * d, M, 3 * d, d, dt, dt, S2S_ACT_NONE, false))) return rc;
        if (dt == S2S_BF16 && (D == 64 || D == 128)) {
            const bool direct_v = attn_mfma_direct_v() && d % 8 == 0;      // V rows straight from the qkv projection (round 6): no transpose pass
            if (!direct_v && (rc = transpose_v(st, at(m->qkv, (size_t)2 * d * e), (long)ENC_T * 3 * d, 3 * d, m->vt, B, H, D, ENC_T, Tp, dt))) return rc;
            AttnMfmaParams p{};
            p.Q = (const __hip_bfloat16*)m->qkv; p.q_bs = (long)ENC_T * 3 * d; p.q_rs = 3 * d;
            p.K = (const __hip_bfloat16*)at(m->qkv, (size_t)d * e); p.k_bs = p.q_bs; p.k_rs = 3 * d;
            p.Vt = direct_v ? nullptr : (const __hip_bfloat16*)m->vt;
            p.V = (const __hip_bfloat16*)at(m->qkv, (size_t)2 * d * e); p.v_bs = p.q_bs; p.v_rs = 3 * d;
            p.O = (__hip_bfloat16*)m->att; p.o_bs = (long)ENC_T * d; p.o_rs = d;
            p.Tk = ENC_T; p.Tp = Tp; p.Tq = ENC_T; p.H = H; p.B = B;
            p.scale_log2e = scale * 1.4426950408889634f;
            if ((rc = attn_mfma(st, p, D))) return rc;
        } else {
            AttnRowsParams p{};
            p.Q = m->qkv; p.q_bs = (long)ENC_T * 3 * d; p.q_rs = 3 * d;
            p.K = at(m->qkv, (size_t)d * e); p.V = at(m->qkv, (size_t)2 * d * e); p.kv_bs = p.q_bs; p.kv_rs = 3 * d;
            p.O = m->att; p.o_bs = (long)ENC_T * d; p.o_rs = d;
            p.kv_len_const = ENC_T; p.B = B; p.H = H; p.Hkv = H; p.Tq = ENC_T; p.D = D; p.scale = scale;
            if ((rc = attn_rows(st, p, dt))) return rc;
        }
        if ((rc = gemm(st, m->att, d, P(m, lname("enc", l, "o.w")), PF(m, lname("enc", l, "o.b")), m->x, m->x, d, M, d, d, dt, S2S_F32, S2S_ACT_NONE, false))) return rc;
        if ((rc = layernorm(st, m->x, PF(m, lname("enc", l, "ln2.g")), PF(m, lname("enc", l, "ln2.b")), m->h, M, d, 1e-5f, dt))) return rc;
        if ((rc = gemm(st, m->h, d, P(m, lname("enc", l, "fc1.w")), PF(m, lname("enc", l, "fc1.b")), nullptr, m->ffn, c.ffn, M, c.ffn, d, dt, dt, S2S_ACT_GELU, false))) return rc;
        if ((rc = gemm(st, m->ffn, c.ffn, P(m, lname("enc", l, "fc2.w")), PF(m, lname("enc", l, "fc2.b")), m->x, m->x, d, M, d, c.ffn, dt, S2S_F32, S2S_ACT_NONE, false))) return rc;
    }
    if ((rc = layernorm(st, m->x, PF(m, "enc.ln_post.g"), PF(m, "enc.ln_post.b"), m->enc_out, M, d, 1e-5f, dt))) return rc;
    // cross-attention K|V of every decoder layer, once per utterance
    for (int l = 0; l < c.dec_layers; ++l) {
        void* ckv = at(m->cross, ((size_t)l * m->max_batch + slot0) * ENC_T * 2 * d * e);      // rows of slots slot0 .. slot0 + B - 1
        if ((rc = gemm(st, m->enc_out, d, P(m, lname("dec", l, "ckv.w")), PF(m, lname("dec", l, "ckv.b")), nullptr, ckv, 2 * d, M, 2 * d, d, dt, dt, S2S_ACT_NONE, false))) return rc;
    }
    return S2S_OK;
}

// decode-step projection on the fragment-major weight-streaming kernel: C = act(A W^T + b) (bf16 / f32) or X += A W^T + b
static int dec_proj(Whisper* m, hipStream_t st, const void* A, const std::string& w, const float* bias, int B, int N, int K, int mode, void* C,
                    bool c_frag, int act, float* X, void* XB = nullptr, float* ss_out = nullptr, int* parts_out = nullptr) {
    DecGemm a{};
    a.A = (const __hip_bfloat16*)A; a.a_frag = true; a.W = (const __hip_bfloat16*)m->fragw.at(w); a.bias = bias;
    a.M = B; a.N = N; a.K = K; a.mode = mode; a.C = C; a.ldc = N; a.c_frag = c_frag; a.act = act; a.X = X; a.ldx = N;
    a.ws = m->dec_ws; a.ws_bytes = DEC_WS_BYTES;
    if (XB) {      // LayerNorm-free path: the residual projection also emits the bf16 copy of the new rows and their pair partials
        a.XB = (__hip_bfloat16*)XB; a.ldxb = N; a.xb_frag = true; a.ss_out = ss_out; a.ss_stride = WH_SS_STRIDE; a.ss_pairs = true;
        if (parts_out) *parts_out = dec_out_parts(a);
    }
    return gemm_dec(st, a);
}

// projection behind a LayerNorm folded into it: out = rstd * (xb W'^T - mean * csum) + b'   (gemm_dec.hip LayerNorm prologue)
static int dec_proj_ln(Whisper* m, hipStream_t st, const void* xb, const std::string& name, int B, int N, int K, void* C, bool c_frag, int act, const float* ss,
                       int parts) {
    DecGemm a{};
    a.A = (const __hip_bfloat16*)xb; a.a_frag = true; a.W = (const __hip_bfloat16*)m->fragw.at(name + ".w.lf"); a.bias = PF(m, name + ".lb");
    a.M = B; a.N = N; a.K = K; a.mode = 0; a.C = C; a.ldc = N; a.c_frag = c_frag; a.act = act;
    a.ln = true; a.csum = PF(m, name + ".cs"); a.ss_in = ss; a.n_parts_in = parts; a.ss_stride = WH_SS_STRIDE; a.inv_dim = 1.0f / (float)K; a.eps = 1e-5f;
    a.ws = m->dec_ws; a.ws_bytes = DEC_WS_BYTES;
    return gemm_dec(st, a);
}

// one decoder step without LayerNorm launches: 8 launches per block (qkv, cache append + self attention, out-proj, cross-q, cross
// attention, out-proj, fc1, fc2) instead of 12; the f32 residual rows xd keep a bf16 fragment-major copy (hd) and pair partials
static int decode_step_dec_ln(Whisper* m, hipStream_t st, int B) {
    const auto& c = m->cfg;
    const int d = c.d_model, H = c.n_heads, D = d / H, dt = c.dtype, e = m->esz;
    const float scale = 1.0f / sqrtf((float)D);
    int rc;
    if ((rc = embed(st, m->cur_ids, m->pos, P(m, "dec.tok"), P(m, "dec.pos"), m->xd, B, d, 1.0f, dt))) return rc;
    if ((rc = rows_prep(st, m->xd, d, m->hd, d, m->ssA, B, d, true, WH_SS_STRIDE))) return rc;
    float *cur = m->ssA, *nxt = m->ssB;
    int parts = 2;
    for (int l = 0; l < c.dec_layers; ++l) {
        void* sk = at(m->self_k, (size_t)l * m->max_batch * m->tmax * d * e);
        void* sv = at(m->self_v, (size_t)l * m->max_batch * m->tmax * d * e);
        void* ckv = at(m->cross, (size_t)l * m->max_batch * ENC_T * 2 * d * e);
        if ((rc = dec_proj_ln(m, st, m->hd, lname("dec", l, "qkv"), B, 3 * d, d, m->qkvd, false, S2S_ACT_NONE, cur, parts))) return rc;
        {   // self attention: cache append + attention over keys 0 .. pos[b] in ONE launch (no head norms, no rotary: Whisper)
            AttnStepParams a{};
            a.qkv = m->qkvd; a.qkv_rs = 3 * d;
            a.K = sk; a.V = sv; a.kv_bs = (long)m->tmax * d; a.kv_rs = d;
            a.O = m->attd; a.o_bs = d; a.o_frag = true;
            a.pos = m->pos; a.B = B; a.H = H; a.Hkv = H; a.D = D; a.kv_cap = m->tmax; a.window = 0; a.scale = scale; a.eps = 0.f;
            if ((rc = attn_step(st, a, dt))) return rc;
        }
        if ((rc = dec_proj(m, st, m->attd, lname("dec", l, "o.w"), PF(m, lname("dec", l, "o.b")), B, d, d, 2, nullptr, false, S2S_ACT_NONE, m->xd, m->hd, nxt, &parts))) return rc;
        std::swap(cur, nxt);
        if ((rc = dec_proj_ln(m, st, m->hd, lname("dec", l, "cq"), B, d, d, m->qd, false, S2S_ACT_NONE, cur, parts))) return rc;
        AttnRowsParams pc{};
        pc.Q = m->qd; pc.q_bs = d; pc.q_rs = d;
        pc.K = ckv; pc.V = at(ckv, (size_t)d * e); pc.kv_bs = (long)ENC_T * 2 * d; pc.kv_rs = 2 * d;
        pc.O = m->attd; pc.o_bs = d; pc.o_rs = d; pc.o_frag = true;
        pc.kv_len_const = ENC_T; pc.B = B; pc.H = H; pc.Hkv = H; pc.Tq = 1; pc.D = D; pc.scale = scale;
        if ((rc = attn_rows(st, pc, dt))) return rc;
        if ((rc = dec_proj(m, st, m->attd, lname("dec", l, "co.w"), PF(m, lname("dec", l, "co.b")), B, d, d, 2, nullptr, false, S2S_ACT_NONE, m->xd, m->hd, nxt, &parts))) return rc;
        std::swap(cur, nxt);
        if ((rc = dec_proj_ln(m, st, m->hd, lname("dec", l, "fc1"), B, c.ffn, d, m->ffnd, true, S2S_ACT_GELU, cur, parts))) return rc;
        if ((rc = dec_proj(m, st, m->ffnd, lname("dec", l, "fc2.w"), PF(m, lname("dec", l, "fc2.b")), B, d, c.ffn, 2, nullptr, false, S2S_ACT_NONE, m->xd, m->hd, nxt, &parts))) return rc;
        std::swap(cur, nxt);
    }
    // the final LayerNorm stays a launch: its consumer, the 51 866-column logits projection, runs the column-block kernel
    if ((rc = layernorm(st, m->xd, PF(m, "dec.ln.g"), PF(m, "dec.ln.b"), m->hd, B, d, 1e-5f, dt, true))) return rc;
    return dec_proj(m, st, m->hd, "dec.tok", nullptr, B, c.vocab, d, 1, m->logits, false, S2S_ACT_NONE, nullptr);
}

// one decoder step for B sessions on the bf16 serving path: LayerNorm rows, attention outputs and the GELU activations are
// written fragment-major by their producers, every projection streams pre-swizzled weights (gemm_dec.hip)
static int decode_step_dec(Whisper* m, hipStream_t st, int B) {
    const auto& c = m->cfg;
    const int d = c.d_model, H = c.n_heads, D = d / H, dt = c.dtype, e = m->esz;
    const float scale = 1.0f / sqrtf((float)D);
    int rc;
    if ((rc = embed(st, m->cur_ids, m->pos, P(m, "dec.tok"), P(m, "dec.pos"), m->xd, B, d, 1.0f, dt))) return rc;
    for (int l = 0; l < c.dec_layers; ++l) {
        void* sk = at(m->self_k, (size_t)l * m->max_batch * m->tmax * d * e);
        void* sv = at(m->self_v, (size_t)l * m->max_batch * m->tmax * d * e);
        void* ckv = at(m->cross, (size_t)l * m->max_batch * ENC_T * 2 * d * e);
        if ((rc = layernorm(st, m->xd, PF(m, lname("dec", l, "ln1.g")), PF(m, lname("dec", l, "ln1.b")), m->hd, B, d, 1e-5f, dt, true))) return rc;
        if ((rc = dec_proj(m, st, m->hd, lname("dec", l, "qkv.w"), PF(m, lname("dec", l, "qkv.b")), B, 3 * d, d, 0, m->qkvd, false, S2S_ACT_NONE, nullptr))) return rc;
        if ((rc = kv_append(st, m->qkvd, 3 * d, d, 2 * d, sk, sv, (long)m->tmax * d, m->pos, B, d, dt))) return rc;
        AttnRowsParams p{};
        p.Q = m->qkvd; p.q_bs = 3 * d; p.q_rs = 3 * d;
        p.K = sk; p.V = sv; p.kv_bs = (long)m->tmax * d; p.kv_rs = d;
        p.O = m->attd; p.o_bs = d; p.o_rs = d; p.o_frag = true;
        p.kv_len_const = m->tmax; p.causal = 1; p.causal_off_arr = m->pos;
        p.B = B; p.H = H; p.Hkv = H; p.Tq = 1; p.D = D; p.scale = scale;
        if ((rc = attn_rows(st, p, dt))) return rc;
        if ((rc = dec_proj(m, st, m->attd, lname("dec", l, "o.w"), PF(m, lname("dec", l, "o.b")), B, d, d, 2, nullptr, false, S2S_ACT_NONE, m->xd))) return rc;
        if ((rc = layernorm(st, m->xd, PF(m, lname("dec", l, "ln2.g")), PF(m, lname("dec", l, "ln2.b")), m->hd, B, d, 1e-5f, dt, true))) return rc;
        if ((rc = dec_proj(m, st, m->hd, lname("dec", l, "cq.w"), PF(m, lname("dec", l, "cq.b")), B, d, d, 0, m->qd, false, S2S_ACT_NONE, nullptr))) return rc;
        AttnRowsParams pc{};
        pc.Q = m->qd; pc.q_bs = d; pc.q_rs = d;
        pc.K = ckv; pc.V = at(ckv, (size_t)d * e); pc.kv_bs = (long)ENC_T * 2 * d; pc.kv_rs = 2 * d;
        pc.O = m->attd; pc.o_bs = d; pc.o_rs = d; pc.o_frag = true;
        pc.kv_len_const = ENC_T; pc.B = B; pc.H = H; pc.Hkv = H; pc.Tq = 1; pc.D = D; pc.scale = scale;
        if ((rc = attn_rows(st, pc, dt))) return rc;
        if ((rc = dec_proj(m, st, m->attd, lname("dec", l, "co.w"), PF(m, lname("dec", l, "co.b")), B, d, d, 2, nullptr, false, S2S_ACT_NONE, m->xd))) return rc;
        if ((rc = layernorm(st, m->xd, PF(m, lname("dec", l, "ln3.g")), PF(m, lname("dec", l, "ln3.b")), m->hd, B, d, 1e-5f, dt, true))) return rc;
        if ((rc = dec_proj(m, st, m->hd, lname("dec", l, "fc1.w"), PF(m, lname("dec", l, "fc1.b")), B, c.ffn, d, 0, m->ffnd, true, S2S_ACT_GELU, nullptr))) return rc;
        if ((rc = dec_proj(m, st, m->ffnd, lname("dec", l, "fc2.w"), PF(m, lname("dec", l, "fc2.b")), B, d, c.ffn, 2, nullptr, false, S2S_ACT_NONE, m->xd))) return rc;
    }
    if ((rc = layernorm(st, m->xd, PF(m, "dec.ln.g"), PF(m, "dec.ln.b"), m->hd, B, d, 1e-5f, dt, true))) return rc;
    // logits = h @ tok_emb^T (tied embedding)
    return dec_proj(m, st, m->hd, "dec.tok", nullptr, B, c.vocab, d, 1, m->logits, false, S2S_ACT_NONE, nullptr);
}

// one decoder step for B sessions: input ids in m->cur_ids, positions in m->pos
static int decode_step(Whisper* m, hipStream_t st, int B, bool split_ok) {
    if (!m->fragw.empty() && B <= 128 && m->tmax >= 64) return m->ln_fold ? decode_step_dec_ln(m, st, B) : decode_step_dec(m, st, B);
    const auto& c = m->cfg;
    const int d = c.d_model, H = c.n_heads, D = d / H, dt = c.dtype, e = m->esz;
    const float scale = 1.0f / sqrtf((float)D);
    int rc;
    if ((rc = embed(st, m->cur_ids, m->pos, P(m, "dec.tok"), P(m, "dec.pos"), m->xd, B, d, 1.0f, dt))) return rc;
    for (int l = 0; l < c.dec_layers; ++l) {
        void* sk = at(m->self_k, (size_t)l * m->max_batch * m->tmax * d * e);
        void* sv = at(m->self_v, (size_t)l * m->max_batch * m->tmax * d * e);
        void* ckv = at(m->cross, (size_t)l * m->max_batch * ENC_T * 2 * d * e);
        if ((rc = layernorm(st, m->xd, PF(m, lname("dec", l, "ln1.g")), PF(m, lname("dec", l, "ln1.b")), m->hd, B, d, 1e-5f, dt))) return rc;
        if ((rc = gemm(st, m->hd, d, P(m, lname("dec", l, "qkv.w")), PF(m, lname("dec", l, "qkv.b")), nullptr, m->qkvd, 3 * d, B, 3 * d, d, dt, dt, S2S_ACT_NONE, false))) return rc;
        if ((rc = kv_append(st, m->qkvd, 3 * d, d, 2 * d, sk, sv, (long)m->tmax * d, m->pos, B, d, dt))) return rc;
        AttnRowsParams p{};
        p.Q = m->qkvd; p.q_bs = 3 * d; p.q_rs = 3 * d;
        p.K = sk; p.V = sv; p.kv_bs = (long)m->tmax * d; p.kv_rs = d;
        p.O = m->attd; p.o_bs = d; p.o_rs = d;
        // keys 0 .. pos[b] (the row just appended): constant capacity, per-session causal bound read on the device
        p.kv_len_const = m->tmax; p.causal = 1; p.causal_off_arr = m->pos;
        p.B = B; p.H = H; p.Hkv = H; p.Tq = 1; p.D = D; p.scale = scale;
        if ((rc = attn_rows(st, p, dt))) return rc;
        if ((rc = gemm(st, m->attd, d, P(m, lname("dec", l, "o.w")), PF(m, lname("dec", l, "o.b")), m->xd, m->xd, d, B, d, d, dt, S2S_F32, S2S_ACT_NONE, split_ok))) return rc;
        // cross attention
        if ((rc = layernorm(st, m->xd, PF(m, lname("dec", l, "ln2.g")), PF(m, lname("dec", l, "ln2.b")), m->hd, B, d, 1e-5f, dt))) return rc;
        if ((rc = gemm(st, m->hd, d, P(m, lname("dec", l, "cq.w")), PF(m, lname("dec", l, "cq.b")), nullptr, m->qd, d, B, d, d, dt, dt, S2S_ACT_NONE, false))) return rc;
        AttnRowsParams pc{};
        pc.Q = m->qd; pc.q_bs = d; pc.q_rs = d;
        pc.K = ckv; pc.V = at(ckv, (size_t)d * e); pc.kv_bs = (long)ENC_T * 2 * d; pc.kv_rs = 2 * d;
        pc.O = m->attd; pc.o_bs = d; pc.o_rs = d;
        pc.kv_len_const = ENC_T; pc.B = B; pc.H = H; pc.Hkv = H; pc.Tq = 1; pc.D = D; pc.scale = scale;
        if ((rc = attn_rows(st, pc, dt))) return rc;
        if ((rc = gemm(st, m->attd, d, P(m, lname("dec", l, "co.w")), PF(m, lname("dec", l, "co.b")), m->xd, m->xd, d, B, d, d, dt, S2S_F32, S2S_ACT_NONE, split_ok))) return rc;
        // feed-forward
        if ((rc = layernorm(st, m->xd, PF(m, lname("dec", l, "ln3.g")), PF(m, lname("dec", l, "ln3.b")), m->hd, B, d, 1e-5f, dt))) return rc;
        if ((rc = gemm(st, m->hd, d, P(m, lname("dec", l, "fc1.w")), PF(m, lname("dec", l, "fc1.b")), nullptr, m->ffnd, c.ffn, B, c.ffn, d, dt, dt, S2S_ACT_GELU, false))) return rc;
        if ((rc = gemm(st, m->ffnd, c.ffn, P(m, lname("dec", l, "fc2.w")), PF(m, lname("dec", l, "fc2.b")), m->xd, m->xd, d, B, d, c.ffn, dt, S2S_F32, S2S_ACT_NONE, split_ok))) return rc;
    }
    if ((rc = layernorm(st, m->xd, PF(m, "dec.ln.g"), PF(m, "dec.ln.b"), m->hd, B, d, 1e-5f, dt))) return rc;
    // logits = h @ tok_emb^T (tied embedding)
    if ((rc = gemm(st, m->hd, d, P(m, "dec.tok"), nullptr, nullptr, m->logits, c.vocab, B, c.vocab, d, dt, S2S_F32, S2S_ACT_NONE, false))) return rc;
    return S2S_OK;
}

// End of a step, one block per session: greedy pick over the vocabulary (while the prompt is being consumed the
// prompt-fill mask applies - language detection placeholders -, afterwards the suppress mask, plus the begin-suppress
// mask on the first generated position), then route the pick (or the forced prompt token) to the next step's input,
// record generated tokens and track EOS.  The step index and the call's scalars are read from device memory.
__global__ __launch_bounds__(256) void whisper_pick_advance_kernel(const float* __restrict__ logits, int V,
                                                                   const unsigned char* __restrict__ mask_sup,
                                                                   const unsigned char* __restrict__ mask_begin,
                                                                   const unsigned char* __restrict__ mask_fill,
                                                                   const int* __restrict__ step_dev, const int* __restrict__ params,
                                                                   const int* __restrict__ prompt, int* cur_ids, int* pos,
                                                                   int* out_ids, int* out_len, int* finished, float* cand_v, int* cand_i,
                                                                   unsigned* ticket) {
    const int b = blockIdx.x;
    const int step = *step_dev, P = params[0], max_new = params[1], eos = params[2];
    const bool in_prompt = step + 1 < P;
    const int picked = sliced_argmax_256(logits + (long)b * V, V, in_prompt ? mask_fill : mask_sup,
                                         (!in_prompt && step == P - 1) ? mask_begin : nullptr, b, cand_v, cand_i, ticket);
    if (threadIdx.x != 0 || picked < 0) return;      // (only the last arriver of the row's slices advances it)
    // `step` is the index of the token just consumed; the next input sits at position step+1
    int next;
    if (in_prompt) {
        next = prompt[b * P + step + 1];
        if (next < 0) next = picked;   // placeholder: take the (masked) argmax, e.g. the language token
        if (next != prompt[b * P + step + 1]) out_ids[b * (max_new + P) + max_new + step + 1] = next;  // report the filled-in prompt token
    } else {
        next = picked;
        const int n = out_len[b];
        if (!finished[b]) {
            if (next == eos) {
                finished[b] = 1;
            } else if (n < max_new) {
                out_ids[b * (max_new + P) + n] = next;
                out_len[b] = n + 1;
            }
        }
        if (finished[b]) next = eos;
    }
    cur_ids[b] = next;
    pos[b] = step + 1;
}

__global__ void whisper_step_inc_kernel(int* step_dev) { *step_dev += 1; }

__global__ void whisper_set_params_kernel(int* params, int P, int max_new, int eos, int* step_dev) {
    params[0] = P; params[1] = max_new; params[2] = eos; *step_dev = 0;
}

// one full greedy step: decoder pass, pick, advance, step counter.  Shape-static -> capturable.
static int step_body(Whisper* m, hipStream_t st, int B, bool split_ok) {
    int rc = decode_step(m, st, B, split_ok);
    if (rc) return rc;
    whisper_pick_advance_kernel<<<dim3(B, ARGMAX_SLICES), 256, 0, st>>>(m->logits, m->cfg.vocab, m->mask_sup, m->mask_begin, m->mask_fill, m->step_dev,
                                                                        m->params_dev, m->prompt_buf, m->cur_ids, m->pos, m->out_ids, m->out_len, m->finished,
                                                                        m->cand_v, m->cand_i, m->pick_ticket);
    S2S_LAUNCH_CHECK();
    whisper_step_inc_kernel<<<1, 1, 0, st>>>(m->step_dev);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

static int step_graph(Whisper* m, int B, bool split_ok, hipGraphExec_t* out) {
    auto it = m->step_graphs.find(B);
    if (it != m->step_graphs.end()) {
        *out = it->second;
        return S2S_OK;
    }
    std::lock_guard<std::mutex> capture_lock(capture_mutex());
    if (!m->cap_stream) S2S_HIP_CHECK(hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
    S2S_HIP_CHECK(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = step_body(m, m->cap_stream, B, split_ok);
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
    m->step_graphs[B] = exec;
    *out = exec;
    return S2S_OK;
}

// ---- continuous batching of the token loop ------------------------------------------------------------------------------------
// End of a step in slot mode, one block per row: as whisper_pick_advance_kernel, but every row reads ITS step / prefix length / budget;
// rows that are idle, or whose session has finished (eos or budget), stand still (their decoder pass re-writes the same cache row).
__global__ __launch_bounds__(256) void whisper_slot_pick_kernel(const float* __restrict__ logits, int V, const unsigned char* __restrict__ mask_sup,
                                                                const unsigned char* __restrict__ mask_begin,
                                                                const unsigned char* __restrict__ mask_fill, int eos, int tmax, int* step_a,
                                                                const int* __restrict__ P_a, const int* __restrict__ maxnew_a,
                                                                const int* __restrict__ active_a, const int* __restrict__ prompt, int* fill,
                                                                int* cur_ids, int* pos, int* out, int* out_len, int* finished, float* cand_v, int* cand_i,
                                                                unsigned* ticket) {
    const int b = blockIdx.x;
    if (!active_a[b] || finished[b]) return;      // (uniform over the row's slices: the row's state only changes after ALL of them took their ticket)
    const int step = step_a[b], P = P_a[b], max_new = maxnew_a[b];
    const bool in_prompt = step + 1 < P;
    const int picked = sliced_argmax_256(logits + (long)b * V, V, in_prompt ? mask_fill : mask_sup, (!in_prompt && step == P - 1) ? mask_begin : nullptr, b,
                                         cand_v, cand_i, ticket);
    if (threadIdx.x != 0 || picked < 0) return;
    int next;
    if (in_prompt) {
        next = prompt[b * WH_SLOT_PMAX + step + 1];
        if (next < 0) {              // placeholder: the (masked) argmax, e.g. the language token; reported back through `fill`
            next = picked;
            fill[b * WH_SLOT_PMAX + step + 1] = next;
        }
    } else {
        next = picked;
        const int n = out_len[b];
        if (next == eos) {
            finished[b] = 1;
        } else {
            out[(long)b * tmax + n] = next;
            out_len[b] = n + 1;
            if (n + 1 >= max_new) finished[b] = 2;      // budget: exactly max_new picks, like the batch loop's P - 1 + max_new steps
        }
        if (finished[b]) return;     // the row stands still from here on
    }
    cur_ids[b] = next;
    pos[b] = step + 1;
    step_a[b] = step + 1;
}

__global__ void whisper_slot_admit_kernel(int slot0, int n, const int* __restrict__ prompts, int P, const int* __restrict__ maxnew, int maxnew_limit, int* step_a, int* P_a,
                                          int* maxnew_a, int* active_a, int* prompt, int* fill, int* cur_ids, int* pos, int* out_len, int* finished) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = slot0 + i;
    for (int k = 0; k < WH_SLOT_PMAX; ++k) {
        prompt[b * WH_SLOT_PMAX + k] = k < P ? prompts[i * P + k] : 0;
        fill[b * WH_SLOT_PMAX + k] = 0;
    }
    const int mn = maxnew[i];
    step_a[b] = 0; P_a[b] = P; maxnew_a[b] = mn < 1 ? 1 : (mn > maxnew_limit ? maxnew_limit : mn); active_a[b] = 1;   // the row's out buffer holds tmax ids: never past the checked limit
    cur_ids[b] = prompts[i * P]; pos[b] = 0; out_len[b] = 0; finished[b] = 0;
}

__global__ void whisper_slot_release_kernel(int slot0, int n, int* active_a, int* finished, int* pos, int* cur_ids) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    active_a[slot0 + i] = 0; finished[slot0 + i] = 0; pos[slot0 + i] = 0; cur_ids[slot0 + i] = 0;
}

// rows [0, rows): pack[b] = {finished, out_len, fill[0 .. PMAX), ids[0 .. tmax)} - ids / fill only for rows that have finished
__global__ __launch_bounds__(256) void whisper_slot_collect_kernel(const int* __restrict__ active_a, const int* __restrict__ finished,
                                                                   const int* __restrict__ out_len, const int* __restrict__ fill,
                                                                   const int* __restrict__ out, int tmax, int* pack) {
    const int b = blockIdx.x, stride = 2 + WH_SLOT_PMAX + tmax;
    int* row = pack + (long)b * stride;
    const int fin = active_a[b] ? finished[b] : 0, n = out_len[b];
    if (threadIdx.x == 0) { row[0] = fin; row[1] = n; }
    if (!fin) return;
    for (int k = threadIdx.x; k < WH_SLOT_PMAX; k += blockDim.x) row[2 + k] = fill[b * WH_SLOT_PMAX + k];
    for (int k = threadIdx.x; k < n; k += blockDim.x) row[2 + WH_SLOT_PMAX + k] = out[(long)b * tmax + k];
}

static int slot_step_body(Whisper* m, hipStream_t st, int rows, bool split_ok) {
    int rc = decode_step(m, st, rows, split_ok);
    if (rc) return rc;
    const int V = m->cfg.vocab;
    whisper_slot_pick_kernel<<<dim3(rows, ARGMAX_SLICES), 256, 0, st>>>(m->logits, V, m->smask, m->smask + V, m->smask + 2 * (size_t)V, m->slot_eos, m->tmax,
                                                                        m->slot_step, m->slot_P, m->slot_maxnew, m->slot_active, m->slot_prompt, m->slot_fill,
                                                                        m->cur_ids, m->pos, m->slot_out, m->out_len, m->finished, m->cand_v, m->cand_i,
                                                                        m->pick_ticket);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

// one captured step per row count (every per-slot quantity lives in device memory; eos is baked in: slots_config drops the graphs)
static int slot_graph(Whisper* m, int rows, bool split_ok, hipGraphExec_t* out) {
    auto it = m->slot_graphs.find(rows);
    if (it != m->slot_graphs.end()) {
        *out = it->second;
        return S2S_OK;
    }
    std::lock_guard<std::mutex> capture_lock(capture_mutex());
    if (!m->cap_stream) S2S_HIP_CHECK(hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
    S2S_HIP_CHECK(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = slot_step_body(m, m->cap_stream, rows, split_ok);
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
    m->slot_graphs[rows] = exec;
    *out = exec;
    return S2S_OK;
}

static int slots_alloc(Whisper* m) {
    if (m->slot_step) return S2S_OK;
    std::lock_guard<std::mutex> no_capture(capture_mutex());     // (synchronous allocation / memset: never under another thread's capture)
    const size_t B = m->max_batch, n_int = B * (4 + 2 * WH_SLOT_PMAX + (size_t)m->tmax);
    int* ib = nullptr;
    S2S_HIP_CHECK(hipMalloc((void**)&ib, sizeof(int) * n_int));
    S2S_HIP_CHECK(hipMemset(ib, 0, sizeof(int) * n_int));
    m->owned.push_back(ib);
    m->slot_step = ib; m->slot_P = ib + B; m->slot_maxnew = ib + 2 * B; m->slot_active = ib + 3 * B;
    m->slot_prompt = ib + 4 * B; m->slot_fill = m->slot_prompt + B * WH_SLOT_PMAX; m->slot_out = m->slot_fill + B * WH_SLOT_PMAX;
    S2S_HIP_CHECK(hipMalloc((void**)&m->smask, 3 * (size_t)m->cfg.vocab));
    S2S_HIP_CHECK(hipMemset(m->smask, 0, 3 * (size_t)m->cfg.vocab));
    m->owned.push_back(m->smask);
    return S2S_OK;
}

}  // namespace s2s

using namespace s2s;

extern "C" int32_t s2s_whisper_create(const s2s_whisper_config* cfg, int32_t max_batch, void** out) {
    S2S_REQUIRE(cfg && out && max_batch > 0 && max_batch <= 128, "whisper_create: bad arguments (max_batch 1..128)");
    S2S_REQUIRE(cfg->dtype == S2S_F32 || cfg->dtype == S2S_BF16, "whisper_create: bad dtype");
    S2S_REQUIRE(cfg->d_model % cfg->n_heads == 0 && cfg->d_model % 8 == 0 && cfg->n_mels % 8 == 0 && cfg->ffn % 8 == 0,
                "whisper_create: dimensions must be multiples of 8");
    Whisper* m = new Whisper();
    m->cfg = *cfg;
    m->max_batch = max_batch;
    m->esz = cfg->dtype == S2S_F32 ? 4 : 2;
    *out = m;
    return S2S_OK;
}

extern "C" int32_t s2s_whisper_destroy(void* model) {
    if (!model) return S2S_OK;
    Whisper* m = (Whisper*)model;
    guard_free(m->arena);
    if (m->host_flag) (void)hipHostFree(m->host_flag);
    if (m->step_dev) (void)hipFree(m->step_dev);
    if (m->mask_sup) (void)hipFree(m->mask_sup);
    for (auto& kv : m->step_graphs) (void)hipGraphExecDestroy(kv.second);
    for (auto& kv : m->slot_graphs) (void)hipGraphExecDestroy(kv.second);
    for (void* p : m->owned) (void)hipFree(p);
    if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
    delete m;
    return S2S_OK;
}

extern "C" int32_t s2s_whisper_set_tensor(void* model, const char* name, const void* dev_ptr, int64_t numel) {
    Whisper* m = (Whisper*)model;
    S2S_REQUIRE(m && name && dev_ptr, "whisper_set_tensor: null argument");
    if (m->finalized) return fail(S2S_ERR_STATE, "whisper_set_tensor: %s after finalize (captured graphs hold the old pointers)", name);
    S2S_REQUIRE(((uintptr_t)dev_ptr & 15) == 0, "whisper_set_tensor: %s is not 16-byte aligned", name);
    m->w[name] = Tensor{dev_ptr, (long)numel};
    return S2S_OK;
}

extern "C" int32_t s2s_whisper_finalize(void* model) {
    // workspace allocation uses synchronous (legacy-stream) memsets / copies, which HIP refuses - and which poison the other thread's
    // capture - while any stream of the process is capturing: take the capture lock
    std::lock_guard<std::mutex> no_capture(s2s::capture_mutex());
    Whisper* m = (Whisper*)model;
    S2S_REQUIRE(m, "whisper_finalize: null model");
    if (m->finalized) return S2S_OK;
    return finalize(m);
}

extern "C" int64_t s2s_whisper_workspace_bytes(void* model) {
    Whisper* m = (Whisper*)model;
    return m ? (int64_t)m->arena_bytes : 0;
}

extern "C" int32_t s2s_whisper_encode(void* model, void* stream, const void* feats_dev, int32_t B) {
    Whisper* m = (Whisper*)model;
    S2S_REQUIRE(m && m->finalized, "whisper_encode: model not finalized");
    S2S_REQUIRE(B > 0 && B <= m->max_batch && feats_dev, "whisper_encode: bad batch %d", B);
    S2S_REQUIRE(m->slots_live == 0, "whisper_encode: %d slot(s) of the continuous token loop are live on this engine (one mode at a time; use s2s_whisper_encode_at)", m->slots_live);
    return encode(m, (hipStream_t)stream, feats_dev, B);
}

extern "C" int32_t s2s_whisper_read_encoder_output(void* model, void* stream, void* dst_dev, int32_t B) {
    Whisper* m = (Whisper*)model;
    S2S_REQUIRE(m && m->finalized && dst_dev && B > 0 && B <= m->max_batch, "whisper_read_encoder_output: bad arguments");
    S2S_HIP_CHECK(hipMemcpyAsync(dst_dev, m->enc_out, (size_t)B * ENC_T * m->cfg.d_model * m->esz, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return S2S_OK;
}

extern "C" int32_t s2s_whisper_decode_greedy(void* model, void* stream, int32_t B, const int32_t* prompt_dev,
                                             int32_t P, const uint8_t* suppress_dev, const uint8_t* begin_suppress_dev,
                                             const uint8_t* prompt_fill_mask_dev, int32_t eos, int32_t max_new_tokens,
                                             int32_t min_steps, int32_t* out_ids_dev, int32_t* out_len_dev,
                                             float* last_logits_dev) {
    Whisper* m = (Whisper*)model;
    S2S_REQUIRE(m && m->finalized, "whisper_decode: model not finalized");
    S2S_REQUIRE(B > 0 && B <= m->max_batch && prompt_dev && P >= 1 && out_ids_dev && out_len_dev, "whisper_decode: bad arguments");
    S2S_REQUIRE(m->slots_live == 0, "whisper_decode: %d slot(s) of the continuous token loop are live on this engine (one mode at a time)", m->slots_live);
    S2S_REQUIRE(max_new_tokens >= 1 && P + max_new_tokens <= m->tmax, "whisper_decode: P + max_new_tokens = %d exceeds %d target positions", P + max_new_tokens, m->tmax);
    hipStream_t st = (hipStream_t)stream;
    const auto& c = m->cfg;
    const int row = max_new_tokens + P;
    S2S_HIP_CHECK(hipMemsetAsync(m->out_len, 0, sizeof(int) * 2 * m->max_batch, st));   // out_len + finished
    S2S_HIP_CHECK(hipMemsetAsync(m->out_ids, 0, sizeof(int) * (size_t)B * row, st));
    S2S_HIP_CHECK(hipMemsetAsync(m->pos, 0, sizeof(int) * 2 * m->max_batch, st));        // pos + step
    // the call's prompt, masks and scalars move into the engine's own buffers (what the captured step body reads)
    S2S_HIP_CHECK(hipMemcpyAsync(m->prompt_buf, prompt_dev, sizeof(int) * (size_t)B * P, hipMemcpyDeviceToDevice, st));
    auto set_mask = [&](unsigned char* dst, const uint8_t* src) -> hipError_t {
        return src ? hipMemcpyAsync(dst, src, c.vocab, hipMemcpyDeviceToDevice, st) : hipMemsetAsync(dst, 0, c.vocab, st);
    };
    S2S_HIP_CHECK(set_mask(m->mask_sup, suppress_dev));
    S2S_HIP_CHECK(set_mask(m->mask_begin, begin_suppress_dev));
    S2S_HIP_CHECK(set_mask(m->mask_fill, prompt_fill_mask_dev));
    whisper_set_params_kernel<<<1, 1, 0, st>>>(m->params_dev, P, max_new_tokens, eos, m->step_dev);
    S2S_LAUNCH_CHECK();
    // first input = prompt[:,0]
    S2S_HIP_CHECK(hipMemcpy2DAsync(m->cur_ids, sizeof(int), prompt_dev, sizeof(int) * P, sizeof(int), B, hipMemcpyDeviceToDevice, st));
    const int total_steps = P - 1 + max_new_tokens;
    const bool split_ok = c.dtype == S2S_BF16;
    const bool use_graph = graphs_enabled() && !prof_active();
    hipGraphExec_t exec = nullptr;
    bool graph_ok = use_graph;
    if (use_graph && step_graph(m, B, split_ok, &exec) != S2S_OK) {   // refused / invalidated capture: run this call as stream launches
        (void)hipGetLastError();
        graph_ok = false;
    }
    for (int s = 0; s < total_steps; ++s) {
        if (graph_ok) {
            S2S_HIP_CHECK(hipGraphLaunch(exec, st));
        } else {
            const int rc = step_body(m, st, B, split_ok);
            if (rc) return rc;
        }
        const bool in_prompt = s + 1 < P;
        if (last_logits_dev && s == total_steps - 1)
            S2S_HIP_CHECK(hipMemcpyAsync(last_logits_dev, m->logits, sizeof(float) * (size_t)B * c.vocab, hipMemcpyDeviceToDevice, st));
        // early exit: poll the finished flags every 8 generated tokens
        if (!in_prompt && s + 1 >= min_steps && ((s - (P - 1)) & 7) == 7 && s + 1 < total_steps) {
            S2S_HIP_CHECK(hipMemcpyAsync(m->host_flag, m->finished, sizeof(int) * B, hipMemcpyDeviceToHost, st));
            S2S_HIP_CHECK(hipStreamSynchronize(st));
            bool all = true;
            for (int b = 0; b < B; ++b) all = all && m->host_flag[b] != 0;
            if (all) break;
        }
    }
    S2S_HIP_CHECK(hipMemcpyAsync(out_ids_dev, m->out_ids, sizeof(int) * (size_t)B * row, hipMemcpyDeviceToDevice, st));
    S2S_HIP_CHECK(hipMemcpyAsync(out_len_dev, m->out_len, sizeof(int) * B, hipMemcpyDeviceToDevice, st));
    return S2S_OK;
}

// ---- continuous batching of the token loop (see include/s2s_mi355x.h) ---------------------------------------------------------------
extern "C" int32_t s2s_whisper_slots_config(void* model, void* stream, const uint8_t* suppress_dev, const uint8_t* begin_suppress_dev,
                                            const uint8_t* prompt_fill_mask_dev, int32_t eos) {
    Whisper* m = (Whisper*)model;
    S2S_REQUIRE(m && m->finalized, "whisper_slots_config: model not finalized");
    S2S_REQUIRE(m->slots_live == 0, "whisper_slots_config: %d slot(s) are live", m->slots_live);
    S2S_REQUIRE(eos >= 0 && eos < m->cfg.vocab, "whisper_slots_config: eos %d outside the vocabulary", eos);
    int rc = slots_alloc(m);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t V = m->cfg.vocab;
    auto set_mask = [&](unsigned char* dst, const uint8_t* src) -> hipError_t {
        return src ? hipMemcpyAsync(dst, src, V, hipMemcpyDeviceToDevice, st) : hipMemsetAsync(dst, 0, V, st);
    };
    S2S_HIP_CHECK(set_mask(m->smask, suppress_dev));
    S2S_HIP_CHECK(set_mask(m->smask + V, begin_suppress_dev));
    S2S_HIP_CHECK(set_mask(m->smask + 2 * V, prompt_fill_mask_dev));
    if (eos != m->slot_eos) {       // eos is a kernel argument of the captured step: new graphs
        for (auto& kv : m->slot_graphs) (void)hipGraphExecDestroy(kv.second);
        m->slot_graphs.clear();
        m->slot_eos = eos;
    }
    return S2S_OK;
}

extern "C" int32_t s2s_whisper_encode_at(void* model, void* stream, const void* feats_dev, int32_t B, int32_t slot0) {
    Whisper* m = (Whisper*)model;
    S2S_REQUIRE(m && m->finalized, "whisper_encode_at: model not finalized");
    S2S_REQUIRE(B > 0 && slot0 >= 0 && slot0 + B <= m->max_batch && feats_dev, "whisper_encode_at: slots %d .. %d outside 0 .. %d", slot0, slot0 + B, m->max_batch);
    if (m->slot_live_map.empty()) m->slot_live_map.assign(m->max_batch, 0);
    for (int i = 0; i < B; ++i)
        S2S_REQUIRE(!m->slot_live_map[slot0 + i], "whisper_encode_at: slot %d is %s (its cross K|V belongs to another session)", slot0 + i,
                    m->slot_live_map[slot0 + i] == 2 ? "live in the token loop" : "reserved by an earlier encoder pass");
    const int rc = encode(m, (hipStream_t)stream, feats_dev, B, slot0);
    if (rc) return rc;
    for (int i = 0; i < B; ++i) m->slot_live_map[slot0 + i] = 1;      // reserved: the batch API (one mode at a time) refuses from here on
    m->slots_live += B;
    return S2S_OK;
}

extern "C" int32_t s2s_whisper_slots_admit(void* model, void* stream, int32_t slot0, int32_t n, const int32_t* prompt_dev, int32_t P,
                                           const int32_t* max_new_dev, int32_t max_new_limit) {
    Whisper* m = (Whisper*)model;
    S2S_REQUIRE(m && m->finalized && m->slot_step && m->slot_eos >= 0, "whisper_slots_admit: call s2s_whisper_slots_config first");
    S2S_REQUIRE(n > 0 && slot0 >= 0 && slot0 + n <= m->max_batch && prompt_dev && max_new_dev, "whisper_slots_admit: bad arguments");
    S2S_REQUIRE(P >= 1 && P <= WH_SLOT_PMAX, "whisper_slots_admit: prefix of %d tokens (1 .. %d)", P, WH_SLOT_PMAX);
    S2S_REQUIRE(max_new_limit >= 1 && P + max_new_limit <= m->tmax, "whisper_slots_admit: P + max_new_tokens = %d exceeds %d target positions", P + max_new_limit, m->tmax);
    if (m->slot_live_map.empty()) m->slot_live_map.assign(m->max_batch, 0);
    for (int i = 0; i < n; ++i) S2S_REQUIRE(m->slot_live_map[slot0 + i] != 2, "whisper_slots_admit: slot %d is already live (double admit)", slot0 + i);
    whisper_slot_admit_kernel<<<cdiv(n, 64), 64, 0, (hipStream_t)stream>>>(slot0, n, prompt_dev, P, max_new_dev, max_new_limit, m->slot_step, m->slot_P, m->slot_maxnew, m->slot_active,
                                                                         m->slot_prompt, m->slot_fill, m->cur_ids, m->pos, m->out_len, m->finished);
    S2S_LAUNCH_CHECK();
    for (int i = 0; i < n; ++i) {
        m->slots_live += m->slot_live_map[slot0 + i] == 0;      // (admission without s2s_whisper_encode_at: cross K|V written by other means)
        m->slot_live_map[slot0 + i] = 2;
    }
    return S2S_OK;
}

extern "C" int32_t s2s_whisper_slots_step(void* model, void* stream, int32_t rows, int32_t n_steps) {
    Whisper* m = (Whisper*)model;
    S2S_REQUIRE(m && m->finalized && m->slot_step && m->slot_eos >= 0, "whisper_slots_step: call s2s_whisper_slots_config first");
    S2S_REQUIRE(rows > 0 && rows <= m->max_batch && n_steps >= 1, "whisper_slots_step: bad arguments (rows %d, steps %d)", rows, n_steps);
    hipStream_t st = (hipStream_t)stream;
    const bool split_ok = m->cfg.dtype == S2S_BF16;
    hipGraphExec_t exec = nullptr;
    bool graph_ok = graphs_enabled() && !prof_active();
    if (graph_ok && slot_graph(m, rows, split_ok, &exec) != S2S_OK) {
        (void)hipGetLastError();
        graph_ok = false;
    }
    for (int s = 0; s < n_steps; ++s) {
        if (graph_ok) {
            S2S_HIP_CHECK(hipGraphLaunch(exec, st));
        } else {
            const int rc = slot_step_body(m, st, rows, split_ok);
            if (rc) return rc;
        }
    }
    return S2S_OK;
}

extern "C" int32_t s2s_whisper_slots_collect(void* model, void* stream, int32_t rows, int32_t* pack_dev) {
    Whisper* m = (Whisper*)model;
    S2S_REQUIRE(m && m->finalized && m->slot_step && pack_dev && rows > 0 && rows <= m->max_batch, "whisper_slots_collect: bad arguments");
    whisper_slot_collect_kernel<<<rows, 256, 0, (hipStream_t)stream>>>(m->slot_active, m->finished, m->out_len, m->slot_fill, m->slot_out, m->tmax, pack_dev);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

extern "C" int32_t s2s_whisper_slots_release(void* model, void* stream, int32_t slot0, int32_t n) {
    Whisper* m = (Whisper*)model;
    S2S_REQUIRE(m && m->finalized && m->slot_step && n > 0 && slot0 >= 0 && slot0 + n <= m->max_batch, "whisper_slots_release: bad arguments");
    if (m->slot_live_map.empty()) m->slot_live_map.assign(m->max_batch, 0);
    const bool reset = slot0 == 0 && n == m->max_batch;          // the whole range = "drop everything" (scheduler abort): idle slots allowed
    for (int i = 0; i < n && !reset; ++i)
        S2S_REQUIRE(m->slot_live_map[slot0 + i], "whisper_slots_release: slot %d is neither reserved nor live (double release)", slot0 + i);
    whisper_slot_release_kernel<<<cdiv(n, 64), 64, 0, (hipStream_t)stream>>>(slot0, n, m->slot_active, m->finished, m->pos, m->cur_ids);
    S2S_LAUNCH_CHECK();
    for (int i = 0; i < n; ++i) {
        m->slots_live -= m->slot_live_map[slot0 + i] != 0;
        m->slot_live_map[slot0 + i] = 0;
    }
    return S2S_OK;
}

extern "C" int32_t s2s_whisper_slots_pack_stride(void* model) {
    Whisper* m = (Whisper*)model;
    return m && m->finalized ? 2 + WH_SLOT_PMAX + m->tmax : 0;
}

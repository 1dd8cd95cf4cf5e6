// Qwen3-TTS as a native kernel graph: talker (dense Qwen3 decoder) + code predictor (per-frame 15-step
// residual-code loop) + codec decoder (sliding-window transformer, causal transposed convolutions, ConvNeXt,
// SnakeBeta residual units), for one batch of sessions.
//
// Replaces FasterQwen3TTS.generate_*_streaming(...) as driven by the reference's Qwen3TTSHandler
// (TTS/qwen3_tts_handler.py:910-1003).  The package behind it (faster-qwen3-tts -> qwentts.cpp) is not available
// offline; the arithmetic is restated from the sibling architecture in transformers
// (models/qwen3_omni_moe/modeling_qwen3_omni_moe.py:2250-2620, 3100-3177, 3180-3697; models/qwen3/modeling_qwen3.py).
//
// HBM layout: residual streams f32; projections / caches in the model dtype; KV caches [L][B][max_pos][Hk*D];
// codec activations time-major [B][LPAD + T][C] with LPAD zero rows in front of every item so that every causal
// (dilated) convolution and transposed convolution is ONE GEMM over overlapping rows (no im2col).
#include "ops.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace s2s {

int rope_qknorm(hipStream_t st, void* qkv, int rows, int T_len, int H, int Hk, int D, const float* qw, const float* kw, float eps,
                const float* cosT, const float* sinT, const int* pos0, void* kc, void* vc, long c_bs, int dtype);
int kv_append_rows(hipStream_t st, const void* src, long src_rs, int k_off, int v_off, void* kc, void* vc, long c_bs,
                   const int* pos0, int B, int T_len, int n, int dtype);
int swiglu(hipStream_t st, const void* gu, void* out, long rows, int ffn, int dtype);
int add_int(hipStream_t st, int* p, int v, int n);
int talker_pick(hipStream_t st, const float* logits, int V, const unsigned char* sup_mask, const unsigned char* eos_mask, int eos,
                const int* frame_b, const int* min_b, const int* utt_b, int* code0, int* finished, int* n_frames, int* codes_out, int max_frames, int G,
                const void* emb, const void* pad, const float* hn, float* xnext, float* pin, void* pin_t, int E, bool pin_frag, int B, int dtype,
                const PickDebug& dbg, const SampleParams& sp, const void* trail, const int* n_trail, int max_trail);
int trail_set(hipStream_t st, const float* rows, const int* n_rows, int n, int Tr, int E, void* trail, int* n_trail, int max_trail, int slot0, int dtype);
int pred_pick(hipStream_t st, const float* logits, int V, long logits_rs, const int* frame_b, const int* utt_b, const int* finished, int* codes_out, int max_frames, int G,
              int g, const void* emb, float* xnext, float* pin, void* pin_t, int E, bool pin_frag, int B, int dtype, const PickDebug& dbg,
              const SampleParams& sp);
int tts_frame_end(hipStream_t st, int* pos_t, int* frame_b, int* finished, const int* budget_b, int max_pos, int B);
int tts_slots_set(hipStream_t st, int* pos_t, int* frame_b, int* min_b, int* budget_b, int* utt_b, int* finished, int* n_frames, int* code0,
                  const int* prompt_len, const int* min_frames, const int* budget, int min_all, int budget_all, int utt0, int max_frames, int slot0,
                  int n, bool live);
int tts_read_codes(hipStream_t st, const int* codes_out, int max_frames, int G, int slot0, int n, int frame0, const int* frame0_rows, int F, int* out);
int tts_valid_samples(hipStream_t st, const int* n_frames, int slot0, int n, int frame0, const int* frame0_rows, int F, int up, int chunk_samples, int* out);
int fold_norm_weight(hipStream_t st, const void* W, const float* gamma, void* out, long rows, int K, int dtype);
int gather_embed(hipStream_t st, const void* table, const int* ids, int id_stride, int id_offset, float* dst, long dst_rs, int B,
                 int dim, int accumulate, const void* addvec, int dtype);
int code_embed_mean(hipStream_t st, const void* table, const int* codes, float* h, long rows, int Q, int cb, int dim, int dtype);
int dwconv7_ln(hipStream_t st, const void* x, const float* w, const float* bias, const float* g, const float* be, void* y, int B,
               int Tn, int C, int lpad, float eps, int dtype);
int wav_out(hipStream_t st, const void* x, float* wav, int B, int Tn, int lpad, int skip, int n, int dtype);

constexpr int DEC_ROWS = 1024;   // most rows (sessions x new tokens) of a norm-free decode pass (gemm_dec.hip up to 128 rows, gemm_decw.hip above)
constexpr int MAX_SLOTS = 512;   // most rows of the frame loop (the code predictor's two-row pass runs 2 x slots rows)
constexpr int SS_PARTS = 256;   // most column slabs of a residual projection (hidden / 16)
constexpr int LPAD = 56;   // >= (7-1)*9 = 54 taps of history for the widest dilated conv, multiple of 8

struct TTensor { const void* ptr; long numel; };

struct StackCfg {
    int hidden, layers, heads, kv_heads, head_dim, ffn, qk_norm, window, max_pos;
    float eps;
};

struct Stack {
    StackCfg c;
    std::string prefix;      // weight name prefix, e.g. "talker."
    void *kc = nullptr, *vc = nullptr;          // [L][B][max_pos][Hk*D]
    const float *cosT = nullptr, *sinT = nullptr;
    // bf16 decode path (norm-free layer, gemm_dec.hip): per layer, the projections that consume a normed input with the RMSNorm
    // weight folded into their columns (qkv <- ln1, gate/up <- ln2); the reciprocal RMS is applied in their epilogue from the
    // partial sums of squares the previous residual projection wrote
    std::vector<void*> qkv_n, gu_n;         // fragment-major, norm weight folded in
    std::vector<void*> o_f, down_f;          // fragment-major copies of the residual projections
};

struct Tts {
    s2s_tts_config cfg;
    int max_batch, esz;
    std::map<std::string, TTensor> w;
    bool finalized = false;
    char* arena = nullptr;
    size_t arena_bytes = 0;
    Stack talker, pred, codec;
    // shared scratch (sized for the largest use)
    void *h, *qkv, *att, *gu, *act;
    float *x;          // talker residual rows [B*Tmax_prompt][E]
    float *xp;         // predictor residual rows [B*2][P]
    float *hn;         // talker normed last hidden (f32) [B][E]
    void *hn_t;        // same in model dtype
    float *logits;     // [B][max(codec_vocab, codebook)]
    float *xnext;      // next talker input [B][E]
    float *pin;        // predictor input before in_proj [B*2][E]
    void *pin_t;
    int *pos_t, *pos_p, *code0, *finished, *n_frames;
    int *codes_out;    // [B][max_frames][G]
    unsigned char* sup_mask;   // [codec_vocab] built-in suppress mask (codec specials except eos)
    int frames_done = 0, cur_B = 0, prompt_max = 0;
    // codec decoder buffers
    float* cx;  void *ca, *cb_, *cc;   // residual stream + ping-pong activations
    std::vector<void*> cstate;         // streaming history rows, one [B][LPAD][C] block per history-carrying stage
    std::vector<int> cstate_c;         // channels C of each of them (the order codec_run visits them in)
    int* pos_c = nullptr;              // codec transformer positions (device, [B])
    int codec_frames_done = 0;
    int* host_flag = nullptr;
    // decode-frame graph: one captured frame body per batch size, replayed once per frame (all per-frame state lives on the device)
    // per-SLOT frame state (device): frame index, eos gate threshold, frame budget, utterance id (sampler key).  A slot is one row of
    // the frame loop; sessions are admitted into / released from slots while the others keep generating (s2s_tts_admit / _release)
    int *frame_b = nullptr, *min_b = nullptr, *budget_b = nullptr, *utt_b = nullptr;
    // streaming-text mode (s2s_tts_set_trailing): per slot the text rows that were NOT pre-filled, consumed one per frame in place of pad_embed
    void* trail = nullptr;             // [B][max_prompt][E] model dtype
    int* n_trail = nullptr;            // [B] rows valid per slot (0: every frame adds pad_embed - the pre-filled-text mode)
    int next_utt = 1;                  // utterance ids handed out by the legacy prefill
    hipStream_t cap_stream = nullptr;
    std::map<int, hipGraphExec_t> frame_graphs;
    std::mutex graphs_mu;                       // frame_graphs is read by the lane thread and replaced by s2s_tts_set_sampling
    std::vector<hipGraphExec_t> retired_graphs; // executables replaced while a stream may still be replaying them: destroyed with the engine
    std::vector<void*> head_n;         // code-predictor heads, fragment-major with pred.norm folded in (bf16 decode path)
    void* in_proj_f = nullptr;         // fragment-major pred.in_proj.w
    // norm-free decode path (gemm_dec.hip): bf16 copy of the residual rows + two partial-sum-of-squares buffers
    void* xb = nullptr;                // [DEC_ROWS][max hidden] model dtype
    float *ssA = nullptr, *ssB = nullptr;   // [SS_PARTS][DEC_ROWS]
    void* dec_ws = nullptr;            // K-slice workspace of the decode projections (gemm_dec2.hip): tickets + partial tiles
    SampleParams sp_talker{0.f, 0, 1.f, 1.f, 0ull}, sp_pred{0.f, 0, 1.f, 1.f, 0ull};   // temperature 0: greedy (s2s_tts_set_sampling)
    PickDebug dbg{};                   // teacher forcing / logit capture (s2s_tts_generate_forced); null on the serving path
    int* forced_buf = nullptr;         // [max_batch][max_frames][G], allocated on first forced call
    int xb_width = 0;                  // columns of xb
    size_t act_elems = 0;              // elements per session of each codec activation buffer (ca / cb_ / cc)
    int ss_parts_A = 0;                // parts currently valid in ssA (written by the last residual projection / rows_prep)
    std::vector<void*> owned;          // device buffers created at finalize (folded weights)
};

static const TTensor* tf(const Tts* m, const std::string& n) {
    auto it = m->w.find(n);
    return it == m->w.end() ? nullptr : &it->second;
}
static const void* TP(const Tts* m, const std::string& n) { const TTensor* t = tf(m, n); return t ? t->ptr : nullptr; }
static const float* TPF(const Tts* m, const std::string& n) { return (const float*)TP(m, n); }

static int need(const Tts* m, const std::string& n, long numel) {
    const TTensor* t = tf(m, n);
    if (!t) return fail(S2S_ERR_STATE, "tts: missing tensor %s", n.c_str());
    if (t->numel != numel) return fail(S2S_ERR_STATE, "tts: tensor %s has %ld elements, expected %ld", n.c_str(), t->numel, numel);
    return S2S_OK;
}

static int check_stack(const Tts* m, const Stack& s) {
    const auto& c = s.c;
    const long qkv_n = (long)(c.heads + 2 * c.kv_heads) * c.head_dim;
    int rc;
    for (int l = 0; l < c.layers; ++l) {
        const std::string p = s.prefix + std::to_string(l) + ".";
        if ((rc = need(m, p + "ln1", c.hidden))) return rc;
        if ((rc = need(m, p + "qkv.w", qkv_n * c.hidden))) return rc;
        if (c.qk_norm) {
            if ((rc = need(m, p + "qnorm", c.head_dim))) return rc;
            if ((rc = need(m, p + "knorm", c.head_dim))) return rc;
        }
        if ((rc = need(m, p + "o.w", (long)c.hidden * c.heads * c.head_dim))) return rc;
        if ((rc = need(m, p + "ln2", c.hidden))) return rc;
        if ((rc = need(m, p + "gu.w", 2L * c.ffn * c.hidden))) return rc;
        if ((rc = need(m, p + "down.w", (long)c.hidden * c.ffn))) return rc;
    }
    if ((rc = need(m, s.prefix + "rope.cos", (long)c.max_pos * c.head_dim / 2))) return rc;
    if ((rc = need(m, s.prefix + "rope.sin", (long)c.max_pos * c.head_dim / 2))) return rc;
    return S2S_OK;
}

using Plan = ArenaPlan;      // (common.h: the shared carve-up, with poisoned gaps under S2S_GUARD)

// rows of a stage for a streamed CONTINUATION chunk of F frames: every frame contributes its full upsampling product (the first chunk
// of a stream is shorter: each causal transposed conv trims its first output group) - this is what the activation buffers must hold
static long codec_len_cont(const s2s_tts_config& c, long F, int stage) {
    long T = F;
    for (int i = 0; i < 2; ++i) T *= c.up_ratio[i];
    for (int i = 0; i < stage; ++i) T *= c.up_rate[i];
    return T;
}

static long codec_len(const s2s_tts_config& c, long F, int stage) {
    // stage 0: after the two x2 upsamplers; stage i (1..4): after decoder block i
    long T = F;
    for (int i = 0; i < 2; ++i) T *= c.up_ratio[i];
    for (int i = 0; i < stage; ++i) T = (T - 1) * c.up_rate[i];
    return T;
}

static int finalize(Tts* m) {
    const auto& c = m->cfg;
    int rc;
    m->talker.c = StackCfg{c.t_hidden, c.t_layers, c.t_heads, c.t_kv_heads, c.t_head_dim, c.t_ffn, 1, 0, c.max_pos, c.eps};
    m->talker.prefix = "talker.";
    m->pred.c = StackCfg{c.p_hidden, c.p_layers, c.p_heads, c.p_kv_heads, c.p_head_dim, c.p_ffn, 1, 0, c.n_groups + 1, c.eps};
    m->pred.prefix = "pred.";
    m->codec.c = StackCfg{c.c_hidden, c.c_layers, c.c_heads, c.c_heads, c.c_hidden / c.c_heads, c.c_ffn, 0, c.c_window,
                          std::max(c.max_codec_frames, c.max_pos), c.c_eps};
    m->codec.prefix = "codec.";
    for (Stack* s : {&m->talker, &m->pred, &m->codec}) {
        if ((rc = check_stack(m, *s))) return rc;
        s->cosT = TPF(m, s->prefix + "rope.cos");
        s->sinT = TPF(m, s->prefix + "rope.sin");
    }
    const long E = c.t_hidden, P = c.p_hidden, G = c.n_groups, CB = c.codebook, H = c.c_hidden;
    if ((rc = need(m, "talker.norm", E)) || (rc = need(m, "talker.emb", (long)c.codec_vocab * E)) || (rc = need(m, "talker.head.w", (long)c.codec_vocab * E)) ||
        (rc = need(m, "talker.pad_embed", E)) || (rc = need(m, "pred.norm", P)))
        return rc;
    if (P != E && ((rc = need(m, "pred.in_proj.w", P * E)) || (rc = need(m, "pred.in_proj.b", P)))) return rc;
    for (int g = 0; g < G - 1; ++g) {
        if ((rc = need(m, "pred.emb." + std::to_string(g), CB * E)) || (rc = need(m, "pred.head." + std::to_string(g) + ".w", CB * P))) return rc;
    }
    if ((rc = need(m, "codec.emb", CB * G * H)) || (rc = need(m, "codec.norm", H))) return rc;
    for (int i = 0; i < 2; ++i) {
        const std::string q = "codec.up." + std::to_string(i) + ".";
        const long r = c.up_ratio[i];
        if ((rc = need(m, q + "tconv.w", r * H * H)) || (rc = need(m, q + "tconv.b", r * H)) || (rc = need(m, q + "dw.w", H * 7)) || (rc = need(m, q + "dw.b", H)) ||
            (rc = need(m, q + "ln.g", H)) || (rc = need(m, q + "ln.b", H)) || (rc = need(m, q + "pw1.w", 4 * H * H)) || (rc = need(m, q + "pw1.b", 4 * H)) ||
            (rc = need(m, q + "pw2.w", 4 * H * H)) || (rc = need(m, q + "pw2.b", H)))
            return rc;
    }
    const long Dd = c.c_decoder_dim;
    if ((rc = need(m, "codec.dec.in.w", Dd * 7 * H)) || (rc = need(m, "codec.dec.in.b", Dd))) return rc;
    for (int i = 0; i < c.n_dec_blocks; ++i) {
        const long cin = Dd >> i, cout = Dd >> (i + 1), r = c.up_rate[i];
        const std::string q = "codec.dec." + std::to_string(i) + ".";
        if ((rc = need(m, q + "snake.a", cin)) || (rc = need(m, q + "snake.ib", cin)) || (rc = need(m, q + "tconv.w", r * cout * 2 * cin)) || (rc = need(m, q + "tconv.b", r * cout))) return rc;
        for (int j = 0; j < 3; ++j) {
            const std::string u = q + "ru" + std::to_string(j) + ".";
            if ((rc = need(m, u + "s1.a", cout)) || (rc = need(m, u + "s1.ib", cout)) || (rc = need(m, u + "c1.w", cout * 7 * cout)) || (rc = need(m, u + "c1.b", cout)) ||
                (rc = need(m, u + "s2.a", cout)) || (rc = need(m, u + "s2.ib", cout)) || (rc = need(m, u + "c2.w", cout * cout)) || (rc = need(m, u + "c2.b", cout)))
                return rc;
        }
    }
    const long co = Dd >> c.n_dec_blocks;
    if ((rc = need(m, "codec.dec.out.a", co)) || (rc = need(m, "codec.dec.out.ib", co)) || (rc = need(m, "codec.dec.out.w", 7 * co)) || (rc = need(m, "codec.dec.out.b", 1))) return rc;

    // ---- workspace
    const size_t B = m->max_batch, e = m->esz;
    const size_t Tp = c.max_prompt, F = c.max_codec_frames;
    // (at least DEC_ROWS rows: fragment-major decode activations are addressed in whole 16-row tiles)
    auto stack_scratch_rows = std::max<size_t>(std::max<size_t>(B * Tp, B * F), DEC_ROWS);
    size_t wid_qkv = 0, wid_ffn = 0, wid_hid = 0, wid_att = 0;
    for (Stack* s : {&m->talker, &m->pred, &m->codec}) {
        wid_qkv = std::max<size_t>(wid_qkv, (size_t)(s->c.heads + 2 * s->c.kv_heads) * s->c.head_dim);
        wid_ffn = std::max<size_t>(wid_ffn, s->c.ffn);
        wid_hid = std::max<size_t>(wid_hid, s->c.hidden);
        wid_att = std::max<size_t>(wid_att, (size_t)s->c.heads * s->c.head_dim);
    }
    Plan a;
    std::vector<size_t> o;
    auto T = [&](size_t b) { o.push_back(a.take(b)); };
    T(stack_scratch_rows * wid_hid * e);       // 0 h
    T(stack_scratch_rows * wid_qkv * e);       // 1 qkv
    T(stack_scratch_rows * wid_att * e);       // 2 att
    T(stack_scratch_rows * 2 * wid_ffn * e);   // 3 gu
    T(stack_scratch_rows * wid_ffn * e);       // 4 act
    T(B * Tp * E * 4);                         // 5 x
    T(B * 2 * P * 4);                          // 6 xp
    T(B * E * 4);                              // 7 hn
    T(B * E * e);                              // 8 hn_t
    T(B * (size_t)std::max(c.codec_vocab, 2 * c.codebook) * 4);   // 9 logits (the predictor's 2-row prefill pass writes both rows' logits)
    T(B * E * 4);                              // 10 xnext
    T(B * 2 * E * 4);                          // 11 pin
    T(((B * 2 + 15) / 16 * 16) * E * e);       // 12 pin_t (whole 16-row tiles when fragment-major)
    T((12 * B + 32) * 4);                      // 13 ints
    T(B * (size_t)c.max_frames * G * 4);       // 14 codes_out
    T((size_t)c.codec_vocab);                  // 15 sup_mask
    for (Stack* s : {&m->talker, &m->pred, &m->codec}) {   // 16..21 caches
        const size_t per = (size_t)s->c.layers * B * s->c.max_pos * s->c.kv_heads * s->c.head_dim * e;
        T(per); T(per);
    }
    // codec conv stack: the widest stage decides (rows * channels)
    size_t act_elems = 0;
    {
        // (sized for a continuation chunk of max_codec_frames frames - longer than a first chunk of the same frames, codec_len_cont)
        size_t r0 = codec_len_cont(c, F, 0);
        act_elems = std::max<size_t>(act_elems, (LPAD + r0) * (size_t)std::max<long>(4 * H, Dd));
        for (int i = 1; i <= c.n_dec_blocks; ++i) act_elems = std::max<size_t>(act_elems, (LPAD + codec_len_cont(c, F, i)) * (size_t)(Dd >> i));
        act_elems = std::max<size_t>(act_elems, (LPAD + codec_len_cont(c, F, c.n_dec_blocks)) * 8);   // final 1-channel conv output
    }
    m->act_elems = act_elems;
    T(B * F * 4 * H * 4);                      // 22 cx (f32 residual for the pre-transformer / unused later)
    T(B * act_elems * e);                      // 23 ca
    T(B * act_elems * e);                      // 24 cb
    T(B * act_elems * e);                      // 25 cc
    T(B * (size_t)c.max_prompt * E * e);       // 26 trail (streaming-text rows per slot)
    T(B * 4 + 64);                             // 27 n_trail
    m->arena_bytes = a.off;
    if ((rc = a.commit(&m->arena, "tts"))) return rc;
    char* b0 = m->arena;
    m->h = b0 + o[0]; m->qkv = b0 + o[1]; m->att = b0 + o[2]; m->gu = b0 + o[3]; m->act = b0 + o[4];
    m->x = (float*)(b0 + o[5]); m->xp = (float*)(b0 + o[6]); m->hn = (float*)(b0 + o[7]); m->hn_t = b0 + o[8];
    m->logits = (float*)(b0 + o[9]); m->xnext = (float*)(b0 + o[10]); m->pin = (float*)(b0 + o[11]); m->pin_t = b0 + o[12];
    int* ints = (int*)(b0 + o[13]);
    m->pos_t = ints; m->code0 = ints + 2 * B; m->finished = ints + 3 * B; m->n_frames = ints + 4 * B;
    m->frame_b = ints + 5 * B; m->min_b = ints + 6 * B;               // per-slot frame index / eos gate threshold
    m->budget_b = ints + 10 * B + 16; m->utt_b = ints + 11 * B + 16;  // per-slot frame budget / utterance id
    m->codes_out = (int*)(b0 + o[14]);
    m->sup_mask = (unsigned char*)(b0 + o[15]);
    m->talker.kc = b0 + o[16]; m->talker.vc = b0 + o[17];
    m->pred.kc = b0 + o[18]; m->pred.vc = b0 + o[19];
    m->codec.kc = b0 + o[20]; m->codec.vc = b0 + o[21];
    m->cx = (float*)(b0 + o[22]); m->ca = b0 + o[23]; m->cb_ = b0 + o[24]; m->cc = b0 + o[25];
    m->trail = b0 + o[26]; m->n_trail = (int*)(b0 + o[27]);
    S2S_HIP_CHECK(hipMemset(m->n_trail, 0, sizeof(int) * B));
    // suppress mask: every codec special token except eos can never be emitted as codebook-0
    std::vector<unsigned char> sup(c.codec_vocab, 0);
    for (int i = c.codebook; i < c.codec_vocab; ++i) sup[i] = (i == c.codec_eos) ? 0 : 1;
    S2S_HIP_CHECK(hipMemcpy(m->sup_mask, sup.data(), sup.size(), hipMemcpyHostToDevice));
    S2S_HIP_CHECK(hipHostMalloc((void**)&m->host_flag, sizeof(int) * B, 0));
    {   // code-predictor positions are host constants per group pass: row g = tokens cached before pass g (0, 2, 3, ...)
        std::vector<int> tab((size_t)c.n_groups * B);
        for (int g = 0; g < c.n_groups; ++g)
            for (long b = 0; b < (long)B; ++b) tab[(size_t)g * B + b] = g == 0 ? 0 : g + 1;
        S2S_HIP_CHECK(hipMalloc((void**)&m->pos_p, tab.size() * sizeof(int)));
        S2S_HIP_CHECK(hipMemcpy(m->pos_p, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    // streaming history: 2 (ConvNeXt inputs) + 2 (ConvNeXt outputs) + per decoder block (1 + 3) + 1 (final conv input)
    {
        const size_t cmax = (size_t)std::max<long>(H, Dd);
        const int n_slots = 4 + 4 * c.n_dec_blocks + 1;
        // channels per history block, in codec_run's order: 2 x (upsampler dw input, ConvNeXt output) at H, the decoder in-conv's snake at
        // Dd, then per decoder block the transposed conv's snake + the three units' next snakes at Dd >> (i + 1)
        m->cstate_c.assign(4, (int)H);
        m->cstate_c.push_back((int)Dd);
        for (int i = 0; i < c.n_dec_blocks; ++i)
            for (int j = 0; j < 4; ++j) m->cstate_c.push_back((int)(Dd >> (i + 1)));
        for (int i = 0; i < n_slots; ++i) {
            ArenaPlan hp;                      // (one buffer per plan: under S2S_GUARD each history block gets its own poisoned gap / unmapped tail)
            hp.take(B * LPAD * cmax * e);
            char* p = nullptr;
            if ((rc = hp.commit(&p, "tts.hist"))) return rc;
            m->cstate.push_back(p);
        }
        m->pos_c = ints + 8 * B + 8;
    }
    if (c.dtype == S2S_BF16 && getenv("S2S_NO_DEC_FUSION") == nullptr && E % 32 == 0 && P % 32 == 0 && E >= 256 && P >= 256 && c.t_ffn % 32 == 0 &&
        c.p_ffn % 32 == 0 && c.t_ffn >= 256 && c.p_ffn >= 256 && (c.t_heads * c.t_head_dim) % 32 == 0 && (c.p_heads * c.p_head_dim) % 32 == 0 &&
        c.t_heads * c.t_head_dim >= 256 && c.p_heads * c.p_head_dim >= 256 && E / 16 <= SS_PARTS && P / 16 <= SS_PARTS) {
        // (ffn / hidden multiples of 32 also make every 16-row tile of the fragment-major weight copies complete)
        // norm-free decode path: RMSNorm weights folded into the columns of the consuming projections (own copies; the
        // prefill path keeps the checkpoint's matrices)
        m->xb_width = (int)std::max(std::max(E, P), (long)c.c_hidden);
        {   // one small arena (zero-filled: padding rows of fragment tiles must stay finite; the K-slice tickets start at zero)
            ArenaPlan dp;
            const size_t o_xb = dp.take((size_t)DEC_ROWS * m->xb_width * e), o_a = dp.take(sizeof(float) * SS_PARTS * DEC_ROWS),
                         o_b = dp.take(sizeof(float) * SS_PARTS * DEC_ROWS), o_ws = dp.take(DEC_WS_BYTES);
            char* base = nullptr;
            if ((rc = dp.commit(&base, "tts.dec"))) return rc;
            m->xb = base + o_xb; m->ssA = (float*)(base + o_a); m->ssB = (float*)(base + o_b); m->dec_ws = base + o_ws;
            m->owned.push_back(base);
        }
        // The fragment-major copies ("<name>.f": s2s_op_to_frag of the projection, RMSNorm weight folded into qkv / gate-up / the
        // predictor heads) are made once by the host side and shared by every engine replica on the GPU; the norm-free decode
        // path is on only when all of them were registered.
        auto fragw = [&](const std::string& name, long rows, long K, void** out) -> bool {
            const TTensor* t = tf(m, name);
            if (!t || t->numel != (rows + 15) / 16 * 16 * K) return false;
            *out = const_cast<void*>(t->ptr);
            return true;
        };
        bool all = true;
        for (Stack* s : {&m->talker, &m->pred}) {
            const auto& sc = s->c;
            const long nq = (long)(sc.heads + 2 * sc.kv_heads) * sc.head_dim;
            s->qkv_n.assign(sc.layers, nullptr);
            s->gu_n.assign(sc.layers, nullptr);
            s->o_f.assign(sc.layers, nullptr);
            s->down_f.assign(sc.layers, nullptr);
            for (int l = 0; l < sc.layers; ++l) {
                const std::string p = s->prefix + std::to_string(l) + ".";
                all = all && fragw(p + "qkv.f", nq, sc.hidden, &s->qkv_n[l]) && fragw(p + "gu.f", 2L * sc.ffn, sc.hidden, &s->gu_n[l]) &&
                      fragw(p + "o.f", sc.hidden, (long)sc.heads * sc.head_dim, &s->o_f[l]) && fragw(p + "down.f", sc.hidden, sc.ffn, &s->down_f[l]);
            }
        }
        m->head_n.assign(c.n_groups - 1, nullptr);
        for (int g = 0; g < c.n_groups - 1; ++g) all = all && fragw("pred.head." + std::to_string(g) + ".f", c.codebook, c.p_hidden, &m->head_n[g]);
        if (P != E) all = all && fragw("pred.in_proj.f", P, E, &m->in_proj_f);
        if (!all) {   // no (or incomplete) fragment-major weights: the engine runs the rmsnorm + row-major projections path
            m->talker.qkv_n.clear();
            m->pred.qkv_n.clear();
            m->head_n.clear();
        }
        {   // the codec's pre-transformer takes the same path when its copies are there (B x chunk frames rows: row slabs of 128)
            Stack* s = &m->codec;
            const auto& sc = s->c;
            const long nq = (long)(sc.heads + 2 * sc.kv_heads) * sc.head_dim;
            bool ok = all && getenv("S2S_NO_CODEC_DEC") == nullptr && sc.hidden % 32 == 0 && sc.hidden >= 256 && sc.ffn % 32 == 0 && sc.ffn >= 256 && (sc.heads * sc.head_dim) % 32 == 0 &&
                      sc.heads * sc.head_dim >= 256 && sc.hidden / 16 <= SS_PARTS;
            s->qkv_n.assign(sc.layers, nullptr); s->gu_n.assign(sc.layers, nullptr); s->o_f.assign(sc.layers, nullptr); s->down_f.assign(sc.layers, nullptr);
            for (int l = 0; ok && l < sc.layers; ++l) {
                const std::string p = s->prefix + std::to_string(l) + ".";
                ok = fragw(p + "qkv.f", nq, sc.hidden, &s->qkv_n[l]) && fragw(p + "gu.f", 2L * sc.ffn, sc.hidden, &s->gu_n[l]) &&
                     fragw(p + "o.f", sc.hidden, (long)sc.heads * sc.head_dim, &s->o_f[l]) && fragw(p + "down.f", sc.hidden, sc.ffn, &s->down_f[l]);
            }
            if (!ok) s->qkv_n.clear();
        }
    }
    m->finalized = true;
    return S2S_OK;
}

static bool fused_attn_step() {
    static const bool on = getenv("S2S_NO_FUSED_ATTN") == nullptr;   // A/B switch
    return on;
}

static inline void* at(void* p, size_t bytes) { return (char*)p + bytes; }
static inline const void* at(const void* p, size_t bytes) { return (const char*)p + bytes; }

static bool dec_fusion_ok(const Tts* m, const Stack& s, int M) {
    // (every operand of the decode projections is fragment-major on this path, so any row count up to DEC_ROWS is served: the 16-column
    // / column-block kernels up to 128 rows, the row-block kernel of gemm_decw.hip above)
    return m->cfg.dtype == S2S_BF16 && M <= DEC_ROWS && !s.qkv_n.empty() && m->xb && s.c.hidden <= m->xb_width;
}

// attention of one stack pass (q/k head norms, RoPE, cache append, softmax(QK^T)V) from m->qkv into m->att
static int stack_attention(Tts* m, hipStream_t st, Stack& s, int l, int B, int T_len, const int* pos0, int kv_cap, const int* kv_len_dev,
                           bool* o_frag = nullptr) {
    const auto& c = s.c;
    const int dt = m->cfg.dtype, e = m->esz;
    const int H = c.heads, Hk = c.kv_heads, D = c.head_dim;
    const int nq = (H + 2 * Hk) * D, M = B * T_len;
    const size_t cache_layer = (size_t)m->max_batch * c.max_pos * Hk * D * e;
    const std::string p = s.prefix + std::to_string(l) + ".";
    void* kc = at(s.kc, l * cache_layer);
    void* vc = at(s.vc, l * cache_layer);
    int rc;
    if (T_len == 1 && pos0 && !kv_len_dev && fused_attn_step()) {
        // decode step: head norms + RoPE + cache append + attention in ONE launch
        AttnStepParams a{};
        a.qkv = m->qkv; a.qkv_rs = nq;
        a.K = kc; a.V = vc; a.kv_bs = (long)c.max_pos * Hk * D; a.kv_rs = (long)Hk * D;
        a.O = m->att; a.o_bs = (long)H * D;
        a.qw = c.qk_norm ? TPF(m, p + "qnorm") : nullptr; a.kw = c.qk_norm ? TPF(m, p + "knorm") : nullptr; a.eps = c.eps;
        a.cosT = s.cosT; a.sinT = s.sinT; a.pos = pos0;
        a.B = B; a.H = H; a.Hkv = Hk; a.D = D; a.kv_cap = kv_cap; a.window = c.window; a.scale = 1.0f / sqrtf((float)D);
        a.o_frag = o_frag != nullptr && (H * D) % 32 == 0;     // fragment-major output rows for the out-projection that follows
        if (o_frag) *o_frag = a.o_frag;
        return attn_step(st, a, dt);
    }
    const bool rows_frag = o_frag != nullptr && (H * D) % 32 == 0;     // several new tokens per session: the row kernel writes the same layout
    if (o_frag) *o_frag = rows_frag;
    if ((rc = rope_qknorm(st, m->qkv, M, T_len, H, Hk, D, c.qk_norm ? TPF(m, p + "qnorm") : nullptr, c.qk_norm ? TPF(m, p + "knorm") : nullptr,
                          c.eps, s.cosT, s.sinT, pos0, kc, vc, (long)c.max_pos * Hk * D, dt))) return rc;
    AttnRowsParams a{};
    a.Q = m->qkv; a.q_bs = (long)T_len * nq; a.q_rs = nq;
    a.K = kc; a.V = vc; a.kv_bs = (long)c.max_pos * Hk * D; a.kv_rs = (long)Hk * D;
    a.O = m->att; a.o_bs = (long)T_len * H * D; a.o_rs = (long)H * D;
    a.kv_len = kv_len_dev; a.kv_len_const = kv_cap; a.B = B; a.H = H; a.Hkv = Hk; a.Tq = T_len; a.D = D;
    a.causal = 1; a.causal_off = 0; a.causal_off_arr = pos0; a.window = c.window; a.scale = 1.0f / sqrtf((float)D);
    a.o_frag = rows_frag;
    return attn_rows(st, a, dt);
}

// Norm-free decode pass (bf16, M = B * T_len <= DEC_ROWS rows): five launches per layer.  On entry m->xb / m->ssA must
// hold the bf16 copy and the partial sums of squares (m->ss_parts_A parts) of the f32 residual rows x; on exit they
// describe the final rows again (for a head with the final norm folded in).
static int stack_forward_dec(Tts* m, hipStream_t st, Stack& s, float* x, int B, int T_len, const int* pos0, int kv_cap) {
    const auto& c = s.c;
    const int H = c.heads, Hk = c.kv_heads, D = c.head_dim, hid = c.hidden;
    const int nq = (H + 2 * Hk) * D, M = B * T_len;
    const bool cached = &s != &m->talker;    // predictor: re-read 15 times per frame; codec: once per 128-row slab - keep them in L2 / MALL
    int rc;
    for (int l = 0; l < c.layers; ++l) {
        const std::string p = s.prefix + std::to_string(l) + ".";
        DecGemm q{};
        q.ws = m->dec_ws; q.ws_bytes = DEC_WS_BYTES;
        q.A = (const __hip_bfloat16*)m->xb; q.a_frag = true; q.W = (const __hip_bfloat16*)s.qkv_n[l]; q.M = M; q.N = nq; q.K = hid;
        q.ss_in = m->ssA; q.n_parts_in = m->ss_parts_A; q.ss_stride = DEC_ROWS; q.inv_dim = 1.0f / (float)hid; q.eps = c.eps;
        q.mode = 0; q.C = m->qkv; q.ldc = nq; q.act = S2S_ACT_NONE; q.w_cached = cached;
        if ((rc = gemm_dec(st, q))) return rc;
        bool att_frag = false;
        if ((rc = stack_attention(m, st, s, l, B, T_len, pos0, kv_cap, nullptr, &att_frag))) return rc;
        DecGemm o{};
        o.ws = m->dec_ws; o.ws_bytes = DEC_WS_BYTES;
        o.A = (const __hip_bfloat16*)m->att; o.lda = (long)H * D; o.a_frag = att_frag; o.W = (const __hip_bfloat16*)s.o_f[l];
        o.M = M; o.N = hid; o.K = H * D; o.mode = 2; o.X = x; o.ldx = hid; o.XB = (__hip_bfloat16*)m->xb; o.ldxb = hid; o.xb_frag = true; o.ss_out = m->ssB;
        o.ss_stride = DEC_ROWS; o.act = S2S_ACT_NONE; o.w_cached = cached;
        if ((rc = gemm_dec(st, o))) return rc;
        const int parts_o = dec_out_parts(o);
        DecGemm g{};
        g.ws = m->dec_ws; g.ws_bytes = DEC_WS_BYTES;
        g.A = (const __hip_bfloat16*)m->xb; g.a_frag = true; g.W = (const __hip_bfloat16*)s.gu_n[l]; g.M = M; g.N = c.ffn; g.K = hid;
        g.ss_in = m->ssB; g.n_parts_in = parts_o; g.ss_stride = DEC_ROWS; g.inv_dim = 1.0f / (float)hid; g.eps = c.eps;
        g.mode = 0; g.C = m->act; g.ldc = c.ffn; g.c_frag = true; g.act = S2S_ACT_NONE; g.glu = true; g.w_cached = cached;
        if ((rc = gemm_dec(st, g))) return rc;
        DecGemm d{};
        d.ws = m->dec_ws; d.ws_bytes = DEC_WS_BYTES;
        d.A = (const __hip_bfloat16*)m->act; d.a_frag = true; d.W = (const __hip_bfloat16*)s.down_f[l];
        d.M = M; d.N = hid; d.K = c.ffn; d.mode = 2; d.X = x; d.ldx = hid; d.XB = (__hip_bfloat16*)m->xb; d.ldxb = hid; d.xb_frag = true; d.ss_out = m->ssA;
        d.ss_stride = DEC_ROWS; d.act = S2S_ACT_NONE; d.w_cached = cached;
        if ((rc = gemm_dec(st, d))) return rc;
        m->ss_parts_A = dec_out_parts(d);
    }
    return S2S_OK;
}

// One pass of a decoder stack over T_len new tokens per session.  x: f32 residual rows [B*T_len][hidden].
// pos0: device int32[B] = tokens already cached per session (nullptr -> 0).  kv_cap: host upper bound of pos0 + T_len.
static int stack_forward(Tts* m, hipStream_t st, Stack& s, float* x, int B, int T_len, const int* pos0, int kv_cap, const int* kv_len_dev) {
    const auto& c = s.c;
    const int dt = m->cfg.dtype;
    const int H = c.heads, Hk = c.kv_heads, D = c.head_dim, hid = c.hidden;
    const int nq = (H + 2 * Hk) * D, M = B * T_len;
    int rc;
    for (int l = 0; l < c.layers; ++l) {
        const std::string p = s.prefix + std::to_string(l) + ".";
        if ((rc = rmsnorm(st, x, TPF(m, p + "ln1"), m->h, M, hid, c.eps, dt))) return rc;
        if ((rc = gemm(st, m->h, hid, TP(m, p + "qkv.w"), nullptr, nullptr, m->qkv, nq, M, nq, hid, dt, dt, S2S_ACT_NONE, false))) return rc;
        if ((rc = stack_attention(m, st, s, l, B, T_len, pos0, kv_cap, kv_len_dev))) return rc;
        if ((rc = gemm(st, m->att, H * D, TP(m, p + "o.w"), nullptr, x, x, hid, M, hid, H * D, dt, S2S_F32, S2S_ACT_NONE, false))) return rc;
        if ((rc = rmsnorm(st, x, TPF(m, p + "ln2"), m->h, M, hid, c.eps, dt))) return rc;
        if (M <= 64) {   // decode rows: SwiGLU fused into the weight-streaming GEMM
            GemmArgs gg{};
            gg.A = m->h; gg.lda = hid; gg.W = TP(m, p + "gu.w"); gg.C = m->act; gg.ldc = c.ffn; gg.M = M; gg.N = c.ffn; gg.K = hid;
            gg.ab_dtype = dt; gg.c_dtype = dt; gg.act = S2S_ACT_NONE; gg.glu = true;
            if ((rc = gemm_ex(st, gg))) return rc;
        } else {
            if ((rc = gemm(st, m->h, hid, TP(m, p + "gu.w"), nullptr, nullptr, m->gu, 2 * c.ffn, M, 2 * c.ffn, hid, dt, dt, S2S_ACT_NONE, false))) return rc;
            if ((rc = swiglu(st, m->gu, m->act, M, c.ffn, dt))) return rc;
        }
        if ((rc = gemm(st, m->act, c.ffn, TP(m, p + "down.w"), nullptr, x, x, hid, M, hid, c.ffn, dt, S2S_F32, S2S_ACT_NONE, false))) return rc;
    }
    return S2S_OK;
}

__global__ void fill_int_kernel_tts(int* p, int v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
#define fill_int_kernel fill_int_kernel_tts

// kv_len[b] = pos0[b] + T  (valid keys after appending)
__global__ void kvlen_kernel(const int* pos0, int T, int* out, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) out[b] = (pos0 ? pos0[b] : 0) + T;
}

// copy the last valid prompt row of every session: dst[b] = x[b*Tp + len[b]-1]
__global__ void last_row_kernel(const float* __restrict__ x, const int* __restrict__ len, int Tp, int dim, float* __restrict__ dst) {
    const int b = blockIdx.x;
    const float* r = x + ((long)b * Tp + max(len[b] - 1, 0)) * dim;
    for (int i = threadIdx.x; i < dim; i += blockDim.x) dst[(long)b * dim + i] = r[i];
}

// talker head + greedy pick of code 0 + everything that only depends on it (see talker_pick_kernel)
static int talker_head(Tts* m, hipStream_t st, int B) {
    const auto& c = m->cfg;
    const int dt = c.dtype, E = c.t_hidden;
    int rc;
    if ((rc = rmsnorm_ex(st, m->x, E, TPF(m, "talker.norm"), m->hn_t, m->hn, B, E, c.eps, dt))) return rc;
    if ((rc = gemm(st, m->hn_t, E, TP(m, "talker.head.w"), nullptr, nullptr, m->logits, c.codec_vocab, B, c.codec_vocab, E, dt, S2S_F32, S2S_ACT_NONE, false))) return rc;
    // x's hidden rows are consumed (hn); the pick writes the next talker input over them
    return talker_pick(st, m->logits, c.codec_vocab, m->sup_mask, (const unsigned char*)TP(m, "talker.eos_mask"), c.codec_eos, m->frame_b,
                       m->min_b, m->utt_b, m->code0, m->finished, m->n_frames, m->codes_out, c.max_frames, c.n_groups, TP(m, "talker.emb"),
                       TP(m, "talker.pad_embed"), m->hn, m->x, m->pin, m->pin_t, E, dec_fusion_ok(m, m->pred, 2 * B) && c.p_hidden != E, B, dt, m->dbg, m->sp_talker,
                       m->trail, m->n_trail, c.max_prompt);
}

// run the code predictor for the current frame; the residual groups' embeddings accumulate into the next talker input (m->x)
static int predictor_frame(Tts* m, hipStream_t st, int B) {
    const auto& c = m->cfg;
    const int dt = c.dtype, E = c.t_hidden, P = c.p_hidden, G = c.n_groups, CB = c.codebook;
    int rc;
    for (int g = 0; g < G - 1; ++g) {
        const int T_len = g == 0 ? 2 : 1;     // prefill rows [normed talker hidden, embedding of code 0], then one row per group
        const int M = B * T_len;
        const bool dec = dec_fusion_ok(m, m->pred, M);
        if (P != E) {
            if (dec) {   // in_proj starts the predictor's residual rows and emits their bf16 copy + sums of squares in its epilogue
                DecGemm ip{};
                ip.ws = m->dec_ws; ip.ws_bytes = DEC_WS_BYTES;
                ip.A = (const __hip_bfloat16*)m->pin_t; ip.a_frag = true; ip.W = (const __hip_bfloat16*)m->in_proj_f; ip.bias = TPF(m, "pred.in_proj.b");
                ip.M = M; ip.N = P; ip.K = E; ip.mode = 2; ip.x_overwrite = true; ip.X = m->xp; ip.ldx = P; ip.XB = (__hip_bfloat16*)m->xb; ip.ldxb = P; ip.xb_frag = true;
                ip.ss_out = m->ssA; ip.ss_stride = DEC_ROWS; ip.act = S2S_ACT_NONE; ip.w_cached = true;
                if ((rc = gemm_dec(st, ip))) return rc;
                m->ss_parts_A = dec_out_parts(ip);
            } else {
                if ((rc = gemm(st, m->pin_t, E, TP(m, "pred.in_proj.w"), TPF(m, "pred.in_proj.b"), nullptr, m->xp, P, M, P, E, dt, S2S_F32, S2S_ACT_NONE, false))) return rc;
            }
        } else {
            if ((rc = cast_f32(st, m->pin, m->xp, (long)M * E, S2S_F32))) return rc;
            if (dec) {
                if ((rc = rows_prep(st, m->xp, P, m->xb, P, m->ssA, M, P, true))) return rc;
                m->ss_parts_A = 1;
            }
        }
        // cached positions before this pass are a host constant: 0, then g + 1 (row g of the constant table)
        const int* pos = m->pos_p + (size_t)g * m->max_batch;
        if (dec) {
            if ((rc = stack_forward_dec(m, st, m->pred, m->xp, B, T_len, pos, g + 2))) return rc;
            // pred.norm (folded into the head's columns, scale from the down-proj partials) -> group head over all M rows; the
            // pick reads the last row of every session (row b * T_len + T_len - 1)
            DecGemm hg{};
            hg.ws = m->dec_ws; hg.ws_bytes = DEC_WS_BYTES;
            hg.A = (const __hip_bfloat16*)m->xb; hg.a_frag = true; hg.W = (const __hip_bfloat16*)m->head_n[g];
            hg.M = M; hg.N = CB; hg.K = P;
            hg.ss_in = m->ssA; hg.n_parts_in = m->ss_parts_A; hg.ss_stride = DEC_ROWS; hg.inv_dim = 1.0f / (float)P; hg.eps = c.eps;
            hg.mode = 1; hg.C = m->logits; hg.ldc = CB; hg.act = S2S_ACT_NONE; hg.w_cached = true;
            if ((rc = gemm_dec(st, hg))) return rc;
        } else {
            if ((rc = stack_forward(m, st, m->pred, m->xp, B, T_len, pos, g + 2, nullptr))) return rc;
            if ((rc = rmsnorm_ex(st, m->xp + (size_t)(T_len - 1) * P, (long)T_len * P, TPF(m, "pred.norm"), m->h, nullptr, B, P, c.eps, dt))) return rc;
            if ((rc = gemm(st, m->h, P, TP(m, "pred.head." + std::to_string(g) + ".w"), nullptr, nullptr, m->logits, CB, B, CB, P, dt, S2S_F32, S2S_ACT_NONE, false))) return rc;
        }
        // dec: logits hold all M rows (row b * T_len + T_len - 1 is session b's); the next pass's input row is written fragment-major
        const bool next_dec = g + 1 < G - 1 && dec_fusion_ok(m, m->pred, B) && P != E;
        if ((rc = pred_pick(st, dec ? m->logits + (size_t)(T_len - 1) * CB : m->logits, CB, dec ? (long)T_len * CB : (long)CB, m->frame_b, m->utt_b, m->finished, m->codes_out,
                            c.max_frames, G, g + 1, TP(m, "pred.emb." + std::to_string(g)), m->x, m->pin, m->pin_t, E, next_dec, B, dt, m->dbg, m->sp_pred))) return rc;
    }
    return S2S_OK;
}

// One codec frame for the whole batch: talker head -> code 0 -> code predictor (groups 1..G-1) -> talker step on the
// summed embeddings.  Every per-frame quantity (frame index, eos gate, positions) is read from device memory, so the
// launch sequence is identical for every frame and can be replayed from a captured graph.
static int frame_body(Tts* m, hipStream_t st, int B) {
    int rc;
    if ((rc = talker_head(m, st, B))) return rc;
    if ((rc = predictor_frame(m, st, B))) return rc;
    // kv capacity is the (constant) cache length; the causal bound pos_t[b] + 1 limits the keys actually read
    if (dec_fusion_ok(m, m->talker, B)) {
        if ((rc = rows_prep(st, m->x, m->cfg.t_hidden, m->xb, m->cfg.t_hidden, m->ssA, B, m->cfg.t_hidden, true))) return rc;
        m->ss_parts_A = 1;
        if ((rc = stack_forward_dec(m, st, m->talker, m->x, B, 1, m->pos_t, m->talker.c.max_pos))) return rc;
    } else {
        if ((rc = stack_forward(m, st, m->talker, m->x, B, 1, m->pos_t, m->talker.c.max_pos, nullptr))) return rc;
    }
    return tts_frame_end(st, m->pos_t, m->frame_b, m->finished, m->budget_b, m->talker.c.max_pos, B);
}

// captured frame body for batch size B (built on first use; the arena and the weights never move after finalize)
static int frame_graph(Tts* m, int B, hipGraphExec_t* out) {
    {
        std::lock_guard<std::mutex> g(m->graphs_mu);
        auto it = m->frame_graphs.find(B);
        if (it != m->frame_graphs.end()) {
            *out = it->second;
            return S2S_OK;
        }
    }
    std::lock_guard<std::mutex> capture_lock(capture_mutex());
    if (!m->cap_stream) S2S_HIP_CHECK(hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
    S2S_HIP_CHECK(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = frame_body(m, m->cap_stream, B);
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
    {
        std::lock_guard<std::mutex> g(m->graphs_mu);
        m->frame_graphs[B] = exec;
    }
    *out = exec;
    return S2S_OK;
}

}  // namespace s2s

using namespace s2s;

extern "C" int32_t s2s_tts_create(const s2s_tts_config* cfg, int32_t max_batch, void** out) {
    S2S_REQUIRE(cfg && out && max_batch > 0 && max_batch <= MAX_SLOTS, "tts_create: bad arguments (max_batch 1..%d)", MAX_SLOTS);
    S2S_REQUIRE(cfg->dtype == S2S_F32 || cfg->dtype == S2S_BF16, "tts_create: bad dtype");
    S2S_REQUIRE(cfg->t_hidden % 8 == 0 && cfg->p_hidden % 8 == 0 && cfg->c_hidden % 8 == 0 && cfg->t_ffn % 8 == 0 && cfg->p_ffn % 8 == 0 &&
                    cfg->c_ffn % 8 == 0 && cfg->n_dec_blocks >= 1 && cfg->n_dec_blocks <= 4 && (cfg->c_decoder_dim >> cfg->n_dec_blocks) % 8 == 0,
                "tts_create: dimensions must be multiples of 8 (decoder_dim / 2^n_dec_blocks too)");
    S2S_REQUIRE(cfg->max_prompt >= 2 && cfg->max_frames >= 1 && cfg->max_pos >= cfg->max_prompt + cfg->max_frames, "tts_create: max_pos too small");
    Tts* m = new Tts();
    m->cfg = *cfg;
    m->max_batch = max_batch;
    m->esz = cfg->dtype == S2S_F32 ? 4 : 2;
    *out = m;
    return S2S_OK;
}

extern "C" int32_t s2s_tts_destroy(void* model) {
    if (!model) return S2S_OK;
    Tts* m = (Tts*)model;
    guard_free(m->arena);
    if (m->host_flag) (void)hipHostFree(m->host_flag);
    if (m->pos_p) (void)hipFree(m->pos_p);
    for (void* p : m->owned) guard_free(p);
    for (void* p : m->cstate) guard_free(p);
    for (auto& kv : m->frame_graphs) (void)hipGraphExecDestroy(kv.second);
    for (hipGraphExec_t e : m->retired_graphs) (void)hipGraphExecDestroy(e);
    if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
    delete m;
    return S2S_OK;
}

extern "C" int32_t s2s_tts_set_tensor(void* model, const char* name, const void* dev_ptr, int64_t numel) {
    Tts* m = (Tts*)model;
    S2S_REQUIRE(m && name && dev_ptr, "tts_set_tensor: null argument");
    if (m->finalized) return fail(S2S_ERR_STATE, "tts_set_tensor: %s after finalize (captured graphs hold the old pointers)", name);
    S2S_REQUIRE(((uintptr_t)dev_ptr & 15) == 0, "tts_set_tensor: %s is not 16-byte aligned", name);
    m->w[name] = TTensor{dev_ptr, (long)numel};
    return S2S_OK;
}

extern "C" int32_t s2s_tts_finalize(void* model) {
    // workspace allocation uses synchronous (legacy-stream) memsets / copies, which HIP refuses - and which poison the other thread's
    // capture - while any stream of the process is capturing: take the capture lock
    std::lock_guard<std::mutex> no_capture(s2s::capture_mutex());
    Tts* m = (Tts*)model;
    S2S_REQUIRE(m, "tts_finalize: null model");
    if (m->finalized) return S2S_OK;
    int rc = need(m, "talker.eos_mask", m->cfg.codec_vocab);
    if (rc) return rc;
    return finalize(m);
}

extern "C" int64_t s2s_tts_workspace_bytes(void* model) {
    Tts* m = (Tts*)model;
    return m ? (int64_t)m->arena_bytes : 0;
}

// Admission of n sessions into the slots [slot0, slot0 + n) of the frame loop (the others keep their state): talker prefill of the
// n prompts straight into those slots' cache rows, then the slots' bookkeeping (position = prompt length, frame 0, eos gate, budget,
// utterance id) and their first input row (hidden of the last prompt position).  The prefill uses the shared scratch rows m->x, whose
// first max_batch rows are the frame loop's state: they are parked in m->xnext around it.
static int tts_admit_impl(Tts* m, hipStream_t st, const float* prompt_embeds_dev, const int32_t* prompt_len_dev, int n, int Tp, int slot0,
                          const int32_t* min_frames_dev, const int32_t* budget_dev, int min_all, int budget_all, int utt0) {
    const auto& c = m->cfg;
    const int E = c.t_hidden, G = c.n_groups, e = m->esz;
    int rc;
    S2S_HIP_CHECK(hipMemcpyAsync(m->xnext, m->x, sizeof(float) * (size_t)m->max_batch * E, hipMemcpyDeviceToDevice, st));
    S2S_HIP_CHECK(hipMemcpyAsync(m->x, prompt_embeds_dev, sizeof(float) * (size_t)n * Tp * E, hipMemcpyDeviceToDevice, st));
    S2S_HIP_CHECK(hipMemsetAsync(m->codes_out + (size_t)slot0 * c.max_frames * G, 0xff, sizeof(int) * (size_t)n * c.max_frames * G, st));
    S2S_HIP_CHECK(hipMemsetAsync(m->n_trail + slot0, 0, sizeof(int) * (size_t)n, st));      // pre-filled-text mode until s2s_tts_set_trailing says otherwise
    // the same stack, its cache rows shifted to the first admitted slot (row b of the prefill batch = slot slot0 + b)
    Stack pre = m->talker;
    const size_t slot_bytes = (size_t)pre.c.max_pos * pre.c.kv_heads * pre.c.head_dim * e;
    pre.kc = at(m->talker.kc, slot0 * slot_bytes);
    pre.vc = at(m->talker.vc, slot0 * slot_bytes);
    // padded rows beyond a session's prompt are cached too but never attended: kv_len = prompt_len
    if ((rc = stack_forward(m, st, pre, m->x, n, Tp, nullptr, Tp, prompt_len_dev))) return rc;
    last_row_kernel<<<n, 256, 0, st>>>(m->x, prompt_len_dev, Tp, E, m->xnext + (size_t)slot0 * E);
    S2S_LAUNCH_CHECK();
    // the frame body always reads the hidden of the last processed position of slot b from x[b]
    S2S_HIP_CHECK(hipMemcpyAsync(m->x, m->xnext, sizeof(float) * (size_t)m->max_batch * E, hipMemcpyDeviceToDevice, st));
    return tts_slots_set(st, m->pos_t, m->frame_b, m->min_b, m->budget_b, m->utt_b, m->finished, m->n_frames, m->code0, prompt_len_dev, min_frames_dev,
                         budget_dev, min_all, budget_all, utt0, c.max_frames, slot0, n, true);
}

// Prefill: prompt_embeds_dev f32 [B][Tp][E] (right-padded), prompt_len_dev int32 [B].  Starts a fresh batch in slots 0 .. B-1
// (every other slot idle).
extern "C" int32_t s2s_tts_prefill(void* model, void* stream, const float* prompt_embeds_dev, const int32_t* prompt_len_dev,
                                   int32_t B, int32_t Tp) {
    Tts* m = (Tts*)model;
    S2S_REQUIRE(m && m->finalized, "tts_prefill: model not finalized");
    S2S_REQUIRE(B > 0 && B <= m->max_batch && Tp >= 1 && Tp <= m->cfg.max_prompt && prompt_embeds_dev && prompt_len_dev, "tts_prefill: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = tts_slots_set(st, m->pos_t, m->frame_b, m->min_b, m->budget_b, m->utt_b, m->finished, m->n_frames, m->code0, nullptr, nullptr, nullptr, 0, 0, 0,
                            m->cfg.max_frames, 0, m->max_batch, false))) return rc;
    if ((rc = tts_admit_impl(m, st, prompt_embeds_dev, prompt_len_dev, B, Tp, 0, nullptr, nullptr, 0, m->cfg.max_frames, m->next_utt))) return rc;
    m->next_utt += B;
    m->cur_B = B;
    m->prompt_max = Tp;
    m->frames_done = 0;
    return S2S_OK;
}

/* Continuous batching (the reference streams every session independently, TTS/qwen3_tts_handler.py:695-749; here sessions share the
 * launches of one frame loop and join / leave it at chunk boundaries):
 *   s2s_tts_admit    n prompts -> slots [slot0, slot0 + n) while the other slots keep generating.  min_frames_dev / budget_dev:
 *                    optional int32[n] per session (else min_frames / max_new_frames for all); a slot retires itself after
 *                    budget frames or at eos.  utterance_id0 + i keys session i's sampler stream.
 *   s2s_tts_step     n_frames frames for the slots [0, rows): every slot advances by its OWN frame index.
 *   s2s_tts_read_codes   codes of frames [frame0, frame0 + n_frames) of slots [slot0, slot0 + n) -> int32 [n][n_frames][G] (-1 after
 *                    a session's end), valid frames so far per slot -> n_frames_dev int32[n] (optional).
 *   s2s_tts_release  slots [slot0, slot0 + n) become idle (their rows are still computed, nothing is emitted). */
extern "C" int32_t s2s_tts_admit(void* model, void* stream, const float* prompt_embeds_dev, const int32_t* prompt_len_dev, int32_t n, int32_t Tp,
                                 int32_t slot0, const int32_t* min_frames_dev, const int32_t* budget_dev, int32_t min_frames, int32_t max_new_frames,
                                 int32_t utterance_id0) {
    Tts* m = (Tts*)model;
    S2S_REQUIRE(m && m->finalized, "tts_admit: model not finalized");
    S2S_REQUIRE(n > 0 && slot0 >= 0 && slot0 + n <= m->max_batch && Tp >= 1 && Tp <= m->cfg.max_prompt && prompt_embeds_dev && prompt_len_dev,
                "tts_admit: bad arguments (slots %d..%d of %d, Tp %d)", slot0, slot0 + n, m->max_batch, Tp);
    S2S_REQUIRE(max_new_frames >= 1 && min_frames >= 0, "tts_admit: bad frame limits");
    m->cur_B = std::max(m->cur_B, slot0 + n);
    return tts_admit_impl(m, (hipStream_t)stream, prompt_embeds_dev, prompt_len_dev, n, Tp, slot0, min_frames_dev, budget_dev, min_frames, max_new_frames,
                          utterance_id0);
}

/* Streaming-text mode of the sessions just admitted into the slots [slot0, slot0 + n) (reference `non_streaming_mode = False`,
 * arguments_classes/qwen3_tts_arguments.py:103-108; sibling modeling_qwen3_omni_moe.py:3744-3801,3169-3174): the prompt that was pre-filled ends
 * with the FIRST text token, the remaining text rows (and the text-side eos row) arrive one per generated frame - the input of the step that
 * produces frame f + 1 is the sum of frame f's code embeddings + trailing row f, and pad_embed once the rows are used up.
 * rows_dev f32 [n][Tr][E] (right-padded), n_rows_dev int32[n] (<= Tr <= max_prompt).  Call after s2s_tts_admit / s2s_tts_prefill (which reset
 * the slots to the pre-filled-text mode) and before the next s2s_tts_step. */
extern "C" int32_t s2s_tts_set_trailing(void* model, void* stream, int32_t slot0, int32_t n, const float* rows_dev, const int32_t* n_rows_dev, int32_t Tr) {
    Tts* m = (Tts*)model;
    S2S_REQUIRE(m && m->finalized, "tts_set_trailing: model not finalized");
    S2S_REQUIRE(n > 0 && slot0 >= 0 && slot0 + n <= m->max_batch && rows_dev && n_rows_dev && Tr >= 1 && Tr <= m->cfg.max_prompt,
                "tts_set_trailing: bad arguments (slots %d..%d of %d, %d rows, at most %d)", slot0, slot0 + n, m->max_batch, Tr, m->cfg.max_prompt);
    return trail_set((hipStream_t)stream, rows_dev, n_rows_dev, n, Tr, m->cfg.t_hidden, m->trail, m->n_trail, m->cfg.max_prompt, slot0, m->cfg.dtype);
}

/* Move n live sessions from the slots [src_slot0, ..) of ONE engine into the slots [dst_slot0, ..) of ANOTHER engine of the same model
 * (replicas sharing the weights; same max_prompt / max_frames): everything a slot owns - the talker cache rows of its first n_pos
 * positions, its next input row, frame index / eos gate / budget / sampler key, the codes emitted so far, the codec decoder's history rows
 * and the first n_codec_pos positions of its pre-transformer cache.  Asynchronous copies on `stream`, which must already be ordered behind
 * the source engine's last launch (event); the source slots may be re-admitted once these copies have completed.
 * Why: a session's FIRST chunk is latency-critical and runs in a small, eager frame loop (a "nursery" engine); from its second chunk on it
 * only has to stay ahead of its playback and continues in the wide, paced loop (speech-to-speech_amd/tts_continuous.py). */
extern "C" int32_t s2s_tts_migrate(void* dst_model, void* src_model, void* stream, int32_t src_slot0, int32_t dst_slot0, int32_t n, int32_t n_pos,
                                   int32_t n_codec_pos, const float* x_rows_dev) {
    Tts *d = (Tts*)dst_model, *s = (Tts*)src_model;
    S2S_REQUIRE(d && s && d != s && d->finalized && s->finalized, "tts_migrate: two finalized engines");
    const auto &a = d->cfg, &b = s->cfg;
    S2S_REQUIRE(a.dtype == b.dtype && a.t_hidden == b.t_hidden && a.t_layers == b.t_layers && a.t_kv_heads == b.t_kv_heads && a.t_head_dim == b.t_head_dim &&
                    a.max_pos == b.max_pos && a.max_prompt == b.max_prompt && a.max_frames == b.max_frames && a.n_groups == b.n_groups && a.c_hidden == b.c_hidden && a.c_layers == b.c_layers &&
                    a.c_heads == b.c_heads && a.c_decoder_dim == b.c_decoder_dim && a.n_dec_blocks == b.n_dec_blocks && d->codec.c.max_pos == s->codec.c.max_pos,
                "tts_migrate: the engines must be replicas of one model with the same max_prompt / max_frames");
    S2S_REQUIRE(n > 0 && src_slot0 >= 0 && src_slot0 + n <= s->max_batch && dst_slot0 >= 0 && dst_slot0 + n <= d->max_batch && n_pos >= 0 && n_pos <= a.max_pos &&
                    n_codec_pos >= 0 && n_codec_pos <= d->codec.c.max_pos, "tts_migrate: bad slot range / positions");
    hipStream_t st = (hipStream_t)stream;
    const size_t e = d->esz;
    auto copy_cache = [&](const Stack& ss, const Stack& ds, int npos) -> int {
        const size_t row = (size_t)ss.c.kv_heads * ss.c.head_dim * e, slot = (size_t)ss.c.max_pos * row;
        if (npos <= 0) return S2S_OK;
        for (int l = 0; l < ss.c.layers; ++l)
            for (int kv = 0; kv < 2; ++kv) {
                const char* sp = (const char*)(kv ? ss.vc : ss.kc) + ((size_t)l * s->max_batch + src_slot0) * slot;
                char* dp = (char*)(kv ? ds.vc : ds.kc) + ((size_t)l * d->max_batch + dst_slot0) * slot;
                S2S_HIP_CHECK(hipMemcpy2DAsync(dp, slot, sp, slot, (size_t)npos * row, n, hipMemcpyDeviceToDevice, st));
            }
        return S2S_OK;
    };
    int rc;
    if ((rc = copy_cache(s->talker, d->talker, n_pos))) return rc;
    if ((rc = copy_cache(s->codec, d->codec, n_codec_pos))) return rc;
    for (size_t k = 0; k < s->cstate.size() && k < s->cstate_c.size(); ++k) {
        const size_t rowb = (size_t)LPAD * s->cstate_c[k] * e;
        S2S_HIP_CHECK(hipMemcpyAsync((char*)d->cstate[k] + dst_slot0 * rowb, (const char*)s->cstate[k] + src_slot0 * rowb, rowb * n, hipMemcpyDeviceToDevice, st));
    }
    const size_t E = a.t_hidden;
    // the next input rows: a snapshot taken right behind the sessions' last frame (s2s_tts_read_x) when the source loop has gone on since -
    // a parked slot's row of x is recomputed (as garbage) by every later frame of the source loop; everything else a parked slot owns stands still
    S2S_HIP_CHECK(hipMemcpyAsync(d->x + dst_slot0 * E, x_rows_dev ? x_rows_dev : s->x + src_slot0 * E, sizeof(float) * E * n, hipMemcpyDeviceToDevice, st));
    int* const dsti[] = {d->pos_t, d->frame_b, d->min_b, d->budget_b, d->utt_b, d->n_frames, d->code0};
    int* const srci[] = {s->pos_t, s->frame_b, s->min_b, s->budget_b, s->utt_b, s->n_frames, s->code0};
    for (int k = 0; k < 7; ++k) S2S_HIP_CHECK(hipMemcpyAsync(dsti[k] + dst_slot0, srci[k] + src_slot0, sizeof(int) * n, hipMemcpyDeviceToDevice, st));
    // streaming-text rows still to be consumed travel with the session (both engines: same max_prompt)
    S2S_HIP_CHECK(hipMemcpyAsync(d->n_trail + dst_slot0, s->n_trail + src_slot0, sizeof(int) * n, hipMemcpyDeviceToDevice, st));
    {
        const size_t tb = (size_t)a.max_prompt * E * e;
        S2S_HIP_CHECK(hipMemcpyAsync((char*)d->trail + dst_slot0 * tb, (const char*)s->trail + src_slot0 * tb, tb * n, hipMemcpyDeviceToDevice, st));
    }
    S2S_HIP_CHECK(hipMemsetAsync(d->finished + dst_slot0, 0, sizeof(int) * n, st));       // live in the destination (the source parks its slots as finished)
    const size_t cw = (size_t)a.max_frames * a.n_groups;
    S2S_HIP_CHECK(hipMemcpyAsync(d->codes_out + dst_slot0 * cw, s->codes_out + src_slot0 * cw, sizeof(int) * cw * n, hipMemcpyDeviceToDevice, st));
    d->cur_B = std::max(d->cur_B, dst_slot0 + n);
    return S2S_OK;
}

/* The next talker input rows of the slots [slot0, slot0 + n) -> x_out_dev f32 [n][t_hidden] (a snapshot for s2s_tts_migrate). */
extern "C" int32_t s2s_tts_read_x(void* model, void* stream, int32_t slot0, int32_t n, float* x_out_dev) {
    Tts* m = (Tts*)model;
    S2S_REQUIRE(m && m->finalized && slot0 >= 0 && n > 0 && slot0 + n <= m->max_batch && x_out_dev, "tts_read_x: bad arguments");
    S2S_HIP_CHECK(hipMemcpyAsync(x_out_dev, m->x + (size_t)slot0 * m->cfg.t_hidden, sizeof(float) * (size_t)n * m->cfg.t_hidden, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return S2S_OK;
}

extern "C" int32_t s2s_tts_release(void* model, void* stream, int32_t slot0, int32_t n) {
    Tts* m = (Tts*)model;
    S2S_REQUIRE(m && m->finalized && slot0 >= 0 && n >= 0 && slot0 + n <= m->max_batch, "tts_release: bad arguments");
    return tts_slots_set((hipStream_t)stream, m->pos_t, m->frame_b, m->min_b, m->budget_b, m->utt_b, m->finished, m->n_frames, m->code0, nullptr, nullptr,
                         nullptr, 0, 0, 0, m->cfg.max_frames, slot0, n, false);
}

// n_frames passes of the frame body over the slots [0, rows): a captured graph per row count, replayed per frame
static int tts_step_impl(Tts* m, hipStream_t st, int rows, int n_frames, bool allow_graph) {
    int rc;
    hipGraphExec_t exec = nullptr;
    if (allow_graph && graphs_enabled() && !prof_active() && frame_graph(m, rows, &exec) != S2S_OK) {
        // the capture was refused or invalidated (HIP aborts a capture when another thread touches the legacy stream): not fatal -
        // clear the sticky error and run this call as plain stream launches; the next call captures again
        (void)hipGetLastError();
        exec = nullptr;
    }
    if (exec) {
        for (int f = 0; f < n_frames; ++f) S2S_HIP_CHECK(hipGraphLaunch(exec, st));
    } else {
        for (int f = 0; f < n_frames; ++f)
            if ((rc = frame_body(m, st, rows))) return rc;
    }
    return S2S_OK;
}

extern "C" int32_t s2s_tts_step(void* model, void* stream, int32_t rows, int32_t n_frames) {
    Tts* m = (Tts*)model;
    S2S_REQUIRE(m && m->finalized && rows >= 1 && rows <= m->max_batch && n_frames >= 1, "tts_step: bad arguments");
    return tts_step_impl(m, (hipStream_t)stream, rows, n_frames, true);
}

extern "C" int32_t s2s_tts_read_codes(void* model, void* stream, int32_t slot0, int32_t n, int32_t frame0, int32_t n_frames, int32_t* codes_dev,
                                      int32_t* n_frames_dev, int32_t chunk_samples, int32_t* valid_samples_dev) {
    Tts* m = (Tts*)model;
    S2S_REQUIRE(m && m->finalized && slot0 >= 0 && n >= 1 && slot0 + n <= m->max_batch && frame0 >= 0 && n_frames >= 1 && codes_dev, "tts_read_codes: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = tts_read_codes(st, m->codes_out, m->cfg.max_frames, m->cfg.n_groups, slot0, n, frame0, nullptr, n_frames, codes_dev))) return rc;
    if (n_frames_dev) S2S_HIP_CHECK(hipMemcpyAsync(n_frames_dev, m->n_frames + slot0, sizeof(int) * n, hipMemcpyDeviceToDevice, st));
    if (valid_samples_dev && (rc = tts_valid_samples(st, m->n_frames, slot0, n, frame0, nullptr, n_frames, m->cfg.total_upsample, chunk_samples, valid_samples_dev))) return rc;
    return S2S_OK;
}

/* The same for rows of DIFFERENT ages in one call: frame0_rows_dev int32[n] = first frame to read per slot (sessions admitted at different
 * chunk boundaries share one codec pass, s2s_codec_decode_rows_at).  chunk_samples = n_frames * total_upsample (continuation chunks). */
extern "C" int32_t s2s_tts_read_codes_rows(void* model, void* stream, int32_t slot0, int32_t n, const int32_t* frame0_rows_dev, int32_t n_frames,
                                           int32_t* codes_dev, int32_t* n_frames_dev, int32_t chunk_samples, int32_t* valid_samples_dev) {
    Tts* m = (Tts*)model;
    S2S_REQUIRE(m && m->finalized && slot0 >= 0 && n >= 1 && slot0 + n <= m->max_batch && frame0_rows_dev && n_frames >= 1 && codes_dev, "tts_read_codes_rows: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = tts_read_codes(st, m->codes_out, m->cfg.max_frames, m->cfg.n_groups, slot0, n, 0, frame0_rows_dev, n_frames, codes_dev))) return rc;
    if (n_frames_dev) S2S_HIP_CHECK(hipMemcpyAsync(n_frames_dev, m->n_frames + slot0, sizeof(int) * n, hipMemcpyDeviceToDevice, st));
    if (valid_samples_dev && (rc = tts_valid_samples(st, m->n_frames, slot0, n, 0, frame0_rows_dev, n_frames, m->cfg.total_upsample, chunk_samples, valid_samples_dev))) return rc;
    return S2S_OK;
}

// Generate up to n_frames more codec frames for the prefilled batch (greedy).  Returns codes of the frames
// produced by THIS call in codes_dev int32 [B][n_frames][G] (-1 after a session's end) and the total number of
// valid frames so far per session in n_frames_dev.  min_total_frames: eos is suppressed until a session has that
// many frames.  all_done_host (optional): set to 1 when every session has emitted eos.
static int32_t tts_generate_impl(Tts* m, hipStream_t st, int32_t n_frames, int32_t min_total_frames, int32_t* codes_dev, int32_t* n_frames_dev,
                                 int32_t* all_done_host, bool allow_graph) {
    S2S_REQUIRE(m && m->finalized && m->cur_B > 0, "tts_generate: prefill first");
    const auto& c = m->cfg;
    S2S_REQUIRE(n_frames >= 1 && m->frames_done + n_frames <= c.max_frames, "tts_generate: frame budget %d exceeded", c.max_frames);
    const int B = m->cur_B, G = c.n_groups;
    int rc;
    const int f0 = m->frames_done;
    fill_int_kernel<<<cdiv(B, 64), 64, 0, st>>>(m->min_b, min_total_frames, B);
    S2S_LAUNCH_CHECK();
    if ((rc = tts_step_impl(m, st, B, n_frames, allow_graph))) return rc;
    m->frames_done = f0 + n_frames;
    if (codes_dev)
        S2S_HIP_CHECK(hipMemcpy2DAsync(codes_dev, sizeof(int) * n_frames * G, m->codes_out + (size_t)f0 * G, sizeof(int) * c.max_frames * G,
                                       sizeof(int) * n_frames * G, B, hipMemcpyDeviceToDevice, st));
    if (n_frames_dev) S2S_HIP_CHECK(hipMemcpyAsync(n_frames_dev, m->n_frames, sizeof(int) * B, hipMemcpyDeviceToDevice, st));
    if (all_done_host) {
        S2S_HIP_CHECK(hipMemcpyAsync(m->host_flag, m->finished, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        S2S_HIP_CHECK(hipStreamSynchronize(st));
        int all = 1;
        for (int b = 0; b < B; ++b) all &= m->host_flag[b] != 0;
        *all_done_host = all;
    }
    return S2S_OK;
}

// Sampling of the frame loop.  which: 0 = talker (code 0; repetition penalty over the utterance's earlier code-0 picks), 1 = code
// predictor (residual groups).  temperature <= 0 -> greedy argmax (default, the parity policy).  The captured frame graphs hold the
// parameters by value, so they are dropped and re-captured on the next generate call.
extern "C" int32_t s2s_tts_set_sampling(void* model, int32_t which, float temperature, int32_t top_k, float top_p, float repetition_penalty,
                                        uint64_t seed) {
    Tts* m = (Tts*)model;
    S2S_REQUIRE(m && (which == 0 || which == 1), "tts_set_sampling: bad arguments");
    S2S_REQUIRE(top_k >= 0 && top_p > 0.f && repetition_penalty > 0.f, "tts_set_sampling: top_k >= 0, top_p > 0, repetition_penalty > 0");
    SampleParams sp{temperature, top_k, top_p, repetition_penalty, (unsigned long long)seed};
    (which == 0 ? m->sp_talker : m->sp_pred) = sp;
    // the captured graphs hold the old parameters by value: they are RETIRED, not destroyed - a lane's stream may still be replaying
    // one (destroying it there would be a use-after-free); the next generate / step call captures again
    std::lock_guard<std::mutex> capture_lock(capture_mutex());
    std::lock_guard<std::mutex> g(m->graphs_mu);
    for (auto& kv : m->frame_graphs) m->retired_graphs.push_back(kv.second);
    m->frame_graphs.clear();
    return S2S_OK;
}

extern "C" int32_t s2s_tts_generate(void* model, void* stream, int32_t n_frames, int32_t min_total_frames, int32_t* codes_dev,
                                    int32_t* n_frames_dev, int32_t* all_done_host) {
    return tts_generate_impl((Tts*)model, (hipStream_t)stream, n_frames, min_total_frames, codes_dev, n_frames_dev, all_done_host, true);
}

// Teacher-forced continuation (in-context continuation of given codec frames; parity tests): like s2s_tts_generate, but every
// code of the n_frames new frames whose entry in forced_codes_dev [B][n_frames][G] is >= 0 is emitted (and fed back) instead of
// the greedy pick, and the raw logits of every pick are copied to logits_dev f32 [B][n_frames][G][vlog] (optional; group 0 rows
// hold codec_vocab values, the others codebook values; vlog >= max of both).  Runs as plain stream launches (no graph replay).
extern "C" int32_t s2s_tts_generate_forced(void* model, void* stream, int32_t n_frames, int32_t min_total_frames, const int32_t* forced_codes_dev,
                                           float* logits_dev, int32_t vlog, int32_t* codes_dev, int32_t* n_frames_dev) {
    Tts* m = (Tts*)model;
    S2S_REQUIRE(m && m->finalized && m->cur_B > 0, "tts_generate_forced: prefill first");
    const auto& c = m->cfg;
    S2S_REQUIRE(n_frames >= 1 && m->frames_done + n_frames <= c.max_frames, "tts_generate_forced: frame budget %d exceeded", c.max_frames);
    S2S_REQUIRE(!logits_dev || vlog >= std::max(c.codec_vocab, c.codebook), "tts_generate_forced: vlog %d too small", vlog);
    hipStream_t st = (hipStream_t)stream;
    const int B = m->cur_B, G = c.n_groups;
    if (forced_codes_dev) {
        if (!m->forced_buf) {
            S2S_HIP_CHECK(hipMalloc((void**)&m->forced_buf, sizeof(int) * (size_t)m->max_batch * c.max_frames * G));
            m->owned.push_back(m->forced_buf);
        }
        S2S_HIP_CHECK(hipMemsetAsync(m->forced_buf, 0xff, sizeof(int) * (size_t)m->max_batch * c.max_frames * G, st));
        S2S_HIP_CHECK(hipMemcpy2DAsync(m->forced_buf + (size_t)m->frames_done * G, sizeof(int) * c.max_frames * G, forced_codes_dev, sizeof(int) * n_frames * G,
                                       sizeof(int) * n_frames * G, B, hipMemcpyDeviceToDevice, st));
    }
    m->dbg = PickDebug{forced_codes_dev ? m->forced_buf : nullptr, logits_dev, m->frames_done, n_frames, vlog};
    const int32_t rc = tts_generate_impl(m, st, n_frames, min_total_frames, codes_dev, n_frames_dev, nullptr, false);
    m->dbg = PickDebug{};
    return rc;
}

// ---- codec decoder ----------------------------------------------------------------------------------------
namespace s2s {

// causal conv (kernel k, dilation d) as one GEMM over the padded time-major buffer: in [B][LPAD+T][Cin] -> out rows
static int conv_gemm(Tts* m, hipStream_t st, const void* in, int T, int Cin, int k, int dil, const std::string& w, const std::string& b,
                     void* out, int out_lpad, int Cout, const float* residual_in_out_layout, int act, int c_dtype) {
    GemmArgs g{};
    const int e = m->esz;
    g.A = at(in, (size_t)(LPAD - (k - 1) * dil) * Cin * e); g.lda = Cin; g.a_bs = (long)(LPAD + T) * Cin;
    g.W = TP(m, w); g.bias = TPF(m, b);
    const size_t ce = c_dtype == S2S_F32 ? 4 : e;
    g.C = at(out, (size_t)out_lpad * Cout * ce); g.ldc = Cout; g.c_bs = (long)(out_lpad + T) * Cout;
    g.residual = residual_in_out_layout; g.ldr = Cout; g.r_bs = g.c_bs;
    g.M = m->cur_B * T; g.N = Cout; g.K = k * Cin; g.G = T;
    g.ab_dtype = m->cfg.dtype; g.c_dtype = c_dtype; g.act = act;
    if (k > 1) { g.tap_len = Cin; g.tap_stride = (long)dil * Cin; }
    return gemm_ex(st, g);
}

}  // namespace s2s

namespace s2s {

// Codec decoder over F frames of codes.  streaming = false: stateless decode of the window, the first ctx frames
// are left context whose samples are dropped (the sibling's chunked_decode pattern).  streaming = true: exact causal
// continuation of the previous call - every convolution's history rows and the transformer's KV cache are carried
// over, so the concatenated chunks equal one decode of the whole utterance and nothing is recomputed.
// row0 / frames_done_io: the streaming state (history rows, pre-transformer cache) of the batch rows lives in rows [row0, row0 + B) of the
// engine's codec state and the stream has *frames_done_io frames behind it (updated) - cohorts of sessions admitted at different
// times stream independently from disjoint row ranges (continuous batching); null: the engine-wide stream of the legacy entry points.
// pos_rows_dev (optional, int32[B] on the device; streaming continuation only): frames already decoded PER ROW - rows of different ages
// share the pass (every stage's history rows and the pre-transformer cache are per row anyway; only the positions differ), and
// *frames_done_io is then the largest of them (the host's bound for the cache length).
static int codec_run(Tts* m, hipStream_t st, const int* codes_dev, int B, int F, int ctx, bool streaming, float* wav_dev, int n_samples, int row0 = 0,
                     int* frames_done_io = nullptr, const int* pos_rows_dev = nullptr) {
    const auto& c = m->cfg;
    const int dt = c.dtype, e = m->esz, H = c.c_hidden, G = c.n_groups;
    int& frames_done = frames_done_io ? *frames_done_io : m->codec_frames_done;
    const bool first = !streaming || (frames_done == 0 && !pos_rows_dev);
    const int saveB = m->cur_B;
    m->cur_B = B;
    int rc;
    int slot = 0;
    auto done = [&](int r) { m->cur_B = saveB; return r; };
    // Buffers are reused with a different [T][C] geometry at every stage.  Before a stage's input is produced its
    // LPAD history rows are (re)established: zeros (stateless / first chunk) or the tail of the previous chunk.
    auto hist_load = [&](void* buf, int Tn, int Cn) -> int {
        const int sl = slot++;
        hipError_t er;
        if (first) er = hipMemset2DAsync(buf, (size_t)(LPAD + Tn) * Cn * e, 0, (size_t)LPAD * Cn * e, B, st);
        else er = hipMemcpy2DAsync(buf, (size_t)(LPAD + Tn) * Cn * e, at(m->cstate[sl], (size_t)row0 * LPAD * Cn * e), (size_t)LPAD * Cn * e, (size_t)LPAD * Cn * e, B, hipMemcpyDeviceToDevice, st);
        return er == hipSuccess ? sl : fail(S2S_ERR_HIP, "codec history load: %s", hipGetErrorString(er));
    };
    auto hist_save = [&](void* buf, int Tn, int Cn, int sl) -> int {
        if (!streaming) return S2S_OK;
        hipError_t er = hipMemcpy2DAsync(at(m->cstate[sl], (size_t)row0 * LPAD * Cn * e), (size_t)LPAD * Cn * e, at(buf, (size_t)Tn * Cn * e), (size_t)(LPAD + Tn) * Cn * e,
                                         (size_t)LPAD * Cn * e, B, hipMemcpyDeviceToDevice, st);
        return er == hipSuccess ? S2S_OK : fail(S2S_ERR_HIP, "codec history save: %s", hipGetErrorString(er));
    };
#define HLOAD(var, buf, Tn, Cn) const int var = hist_load(buf, Tn, Cn); if (var < 0) return done(var)
    static const bool dbg_sync = getenv("S2S_CODEC_DEBUG_SYNC") != nullptr;      // localise a faulting stage: sync + trace after every stage
    auto stage_ok = [&](const char* what, int i, int j) {
        if (!dbg_sync) return;
        const hipError_t er = hipStreamSynchronize(st);
        fprintf(stderr, "[codec] %s %d.%d: %s\n", what, i, j, hipGetErrorString(er));
        fflush(stderr);
    };
    {   // the widest stage of THIS call's geometry must fit the activation buffers (a continuation chunk is longer than a first chunk)
        size_t need_e = 0;
        for (int i = 0; i <= c.n_dec_blocks; ++i) {
            const long Ti = first ? codec_len(c, F, i) : codec_len_cont(c, F, i);
            need_e = std::max<size_t>(need_e, (size_t)(LPAD + Ti) * (size_t)(i == 0 ? std::max<long>(4 * H, c.c_decoder_dim) : (c.c_decoder_dim >> i)));
        }
        if (need_e > m->act_elems) return done(fail(S2S_ERR_ARG, "codec: %d frames need %zu activation elements per session, the engine holds %zu (max_codec_frames %d)", F, need_e,
                                                   m->act_elems, c.max_codec_frames));
    }
    // code embeddings -> sliding-window transformer (KV cache continues across streamed chunks)
    if ((rc = code_embed_mean(st, TP(m, "codec.emb"), codes_dev, m->cx, (long)B * F, G, c.codebook, H, dt))) return done(rc);
    const int f0 = streaming ? frames_done : 0;
    if (streaming && pos_rows_dev) {
        S2S_HIP_CHECK(hipMemcpyAsync(m->pos_c, pos_rows_dev, sizeof(int) * B, hipMemcpyDeviceToDevice, st));
    } else if (streaming) {
        fill_int_kernel<<<cdiv(B, 64), 64, 0, st>>>(m->pos_c, f0, B);
        S2S_LAUNCH_CHECK();
    }
    Stack cs = m->codec;         // the pre-transformer's cache rows of this row range
    {
        const size_t row_bytes = (size_t)cs.c.max_pos * cs.c.kv_heads * cs.c.head_dim * e;
        cs.kc = at(m->codec.kc, row0 * row_bytes);
        cs.vc = at(m->codec.vc, row0 * row_bytes);
    }
    if (dec_fusion_ok(m, cs, B * F)) {
        if ((rc = rows_prep(st, m->cx, H, m->xb, H, m->ssA, B * F, H, true))) return done(rc);
        m->ss_parts_A = 1;
        if ((rc = stack_forward_dec(m, st, cs, m->cx, B, F, streaming ? m->pos_c : nullptr, f0 + F))) return done(rc);
    } else {
        if ((rc = stack_forward(m, st, cs, m->cx, B, F, streaming ? m->pos_c : nullptr, f0 + F, nullptr))) return done(rc);
    }
    stage_ok("transformer", 0, 0);
    if ((rc = rmsnorm(st, m->cx, TPF(m, "codec.norm"), m->h, B * F, H, c.c_eps, dt))) return done(rc);
    S2S_HIP_CHECK(hipMemcpy2DAsync(at(m->ca, (size_t)LPAD * H * e), (size_t)(LPAD + F) * H * e, m->h, (size_t)F * H * e, (size_t)F * H * e, B,
                                   hipMemcpyDeviceToDevice, st));
    int T = F;
    void *cur = m->ca, *alt = m->cb_, *tmp = m->cc;
    for (int i = 0; i < 2; ++i) {
        const std::string q = "codec.up." + std::to_string(i) + ".";
        const int r = c.up_ratio[i];
        // transposed conv kernel == stride: out[t*r + j] = x[t] . W[j]  -> GEMM N = r*H, rows land time-major (stateless)
        GemmArgs g{};
        g.A = at(cur, (size_t)LPAD * H * e); g.lda = H; g.a_bs = (long)(LPAD + T) * H;
        g.W = TP(m, q + "tconv.w"); g.bias = TPF(m, q + "tconv.b");
        g.C = at(alt, (size_t)LPAD * H * e); g.ldc = (long)r * H; g.c_bs = (long)(LPAD + T * r) * H;
        g.M = B * T; g.N = r * H; g.K = H; g.G = T; g.ab_dtype = dt; g.c_dtype = dt; g.act = S2S_ACT_NONE;
        HLOAD(s_dw, alt, T * r, H);
        if ((rc = gemm_ex(st, g))) return done(rc);
        T *= r;
        if ((rc = hist_save(alt, T, H, s_dw))) return done(rc);
        // ConvNeXt: dwconv7 + LN -> pw1 GELU -> pw2 (gamma folded); x_out = x + that, written to the third buffer
        if ((rc = dwconv7_ln(st, alt, TPF(m, q + "dw.w"), TPF(m, q + "dw.b"), TPF(m, q + "ln.g"), TPF(m, q + "ln.b"), cur, B, T, H, LPAD, 1e-6f, dt))) return done(rc);
        if ((rc = gemm(st, cur, H, TP(m, q + "pw1.w"), TPF(m, q + "pw1.b"), nullptr, tmp, 4 * H, B * T, 4 * H, H, dt, dt, S2S_ACT_GELU, false))) return done(rc);
        GemmArgs g2{};
        g2.A = tmp; g2.lda = 4 * H; g2.a_bs = (long)T * 4 * H;
        g2.W = TP(m, q + "pw2.w"); g2.bias = TPF(m, q + "pw2.b");
        HLOAD(s_out, cur, T, H);
        g2.C = at(cur, (size_t)LPAD * H * e); g2.ldc = H; g2.c_bs = (long)(LPAD + T) * H;     // cur = alt + pw2(...)  (residual read in the epilogue)
        g2.res_c = at(alt, (size_t)LPAD * H * e);
        g2.M = B * T; g2.N = H; g2.K = 4 * H; g2.G = T; g2.ab_dtype = dt; g2.c_dtype = dt; g2.act = S2S_ACT_NONE;
        if ((rc = gemm_ex(st, g2))) return done(rc);
        if ((rc = hist_save(cur, T, H, s_out))) return done(rc);
        stage_ok("upsample stage", i, 0);
    }
    // Decoder.  Three rotating buffers per block: X (the block's x), S (a snake'd activation, with conv history rows), F (free).
    // Every SnakeBeta runs in the epilogue of the GEMM that produces its input (second output C2), every residual add in
    // the epilogue of the unit's 1x1 conv (in place on x): the blocks are GEMM launches + history copies only.
    // decoder.0: causal conv k7 H -> Dd; second output = snake of block 0
    int C = c.c_decoder_dim;
    void *X = alt, *S = tmp, *Fb = cur;
    auto snake_tables = [&](int i, const float** a, const float** ib) {      // the snake that consumes block i's input (or the output snake)
        const std::string q = i < c.n_dec_blocks ? "codec.dec." + std::to_string(i) + ".snake." : std::string("codec.dec.out.");
        *a = TPF(m, q + "a"); *ib = TPF(m, q + "ib");
    };
    {
        GemmArgs g{};
        g.A = at(cur, (size_t)(LPAD - 6) * H * e); g.lda = H; g.a_bs = (long)(LPAD + T) * H;
        g.W = TP(m, "codec.dec.in.w"); g.bias = TPF(m, "codec.dec.in.b");
        g.C = at(X, (size_t)LPAD * C * e); g.ldc = C; g.c_bs = (long)(LPAD + T) * C;
        g.M = B * T; g.N = C; g.K = 7 * H; g.G = T; g.ab_dtype = dt; g.c_dtype = dt; g.act = S2S_ACT_NONE;
        g.tap_len = H; g.tap_stride = H;
        g.C2 = at(S, (size_t)LPAD * C * e); snake_tables(0, &g.snake2_a, &g.snake2_ib);
        HLOAD(s_sn, S, T, C);
        if ((rc = gemm_ex(st, g))) return done(rc);
        if ((rc = hist_save(S, T, C, s_sn))) return done(rc);
        stage_ok("decoder in-conv", 0, 0);
    }
    for (int i = 0; i < c.n_dec_blocks; ++i) {
        const std::string q = "codec.dec." + std::to_string(i) + ".";
        const int r = c.up_rate[i], Co = C >> 1;
        // transposed conv k=2r, s=r (trimmed r on both sides): out[(t-1)*r + j] = x[t-1].W[j+r] + x[t].W[j]
        // first chunk: t = 1..T-1; later chunks: t = 0..T-1 with x[-1] = last row of the previous chunk (history row)
        const int Tm = first ? T - 1 : T;
        void* xbuf = Fb;     // the block's residual stream
        void* y1 = X;        // snake(x) with history (the old x is dead: its snake is in S)
        void* y2 = S;        // unit scratch, free once the transposed conv has read S
        {
            GemmArgs g{};
            g.A = at(S, (size_t)(first ? LPAD : LPAD - 1) * C * e); g.lda = C; g.a_bs = (long)(LPAD + T) * C;
            g.W = TP(m, q + "tconv.w"); g.bias = TPF(m, q + "tconv.b");
            g.C = at(xbuf, (size_t)LPAD * Co * e); g.ldc = (long)r * Co; g.c_bs = (long)(LPAD + Tm * r) * Co;
            g.M = B * Tm; g.N = r * Co; g.K = 2 * C; g.G = Tm; g.ab_dtype = dt; g.c_dtype = dt; g.act = S2S_ACT_NONE;
            g.C2 = at(y1, (size_t)LPAD * Co * e); g.snake2_a = TPF(m, q + "ru0.s1.a"); g.snake2_ib = TPF(m, q + "ru0.s1.ib"); g.snake_mod = Co;
            HLOAD(s_y1, y1, Tm * r, Co);
            if ((rc = gemm_ex(st, g))) return done(rc);
            if ((rc = hist_save(y1, Tm * r, Co, s_y1))) return done(rc);
            stage_ok("transposed conv", i, 0);
        }
        T = Tm * r;
        C = Co;
        for (int j = 0; j < 3; ++j) {
            const std::string u = q + "ru" + std::to_string(j) + ".";
            const int dil = j == 0 ? 1 : (j == 1 ? 3 : 9);
            {   // y2 = snake_s2(conv k7 dilated(y1))   (unpadded rows: the 1x1 conv needs no history)
                GemmArgs g{};
                g.A = at(y1, (size_t)(LPAD - 6 * dil) * C * e); g.lda = C; g.a_bs = (long)(LPAD + T) * C;
                g.W = TP(m, u + "c1.w"); g.bias = TPF(m, u + "c1.b");
                g.C = y2; g.ldc = C; g.c_bs = (long)T * C;
                g.M = B * T; g.N = C; g.K = 7 * C; g.G = T; g.ab_dtype = dt; g.c_dtype = dt;
                g.tap_len = C; g.tap_stride = (long)dil * C;
                g.act = S2S_ACT_SNAKE; g.snake_a = TPF(m, u + "s2.a"); g.snake_ib = TPF(m, u + "s2.ib");
                if ((rc = gemm_ex(st, g))) return done(rc);
                stage_ok("unit conv k7", i, j);
            }
            // x += conv 1x1(y2) in place; second output y1 = the NEXT snake of x (next unit's s1, next block's snake, or the output snake)
            GemmArgs g{};
            g.A = y2; g.lda = C; g.a_bs = (long)T * C;
            g.W = TP(m, u + "c2.w"); g.bias = TPF(m, u + "c2.b");
            g.C = at(xbuf, (size_t)LPAD * C * e); g.ldc = C; g.c_bs = (long)(LPAD + T) * C; g.res_c = g.C;
            g.M = B * T; g.N = C; g.K = C; g.G = T; g.ab_dtype = dt; g.c_dtype = dt; g.act = S2S_ACT_NONE;
            g.C2 = at(y1, (size_t)LPAD * C * e);
            if (j < 2) {
                const std::string un = q + "ru" + std::to_string(j + 1) + ".";
                g.snake2_a = TPF(m, un + "s1.a"); g.snake2_ib = TPF(m, un + "s1.ib");
            } else {
                snake_tables(i + 1, &g.snake2_a, &g.snake2_ib);
            }
            HLOAD(s_nx, y1, T, C);
            if ((rc = gemm_ex(st, g))) return done(rc);
            if ((rc = hist_save(y1, T, C, s_nx))) return done(rc);
            stage_ok("unit conv 1x1 + residual", i, j);
        }
        X = xbuf; S = y1; Fb = y2;
    }
    // output: conv k7 C -> 1 over the output snake, clamp
    if ((rc = conv_gemm(m, st, S, T, C, 7, 1, "codec.dec.out.w", "codec.dec.out.b", Fb, LPAD, 1, nullptr, S2S_ACT_NONE, dt))) return done(rc);
    tmp = Fb;
#undef HLOAD
    const long skip = (long)ctx * c.total_upsample;
    S2S_REQUIRE(n_samples == T - skip, "codec_decode: n_samples %d != %ld", n_samples, T - skip);
    if ((rc = wav_out(st, tmp, wav_dev, B, T, LPAD, (int)skip, n_samples, dt))) return done(rc);
    if (streaming) frames_done += F;
    return done(S2S_OK);
}

}  // namespace s2s

// codes_dev int32 [B][F][G] frame-major (ctx frames of left context first); wav_dev f32 [B][n_samples] receives the
// samples after the first ctx*total_upsample ones; n_samples = codec_samples(F) - ctx*total_upsample.
extern "C" int32_t s2s_codec_decode(void* model, void* stream, const int32_t* codes_dev, int32_t B, int32_t F, int32_t ctx,
                                    float* wav_dev, int32_t n_samples) {
    Tts* m = (Tts*)model;
    S2S_REQUIRE(m && m->finalized, "codec_decode: model not finalized");
    S2S_REQUIRE(B > 0 && B <= m->max_batch && F >= 1 && F <= m->cfg.max_codec_frames && ctx >= 0 && ctx < F && codes_dev && wav_dev, "codec_decode: bad arguments");
    return codec_run(m, (hipStream_t)stream, codes_dev, B, F, ctx, false, wav_dev, n_samples);
}

extern "C" int32_t s2s_codec_stream_reset(void* model) {
    Tts* m = (Tts*)model;
    S2S_REQUIRE(m && m->finalized, "codec_stream_reset: model not finalized");
    m->codec_frames_done = 0;
    return S2S_OK;
}

// Streaming continuation: F new frames -> exactly the samples a whole-utterance decode adds for them
// (first call: s2s_codec_samples(F); later calls: F * total_upsample).
extern "C" int32_t s2s_codec_stream_decode(void* model, void* stream, const int32_t* codes_dev, int32_t B, int32_t F, float* wav_dev,
                                           int32_t n_samples) {
    Tts* m = (Tts*)model;
    S2S_REQUIRE(m && m->finalized, "codec_stream_decode: model not finalized");
    S2S_REQUIRE(B > 0 && B <= m->max_batch && F >= 1 && F <= m->cfg.max_codec_frames && codes_dev && wav_dev, "codec_stream_decode: bad arguments");
    S2S_REQUIRE(m->codec_frames_done + F <= m->codec.c.max_pos, "codec_stream_decode: %d frames exceed the stream capacity %d", m->codec_frames_done + F, m->codec.c.max_pos);
    return codec_run(m, (hipStream_t)stream, codes_dev, B, F, 0, true, wav_dev, n_samples);
}

/* Streaming continuation of a COHORT: rows [slot0, slot0 + n) of the engine's codec state, frames_done frames already decoded for
 * them (0: the stream starts here).  codes_dev int32 [n][F][G]; wav_dev f32 [n][n_samples] with n_samples = s2s_codec_samples(F) for
 * the first call of a stream and F * total_upsample afterwards.  Cohorts on disjoint row ranges stream independently. */
extern "C" int32_t s2s_codec_decode_rows(void* model, void* stream, const int32_t* codes_dev, int32_t slot0, int32_t n, int32_t F, int32_t frames_done,
                                         float* wav_dev, int32_t n_samples) {
    Tts* m = (Tts*)model;
    S2S_REQUIRE(m && m->finalized, "codec_decode_rows: model not finalized");
    S2S_REQUIRE(slot0 >= 0 && n > 0 && slot0 + n <= m->max_batch && F >= 1 && F <= m->cfg.max_codec_frames && frames_done >= 0 && codes_dev && wav_dev,
                "codec_decode_rows: bad arguments");
    S2S_REQUIRE(frames_done + F <= m->codec.c.max_pos, "codec_decode_rows: %d frames exceed the stream capacity %d", frames_done + F, m->codec.c.max_pos);
    int fd = frames_done;
    return codec_run(m, (hipStream_t)stream, codes_dev, n, F, 0, true, wav_dev, n_samples, slot0, &fd);
}

/* Streaming continuation of rows of DIFFERENT ages in one pass: frames_done_rows_dev int32[n] (device) = frames already decoded per row
 * (every row continues a stream: the first chunk of a stream has its own geometry and goes through s2s_codec_decode_rows with
 * frames_done = 0), max_frames_done = the largest entry.  Rows of the range that carry no live stream may be passed with any
 * position; their output is meaningless and their state is rewritten by the first chunk of the next stream that uses the slot.
 * wav_dev f32 [n][F * total_upsample]. */
extern "C" int32_t s2s_codec_decode_rows_at(void* model, void* stream, const int32_t* codes_dev, int32_t slot0, int32_t n, int32_t F,
                                            const int32_t* frames_done_rows_dev, int32_t max_frames_done, float* wav_dev, int32_t n_samples) {
    Tts* m = (Tts*)model;
    S2S_REQUIRE(m && m->finalized, "codec_decode_rows_at: model not finalized");
    S2S_REQUIRE(slot0 >= 0 && n > 0 && slot0 + n <= m->max_batch && F >= 1 && F <= m->cfg.max_codec_frames && max_frames_done >= 0 && codes_dev && wav_dev &&
                    frames_done_rows_dev, "codec_decode_rows_at: bad arguments");
    S2S_REQUIRE(max_frames_done + F <= m->codec.c.max_pos, "codec_decode_rows_at: %d frames exceed the stream capacity %d", max_frames_done + F, m->codec.c.max_pos);
    int fd = max_frames_done;
    return codec_run(m, (hipStream_t)stream, codes_dev, n, F, 0, true, wav_dev, n_samples, slot0, &fd, frames_done_rows_dev);
}

extern "C" int64_t s2s_codec_samples(void* model, int32_t F) {
    Tts* m = (Tts*)model;
    if (!m) return 0;
    return s2s::codec_len(m->cfg, F, m->cfg.n_dec_blocks);
}

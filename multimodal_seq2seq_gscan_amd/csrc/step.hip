// Host-side sequencing of one training step: Model.forward (seq2seq/model.py:206-219) and
// its reverse.  Pure launch code — no allocation, no synchronisation, one stream — so the
// whole step can be captured in a hipGraph by the caller.
//
// Data layout in HBM.  Parameters and gradients keep the reference's state_dict layout and
// are consumed in place through strided GEMM operands.  Activations live in one caller-owned
// workspace, carved by workspace_layout() below; rows are always (batch, time|cell)-major with
// the feature dimension innermost, so every dense product is a plain [rows, K] x [K, N] GEMM:
//   S[b,t,:] = [ e_t | ctx_text_t | ctx_vis_t | h_t ]  (4H)  feeds the LSTM weight-gradient GEMM,
//   the output head and the context gradients without any copy or concat.
#include "step.h"

namespace gscan {

static inline int enc_layers(const gscan_dims &d) { return d.enc_layers > 1 ? d.enc_layers : 1; }

// Parameters of encoder layer l: layer 0 sits in the named fields, layers 1.. in enc_deep (include/gscan_hip.h).
struct EncLayer { float *w_ih, *w_hh, *b_ih, *b_hh, *w_ih_rev, *w_hh_rev, *b_ih_rev, *b_hh_rev; };
static inline EncLayer enc_layer(const gscan_params &p, int l) {
    if (l == 0) return {p.enc_w_ih, p.enc_w_hh, p.enc_b_ih, p.enc_b_hh, p.enc_w_ih_rev, p.enc_w_hh_rev, p.enc_b_ih_rev,
                        p.enc_b_hh_rev};
    float *const *q = p.enc_deep[l - 1];
    return {q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]};
}

// --------------------------------------------------------------------------------------
// workspace
// --------------------------------------------------------------------------------------
// Long target sequences (round 5): a context is alpha . PK, so what the LSTM input and the conditional query hand back to
// the two contexts can be summed over TIME per memory first (alpha_reduce, attention_grad.hip: G = alpha^T . [delta | dzq]) and
// pushed through the weights afterwards — d PK += G . W over B (L + G^2) rows instead of d ctx = [delta | dzq] . W over B T rows,
// and the context columns of dW_ih / dW_q2k as G^T . PK with K = B (L + G^2) instead of K = B T.  Pays when T is well above
// the number of memories (S3: T = 120 against 46; on from T >= 1.2 (L + G^2)): GSCAN_TIME_REDUCED_T=<T> moves the threshold (0 = never; given explicitly it
// is the only condition, which is how the tests force the path on short sequences).
bool attention_time_reduced(int T, int L, int M) {
    static const int given = [] { const char *e = getenv("GSCAN_TIME_REDUCED_T"); return e ? atoi(e) : -1; }();
    if (given >= 0) return given > 0 && T >= given;
    return T >= 48 && 5 * T >= 6 * (L + M);      // measured crossover at L + G^2 = 46: T = 40 loses 1 %, 56 gains 0.7 %, 72 1.2 %, 96 2.5 %, 120 3.5 %
}

int workspace_layout(const gscan_dims &d, Workspace *ws) {
    const int64_t B = d.B, L = d.L, T = d.T, M = (int64_t)d.G * d.G, Co = d.Co, F = 3 * Co, E = d.E, He = d.He,
                  H = d.H, V = d.V, D = d.bidirectional ? 2 : 1;
    int64_t p = 0;
    int n = 0;
    auto take = [&](const char *name, int64_t count) {
        ws->slot[n].name = name;
        ws->slot[n].offset = p;
        ws->slot[n].count = count;
        ++n;
        p += (count + 63) / 64 * 64;          // 256-byte aligned slots
        return ws->slot[n - 1].offset;
    };
#define SLOT(field, count) ws->field = take(#field, (count))
    SLOT(conv_img, conv_image_floats(d.C, d.Co, d.K3));
    SLOT(conv_flags, kFusedMaxFlags + 64);           // 32-bit flags of the prologue + world encoder launch, then its count of self-served chunks (conv.hip)
    SLOT(conv_lists, (int64_t)world_conv_backward_scratch_floats(d.B, d.G, d.C));
    // per-chunk partial sums of the convolution bias gradients: the fixed-order form (deterministic mode only)
    SLOT(conv_bias_part, gemm_macro_tile_mode() > 0 ? (int64_t)world_conv_bias_partial_floats(d.B, d.G, d.Co) : 0);
    SLOT(embed_part_dec, gemm_macro_tile_mode() > 0 ? (int64_t)embed_grad_partial_floats(B * T, H, V) : 0);
    SLOT(embed_part_enc, gemm_macro_tile_mode() > 0 ? (int64_t)embed_grad_partial_floats(B * L, d.E, d.Vi) : 0);
    SLOT(feat, B * M * F);
    SLOT(pkv, B * M * H);
    SLOT(uv, B * M * 4 * H);
    SLOT(xe, B * L * E);
    SLOT(gx, B * L * D * 4 * He);
    SLOT(enc_out, B * L * He);                       // enc_out | hN | dxe are zeroed together by the prologue
    SLOT(hN, B * He);                                // (accumulation targets: direction sums, split-K tail)
    SLOT(dxe, B * L * E);
    SLOT(enc_gates, B * L * D * 4 * He);
    SLOT(enc_cells, B * L * D * He);
    SLOT(enc_hprev, B * L * D * He);
    SLOT(pkt, B * L * H);
    SLOT(ut, B * L * 4 * H);
    SLOT(u2t, B * L * H);
    SLOT(bsum, 4 * H);
    SLOT(dwc, V * 4 * H);                            // gradient of the composite head (head_wc below)
    SLOT(wih_stack, D * 4 * He * E);
    SLOT(wih_t, D * 4 * He * (E + 1));
    SLOT(w_sk, 4 * H * F);
    SLOT(w_ck, 4 * H * He);
    SLOT(w_2kk, H * He);
    // (the streaming decoder keeps its trip-major copy of the weights that multiply h here: any_stream_image_floats, anyshape.h)
    SLOT(dec_w_fwd, std::max<int64_t>(decoder_geometry(d.H, d.conditional != 0).image_floats,
                                      (int64_t)(((6 * d.H + 255) / 256) * ((d.H + 63) / 64)) * 4 * 1024 * 4));
    SLOT(dec_w_bwd, decoder_geometry(d.H, d.conditional != 0).image_floats);
    SLOT(enc_w_image, D * 4 * He * He);
    SLOT(hprev, B * T * H);
    SLOT(S, B * T * 4 * H);
    SLOT(ge, B * T * 4 * H);
    SLOT(cells, B * T * H);
    SLOT(gates, B * T * 4 * H);
    SLOT(alpha_c, B * T * L);
    SLOT(alpha_s, B * T * M);
    SLOT(q2, B * T * H);
    SLOT(qt, B * T * H);
    SLOT(qv, B * T * H);
    SLOT(att_sum, B * M);
    SLOT(wcat5, 5 * H * 3 * H);
    SLOT(logits, B * T * V);
    SLOT(logp_saved, B * T * V);
    SLOT(aux_saved, B * M);
    SLOT(row_stats, B * 4);                          // per-row [sum NLL, live tokens, aux NLL, 1]
    // backward scratch
    SLOT(dlogits, B * T * V);
    SLOT(dS, B * T * 4 * H);
    SLOT(datt, B * M);
    SLOT(delta, B * T * 5 * H);                      // [delta (4H) | dzq (H)] per (b,t)
    SLOT(dzq, 64);
    SLOT(dqt, B * T * H);
    SLOT(dqv, B * T * H);
    SLOT(dpk_t, B * L * H);
    SLOT(dpk_v, B * M * H);
    SLOT(dv_t, B * H);
    SLOT(dv_v, B * H);
    SLOT(dh0, B * H);
    SLOT(denc, B * L * He);
    SLOT(dhN, B * He);
    SLOT(enc_delta, B * L * D * 4 * He);
    SLOT(dfeat, B * M * F);
    SLOT(stamps, 64);
    // split-K slabs of the macro-tile GEMM (gemm_mt.hip): one region per stream that carries split products (side 1:
    // decoder leaves, then key leaves; the caller's: the encoder's weight gradients), kGemmSlabs partial tiles each
    // (deterministic mode only: GSCAN_DETERMINISTIC=1 is read once per process, so the layout is the same for every call)
    const int64_t slabs = gemm_macro_tile_mode() > 0 ? kGemmSlabs * (int64_t)gemm_slab_floats() : 0;
    SLOT(gemm_slabs_side, slabs);
    SLOT(gemm_slabs_main, slabs);
    // encoder layers below the last one (num_encoder_layers > 1): saved activations, outputs (= the next layer's
    // input, [B,L,D*He]) and their gradients per layer; register images of W_hh for layers 1..
    const int64_t deep = enc_layers(d) - 1;
    SLOT(deep_gates, deep * B * L * D * 4 * He);
    SLOT(deep_cells, deep * B * L * D * He);
    SLOT(deep_hprev, deep * B * L * D * He);
    SLOT(deep_y, deep * B * L * D * He);
    SLOT(deep_dy, deep * B * L * D * He);
    SLOT(deep_delta, deep * B * L * D * 4 * He);
    SLOT(deep_image, deep * D * 4 * He * He);
    SLOT(ge_table, V * 4 * H);                       // greedy decoding: Emb . W_ih[:, :H]^T + biases
    SLOT(head_wc, V * 4 * H);                        // the output head as one matrix: W_h2o . W_o2h (S order), step prologue
    SLOT(drawn_mask_enc, B * L * E);                 // dropout drawn in the kernels (gscan_masks::in_kernel): the embedding
    SLOT(drawn_mask_dec, B * T * H);                 // gathers leave their keep values here for the embedding gradients
    const bool reduced = attention_time_reduced((int)T, (int)L, (int)M);
    SLOT(g_t, reduced ? B * L * 5 * H : 0);          // sums over time of [delta | dzq] per textual / visual memory
    SLOT(g_v, reduced ? B * M * 4 * H : 0);
#undef SLOT
    ws->nslots = n;
    ws->total_floats = p;
    return 0;
}

int check_dims(const gscan_dims &d) {
    GSCAN_CHECK(d.B > 0 && d.L > 0 && d.T > 0 && d.G > 0 && d.C > 0 && d.Co > 0 && d.E > 0 && d.V > 1 && d.Vi > 1,
                "dims: non-positive dimension (B=%d L=%d T=%d G=%d C=%d Co=%d E=%d V=%d Vi=%d)", d.B, d.L, d.T, d.G,
                d.C, d.Co, d.E, d.V, d.Vi);
    GSCAN_CHECK(d.K3 > 0 && (d.K3 & 1), "dims: cnn_kernel_size must be odd (got %d)", d.K3);
    // Hidden sizes, command lengths and grid sizes outside what the register/LDS-resident kernels take run on the
    // streaming kernels (decoder_any.hip, lstm_encoder.hip's *_any kernels); what remains are the limits of those.
    GSCAN_CHECK(d.He >= 1 && d.He <= 2048, "dims: encoder_hidden_size %d is outside 1..2048", d.He);
    GSCAN_CHECK(d.H >= 1 && d.H <= 1024, "dims: decoder_hidden_size %d is outside 1..1024 (the streaming decoder gives every "
                "feature of an attention a thread)", d.H);
    GSCAN_CHECK(d.E <= 1024, "dims: embedding_dimension %d is outside 1..1024", d.E);
    GSCAN_CHECK((int64_t)d.B * d.T * 4 * d.H < (1ll << 31) && (int64_t)d.B * d.G * d.G * 4 * d.H < (1ll << 31),
                "dims: batch too large for 32-bit activation offsets (B=%d T=%d H=%d)", d.B, d.T, d.H);
    GSCAN_CHECK(d.enc_layers >= 0 && d.enc_layers <= GSCAN_MAX_ENC_LAYERS,
                "dims: at most %d encoder layers are supported (got %d)", GSCAN_MAX_ENC_LAYERS, d.enc_layers);
    GSCAN_CHECK(d.L <= 4096 && d.G * d.G <= 4096, "dims: more than 4096 memories per attention (L=%d, G=%d)", d.L, d.G);
    return 0;
}

// Split of the long K dimension of a weight-gradient product.  A workgroup's K loop is a chain of dependent
// ~1.2 us rounds when it has a CU to itself, whatever the round computes (tools/gemm_shapes.py: 400x100x2560
// unsplit = 109 us, 8 slices = 17 us), and every slice ends in one float atomic per output element; the atomics of
// all slices of an element arrive together.  Measured on the whole step: slices of 640 rows (20 rounds), but at
// least 8 of them while they stay >= 160 rows.  Shorter slices make single products faster in isolation
// (100x150x9216: 32 -> 21 us from 15 to 58 slices) but the overlapped step slower (0.691 vs 0.683 ms), and the
// eight-product decoder launch much slower (85 vs 62 us at 160/320-row slices).
static thread_local int g_split_override = 0;      // set around one launch's add_grad calls (experiments)
static int pick_split(int K) {
    static const int forced = [] { const char *e = getenv("GSCAN_SPLIT"); return e ? atoi(e) : 0; }();
    if (g_split_override > 0 && K >= 2560) return std::min(g_split_override, cdiv(K, 160));
    if (forced > 0 && K >= 2560) return std::min(forced, cdiv(K, 160));
    return std::max(cdiv(K, 640), std::min(8, cdiv(K, 160)));
}

// weight gradient: C[M,N] += A^T . B with the long dimension (rows of the activations) as K, split over
// workgroups; bias1/bias2 (optional) += column sums of the activation gradient = sum over K of A(m,k)
static inline void add_grad(GemmBatch &g, int M, int N, int K, const float *a, int64_t sam, int64_t sak, const float *b,
                            int64_t sbk, int64_t sbn, float *c, int64_t ldc, float *bias1 = nullptr,
                            float *bias2 = nullptr) {
    g.add(M, N, K, a, sam, sak, b, sbk, sbn, c, ldc, 1.f, nullptr, 0, nullptr, pick_split(K), bias1, bias2);
}

#define TRY(expr) do { if (int rc_ = (expr)) return rc_; } while (0)

static DecoderArgs decoder_args(const gscan_dims &d, const gscan_params &p, const gscan_batch &bt, float *w,
                                const Workspace &ws) {
    DecoderArgs a{};
    a.T = d.T; a.L = d.L; a.M = d.G * d.G;
    a.cmd_lengths = bt.cmd_lengths;
    a.pk_t = w + ws.pkt; a.u_t = w + ws.ut; a.u2_t = w + ws.u2t;
    a.pk_v = w + ws.pkv; a.u_v = w + ws.uv;
    a.ge = w + ws.ge;
    a.b_q2k = p.q2k_b; a.v_t = p.txt_energy_w; a.v_v = p.vis_energy_w;
    a.any_w_hh = p.dec_w_hh; a.any_w_ih = p.dec_w_ih; a.any_w_qt = p.txt_query_w; a.any_w_qv = p.vis_query_w;
    a.any_w_q2k = p.q2k_w;
    a.any_wcat = w + ws.dec_w_bwd; a.any_wcat_stream = w + ws.dec_w_fwd;
    a.hprev = w + ws.hprev; a.s = w + ws.S; a.cells = w + ws.cells; a.gates = w + ws.gates;
    a.alpha_c = w + ws.alpha_c; a.alpha_s = w + ws.alpha_s;
    a.q2 = w + ws.q2; a.qt = w + ws.qt; a.qv = w + ws.qv; a.att_sum = w + ws.att_sum;
    a.V = d.V; a.head_wc = w + ws.head_wc;
    a.logits = w + ws.logits; a.logp_saved = w + ws.logp_saved;
    a.aux_saved = d.auxiliary ? w + ws.aux_saved : nullptr;
    a.targets = bt.targets; a.positions = d.auxiliary ? bt.target_positions : nullptr;
    a.pad_tgt = d.pad_tgt; a.B = d.B; a.row_stats = w + ws.row_stats;
    return a;
}

// --------------------------------------------------------------------------------------
// Two-stream schedule.  Most launches of the step are small and latency-bound, and the dependency graph has
// width: in forward the world branch (sparse conv -> visual keys) and the command branch (input projection
// -> BiLSTM -> textual keys) are independent until the decoder; in backward every weight-gradient product is
// a leaf that nothing waits for except the optimiser.  The main stream (the caller's) carries the critical
// chain; a side stream carries the other branch / the leaves, tied together with HIP events.  Event and
// stream objects are created once per process on first use (the only non-launch work this file ever does) and
// the same sequence can be captured into a hipGraph (cross-stream capture through the events).
// --------------------------------------------------------------------------------------
struct SideStream {
    hipStream_t stream = nullptr, stream2 = nullptr;
    bool single = false;             // GSCAN_SINGLE_STREAM=1 (diagnostic): everything on the caller's stream
    hipEvent_t ev[16] = {};
    hipEvent_t early = nullptr;      // the backward pass's first leaf stream is done (gscan_early_gradients_wait)
    bool early_recorded = false;
    bool ready = false;
    int next = 0;
};
static SideStream g_side;
// two-bucket gradient exchange: the range of the flat gradient that the backward pass all-reduces on its first leaf stream
// (gscan_comm_set_early_allreduce); NULL = none
static void *g_early_comm = nullptr;
static float *g_early_buf = nullptr;
static size_t g_early_n = 0;

static int side_init() {
    if (g_side.ready) return 0;
    const char *single = getenv("GSCAN_SINGLE_STREAM");
    g_side.single = single && single[0] == '1';
    // highest priority: the leaves these streams carry end the step (the optimiser waits for the last of them),
    // measured 0.5% faster than lowest priority
    int prio_least = 0, prio_greatest = 0;
    GSCAN_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    GSCAN_HIP(hipStreamCreateWithPriority(&g_side.stream, hipStreamNonBlocking, prio_greatest));
    GSCAN_HIP(hipStreamCreateWithPriority(&g_side.stream2, hipStreamNonBlocking, prio_greatest));
    // The events order kernels of ONE device across streams: no system-scope fence (cache write-back / invalidate for
    // the host's benefit) is needed when one completes, and leaving it out takes 6 us off a step (0.5525 -> 0.546 ms,
    // profiles/r02_ab_event_flags.txt).  GSCAN_EVENT_FLAGS=0 restores the default fence.
    const char *evf = getenv("GSCAN_EVENT_FLAGS");
    const unsigned extra = evf ? (unsigned)strtoul(evf, nullptr, 0) : (unsigned)hipEventDisableSystemFence;
    for (auto &e : g_side.ev) GSCAN_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming | extra));
    GSCAN_HIP(hipEventCreateWithFlags(&g_side.early, hipEventDisableTiming | extra));
    g_side.ready = true;
    return 0;
}
// `waiter` will not run anything issued after this call until everything issued so far on `signaller` is done
// (a record costs the signalling stream a ~6 us bubble on MI355X, so fork points are few and an event is shared)
static int order_after(hipStream_t waiter, hipStream_t signaller, hipStream_t waiter2 = nullptr) {
    if (waiter == signaller) return 0;
    hipEvent_t e = g_side.ev[g_side.next];
    g_side.next = (g_side.next + 1) % 16;
    GSCAN_HIP(hipEventRecord(e, signaller));
    GSCAN_HIP(hipStreamWaitEvent(waiter, e, 0));
    if (waiter2 && waiter2 != signaller) GSCAN_HIP(hipStreamWaitEvent(waiter2, e, 0));
    return 0;
}

// --------------------------------------------------------------------------------------
// forward
// --------------------------------------------------------------------------------------
// Everything before the decoder: both encoders, the projected keys and their gate images, the bridge, and the
// per-step weight images.  teacher_forced = the target tokens of all T steps are known (training / scoring):
// their embeddings and the embedding part of the gate pre-activations are computed here too.
// `given` (Model.decode_input_batched, model.py:190-204): the encodings come from the caller — the two encoders are
// skipped, the given tensors take the place of their outputs in the workspace and everything behind them runs.
struct GivenEncodings { const float *feat, *enc_out, *hN; };   // [B,G*G,3Co] [B,L,He] [B,He]
// the dropout of segment `seg` as the kernels see it: drawn in the kernels (gscan_masks::in_kernel) or, if not, the mask
// pointer decides as before
static DropSpec drop_of(const gscan_masks &mk, int seg) {
    const float p = seg == kDropSegCnn ? mk.p_cnn : seg == kDropSegEnc ? mk.p_enc : mk.p_dec;
    return drop_spec(mk.in_kernel != 0, mk.seed, mk.stream_id, p);
}
static int check_drop(const gscan_masks &mk) {
    if (!mk.in_kernel) return 0;
    for (float p : {mk.p_cnn, mk.p_enc, mk.p_dec})
        GSCAN_CHECK(p >= 0.f && p < 1.f, "dropout drawn in the kernels: p=%g out of [0,1)", p);
    return 0;
}

static int encode_branches(const gscan_dims &d, const gscan_params &p, const gscan_batch &bt, const gscan_masks &mk,
                           float *w, const Workspace &ws, bool teacher_forced, hipStream_t st,
                           const GivenEncodings *given = nullptr) {
    const int B = d.B, L = d.L, T = d.T, M = d.G * d.G, C = d.C, Co = d.Co, F = 3 * Co, E = d.E, He = d.He, H = d.H,
              V = d.V, D = d.bidirectional ? 2 : 1;
    const bool cond = d.conditional != 0;
    GSCAN_CHECK(!cond || (p.q2k_w && p.q2k_b), "forward: conditional attention needs queries_to_keys parameters");
    GSCAN_CHECK(D == 1 || (p.enc_w_ih_rev && p.enc_w_hh_rev && p.enc_b_ih_rev && p.enc_b_hh_rev),
                "forward: bidirectional encoder needs the *_reverse parameters");
    GSCAN_CHECK(bt.cmd_lengths && (given || (bt.commands && (bt.world || bt.world_u8))) && (!teacher_forced || bt.targets),
                "forward: NULL array in the batch");
    TRY(side_init());
    TRY(check_drop(mk));
    const DropSpec drop_cnn = drop_of(mk, kDropSegCnn), drop_enc = drop_of(mk, kDropSegEnc), drop_dec = drop_of(mk, kDropSegDec);
    hipStream_t sd = g_side.single ? st : g_side.stream;

    // ================= prelude schedule ============================================================================
    // prologue -> world encoder -> command recurrence (its first layer projects its own input) -> ONE GEMM launch
    // {textual keys + gate images, bridge, visual keys + gate images, embedding part of the decoder gates} -> decoder.
    // Round 1 ran the world branch on a side stream as two chip-filling GEMM launches racing the recurrence, whose
    // workgroups (two GEMM slots freed on the same CU at once) started 55 us late; round 2 first moved everything only
    // the decoder waits for onto two side streams (the three-stream schedule kept below for A/B runs), then found the
    // cross-stream events themselves to be the cost.  Measured and dropped (profiles/r02_ab_*): forking behind L1 or
    // behind the prologue (+20..30 us: the event latency lands on the critical path), host issue order (no effect).
    hipStream_t sd2 = g_side.single ? st : g_side.stream2;
    const bool gate_images = decoder_fast_supported(H, L, M, V, cond) || decoder_any_uses_gate_images(H, L, M);
    // one prologue launch with the segments of `which` (0: caller's stream, 1: side 1, 2: side 2); the others stay empty
    // fuse_world: the world encoder runs in the same launch (conv.hip, prologue_world_kernel), whose image workgroups
    // take the convolution weight image off the prologue's index space; returns -1 if that launch does not fit the shape
    auto prologue = [&](int which, hipStream_t stream, bool fuse_world = false) -> int {
        PrologueArgs a{};
        a.b_ih = p.dec_b_ih; a.b_hh = p.dec_b_hh; a.w_o2h = p.out2hid_w; a.w_h2o = p.hid2out_w;
        a.w_ih_f = p.enc_w_ih; a.w_ih_r = p.enc_w_ih_rev; a.enc_emb = p.enc_emb; a.dec_emb = p.dec_emb;
        a.mask_enc = mk.in_kernel ? nullptr : mk.enc; a.mask_dec = mk.in_kernel ? nullptr : mk.dec; a.commands = bt.commands;
        a.drop_enc = drop_enc; a.drop_dec = drop_dec;
        a.mask_enc_out = w + ws.drawn_mask_enc; a.mask_dec_out = w + ws.drawn_mask_dec;
        a.targets = teacher_forced ? bt.targets : nullptr;
        a.bsum = w + ws.bsum; a.head_wc = w + ws.head_wc; a.wih_stack = w + ws.wih_stack; a.wih_t = w + ws.wih_t;
        a.dwc = w + ws.dwc; a.xe = w + ws.xe; a.S = w + ws.S;
        a.H = H; a.He = He; a.E = E; a.D = D; a.BL = B * L; a.BT = B * T; a.Vi = d.Vi; a.V = V;
        a.wcat5 = w + ws.wcat5; a.w_ih_dec = p.dec_w_ih; a.w_q2k = p.q2k_w; a.cond = cond ? 1 : 0;
        a.zero_extra = w + ws.enc_out;                     // adjacent slots enc_out | hN | dxe
        a.zero_extra_count = (ws.dxe + (int64_t)B * L * E) - ws.enc_out;
        const DecoderGeometry geo = decoder_geometry(H, cond);
        // the register images belong to the resident decoder kernels (the gate images U and the composite weights behind
        // them are read by the streaming kernels too)
        const bool fast = decoder_fast_supported(H, L, M, V, cond);
        const bool images = fast || decoder_any_uses_gate_images(H, L, M);     // gate images U and their composite weights
        a.img = DecoderImageArgs{p.dec_w_hh, p.txt_query_w, p.vis_query_w, p.q2k_w, p.out2hid_w, w + ws.dec_w_fwd,
                                 w + ws.dec_w_bwd, H, cond ? 1 : 0, geo.slots, geo.k0};
        a.enc_w_hh_f = p.enc_w_hh; a.enc_w_hh_r = p.enc_w_hh_rev; a.enc_image = w + ws.enc_w_image;
        a.enc_b_ih_f = p.enc_b_ih; a.enc_b_hh_f = p.enc_b_hh; a.enc_b_ih_r = p.enc_b_ih_rev; a.enc_b_hh_r = p.enc_b_hh_rev;
        a.enc_rows = encoder_rows_per_thread(He);
        a.conv_w[0] = p.conv1_w; a.conv_w[1] = p.conv2_w; a.conv_w[2] = p.conv3_w;
        a.conv_img = w + ws.conv_img; a.cC = C; a.cCo = Co; a.cK3 = d.K3;
        a.w_key_vis = p.vis_key_w; a.w_key_txt = p.txt_key_w; a.F = F;
        a.w_sk = w + ws.w_sk; a.w_ck = w + ws.w_ck; a.w_2kk = w + ws.w_2kk;
        const int64_t n[14] = {4 * H, (int64_t)V * 4 * H, (int64_t)D * 4 * He * (E + 1), (int64_t)V * 4 * H,
                               given ? 0 : (drop_enc.on ? (int64_t)cdiv((int64_t)B * L, 4) * E : (int64_t)B * L * E),   // in-kernel dropout: an element per (four rows, column)
                               teacher_forced ? (drop_dec.on ? (int64_t)cdiv((int64_t)B * T, 4) * H : (int64_t)B * T * H) : 0,
                               (int64_t)5 * H * 3 * H,
                               a.zero_extra_count, fast ? 2 * geo.image_floats : 0,
                               encoder_fast_supported(He, L, E) ? (int64_t)D * 4 * He * He : 0,
                               (given || fuse_world) ? 0 : conv_image_floats(C, Co, d.K3),
                               images ? (int64_t)4 * H * F : 0, images ? (int64_t)4 * H * He : 0,
                               (images && cond) ? (int64_t)H * He : 0};
        int64_t acc = 0;
        for (int i = 0; i < 14; ++i) {
            // side 1: decoder bias sum, embedded targets; side 2: convolution weight image, visual composite weight
            const int owner = (i == 0 || i == 5) ? 1 : ((i == 10 || i == 11) ? 2 : 0);
            acc += (owner == which || which == 3) ? n[i] : 0;   // which 3: every segment in one launch
            a.end[i] = acc;
        }
        if (fuse_world) {
            const float *const cb[3] = {p.conv1_b, p.conv2_b, p.conv3_b};
            return prologue_world_forward(a, bt.world_u8 ? (const void *)bt.world_u8 : (const void *)bt.world,
                                          bt.world_u8 != nullptr, cb, mk.in_kernel ? nullptr : mk.cnn, B, d.G, C, Co, d.K3, w + ws.feat,
                                          reinterpret_cast<uint32_t *>(w + ws.conv_flags), stream, &drop_cnn);
        }
        return step_prologue(a, stream);
    };
    // the dense products of the prelude, each added to whichever launch the schedule below puts it in
    auto add_visual = [&](GemmBatch &k) {     // projected visual keys (seq2seq_model.py:466-467) and their gate images
        k.add(B * M, H, F, w + ws.feat, F, 1, p.vis_key_w, 1, F, w + ws.pkv, H);
        if (!gate_images) return;
        k.add(B * M, 4 * H, F, w + ws.feat, F, 1, w + ws.w_sk, 1, F, w + ws.uv, 4 * H);
        k.overhead();     // U image: its algorithmic counterpart, W_ih[:, ctx_vis] . ctx_vis per step, is charged to the decoder kernel
    };
    auto add_ge = [&](GemmBatch &g) {         // embedding part of the decoder gate pre-activations for all t (teacher forcing)
        if (teacher_forced)
            g.add(B * T, 4 * H, H, w + ws.S, 4 * H, 1, p.dec_w_ih, 1, 3 * H, w + ws.ge, 4 * H, 0.f, w + ws.bsum);
    };
    auto add_textual = [&](GemmBatch &g) {    // projected textual keys (:468-469), their images, and the bridge (model.py:195)
        g.add(B * L, H, He, w + ws.enc_out, He, 1, p.txt_key_w, 1, He, w + ws.pkt, H);
        if (gate_images) {
            g.add(B * L, 4 * H, He, w + ws.enc_out, He, 1, w + ws.w_ck, 1, He, w + ws.ut, 4 * H);
            g.overhead(); // U images of the textual memories: charged to the decoder kernel as the context terms they replace
        }
        if (cond && gate_images) { g.add(B * L, H, He, w + ws.enc_out, He, 1, w + ws.w_2kk, 1, He, w + ws.u2t, H); g.overhead(); }
        g.add(B, H, He, w + ws.hN, He, 1, p.bridge_w, 1, He, w + ws.hprev, (int64_t)T * H, 0.f, p.bridge_b, 2);
    };
    auto world_encoder = [&](hipStream_t stream) -> int {      // cnn_model.py:22-36, input-sparse kernel (conv.hip)
        if (given) {
            GSCAN_HIP(hipMemcpyAsync(w + ws.feat, given->feat, sizeof(float) * (size_t)B * M * F, hipMemcpyDeviceToDevice,
                                     stream));
            return 0;
        }
        const float *const cb[3] = {p.conv1_b, p.conv2_b, p.conv3_b};
        return world_conv_forward(bt.world_u8 ? (const void *)bt.world_u8 : (const void *)bt.world, bt.world_u8 != nullptr,
                                  w + ws.conv_img, cb, mk.in_kernel ? nullptr : mk.cnn, B, d.G, C, Co, d.K3, w + ws.feat, stream,
                                  &drop_cnn);
    };
    // Default: the whole prelude on the caller's stream — one prologue, the world encoder, the recurrence, and ONE GEMM
    // launch with every dense product.  No cross-stream event: each costs 25-30 us of latency on this stack (record ->
    // the other queue's first kernel), more than overlapping the now short kernels buys, and the recurrence (128+
    // VGPRs per SIMD for one workgroup) no longer races floods of GEMM workgroups for CUs.  Measured: 0.553 ms per step
    // against 0.572 for the three-stream schedule below (profiles/r02_ab_forward_streams.txt), which
    // GSCAN_FORWARD_STREAMS=3 selects for A/B runs.
    static const int fwd_streams = [] { const char *e = getenv("GSCAN_FORWARD_STREAMS"); return e ? atoi(e) : 1; }();
    const bool merged = fwd_streams == 1 || g_side.single;
    {
        if (merged) {
            // the prologue and the world encoder as ONE launch (GSCAN_FUSED_PROLOGUE=0: two launches, for A/B runs)
            static const int fused = [] { const char *e = getenv("GSCAN_FUSED_PROLOGUE"); return e ? atoi(e) : 1; }();
            int rc = -1;
            // not while the stream is being captured into a graph: a replay would repeat the launch's epoch argument,
            // and the flags of the previous replay would already hold it
            hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
            GSCAN_HIP(hipStreamIsCapturing(st, &capturing));
            if (fused && !given && capturing == hipStreamCaptureStatusNone) rc = prologue(3, st, true);
            if (rc > 0) return rc;
            if (rc < 0) {
                TRY(prologue(3, st));
                TRY(world_encoder(st));
            }
        } else {
            TRY(order_after(sd, st, sd2));   // fork: whatever produced the inputs / masks on the caller's stream
            TRY(prologue(0, st));
            {   // ---- side 2: world encoder, then ONE GEMM launch with the visual keys and their gate images
                TRY(prologue(2, sd2));
                TRY(world_encoder(sd2));
                GemmBatch k;
                add_visual(k);
                TRY(k.launch(sd2));
            }
            {   // ---- side 1: the embedding part of the decoder gates
                TRY(prologue(1, sd));
                GemmBatch g;
                add_ge(g);
                TRY(g.launch(sd));
            }
        }
        // command encoder recurrence (seq2seq_model.py:62-88).  With more than one layer (nn.LSTM(num_layers=n), :44-45)
        // a layer below the last writes its h per direction, [B,L,D*He] times the inter-layer dropout mask: the next
        // layer's input; the direction sums and the final state come from the last layer (:76-82).
        const int NL = given ? 0 : enc_layers(d);
        if (given) {
            GSCAN_HIP(hipMemcpyAsync(w + ws.enc_out, given->enc_out, sizeof(float) * (size_t)B * L * He,
                                     hipMemcpyDeviceToDevice, st));
            GSCAN_HIP(hipMemcpyAsync(w + ws.hN, given->hN, sizeof(float) * (size_t)B * He, hipMemcpyDeviceToDevice, st));
        }
        const int64_t lay_h = (int64_t)B * L * D * He, lay_g = 4 * lay_h, lay_img = (int64_t)D * 4 * He * He;
        for (int l = 0; l < NL; ++l) {
            const EncLayer q = enc_layer(p, l);
            const bool last = l == NL - 1;
            GSCAN_CHECK(q.w_ih && q.w_hh && q.b_ih && q.b_hh && (D == 1 || (q.w_ih_rev && q.w_hh_rev && q.b_ih_rev && q.b_hh_rev)),
                        "forward: parameters of encoder layer %d are missing", l);
            const float *image = w + ws.enc_w_image;                 // layer 0: written by the prologue
            if (l > 0) {
                const int Din = D * He;
                const float *x = w + ws.deep_y + (l - 1) * lay_h;    // the layer below, already dropped out
                GemmBatch g;
                g.add(B * L, 4 * He, Din, x, Din, 1, q.w_ih, 1, Din, w + ws.gx, (int64_t)D * 4 * He, 0.f, q.b_ih);
                if (D == 2)
                    g.add(B * L, 4 * He, Din, x, Din, 1, q.w_ih_rev, 1, Din, w + ws.gx + 4 * He, (int64_t)D * 4 * He, 0.f,
                          q.b_ih_rev);
                TRY(g.launch(st));
                float *img = w + ws.deep_image + (l - 1) * lay_img;
                if (encoder_fast_supported(He, L, 0)) TRY(encoder_weight_image(q.w_hh, q.w_hh_rev, He, D, img, st));
                image = img;
            }
            // the first layer projects its own input (the embedded command, E floats per token) inside the recurrent
            // kernel: no launch between the prologue and the recurrence; deeper layers read the GEMM above
            const EncInput own{w + ws.xe, E, q.w_ih, q.b_ih, q.w_ih_rev, q.b_ih_rev, w + ws.wih_t};
            const EncInput *input = l == 0 ? &own : nullptr;
            if (last) {
                TRY(encoder_lstm_forward(B, L, He, D, w + ws.gx, bt.cmd_lengths, q.w_hh, q.b_hh, q.w_hh_rev, q.b_hh_rev,
                                         w + ws.enc_out, w + ws.hN, w + ws.enc_gates, w + ws.enc_cells, w + ws.enc_hprev,
                                         image, st, nullptr, nullptr, input));
            } else {
                TRY(encoder_lstm_forward(B, L, He, D, w + ws.gx, bt.cmd_lengths, q.w_hh, q.b_hh, q.w_hh_rev, q.b_hh_rev,
                                         nullptr, nullptr, w + ws.deep_gates + l * lay_g, w + ws.deep_cells + l * lay_h,
                                         w + ws.deep_hprev + l * lay_h, image, st, w + ws.deep_y + l * lay_h,
                                         mk.enc_deep ? mk.enc_deep + l * lay_h : nullptr, input));
            }
        }
        {
            GemmBatch g;
            add_textual(g);
            if (merged) { add_visual(g); add_ge(g); }
            TRY(g.launch(st));
        }
    }
    if (merged) return 0;
    TRY(order_after(st, sd));          // join: the decoder needs all three branches
    TRY(order_after(st, sd2));
    return 0;
}

int step_forward(const gscan_dims &d, const gscan_params &p, const gscan_batch &bt, const gscan_masks &mk,
                 float *w, float *logp, float *aux_logp, hipStream_t st, const float *given_feat,
                 const float *given_enc_out, const float *given_hN) {
    TRY(check_dims(d));
    Workspace ws;
    TRY(workspace_layout(d, &ws));
    const int B = d.B, H = d.H;
    const bool cond = d.conditional != 0;
    GSCAN_CHECK(logp != nullptr, "forward: logp is NULL");
    GSCAN_CHECK(!d.auxiliary || aux_logp, "forward: auxiliary task set but aux_logp is NULL");
    const GivenEncodings given{given_feat, given_enc_out, given_hN};
    const bool have = given_feat || given_enc_out || given_hN;
    GSCAN_CHECK(!have || (given_feat && given_enc_out && given_hN), "decode_batched: all three encodings must be given");
    TRY(encode_branches(d, p, bt, mk, w, ws, true, st, have ? &given : nullptr));

    // ---- the T-step recurrence; its epilogue is the output head, which does not feed back
    // (seq2seq_model.py:421-424 as the one matrix W_h2o . W_o2h) and log_softmax (model.py:203, :166-170) of the row's
    // T steps, and the auxiliary log_softmax over the summed visual attention (model.py:205)
    DecoderArgs a = decoder_args(d, p, bt, w, ws);
    a.w_image = w + ws.dec_w_fwd;
    a.logp_out = logp;
    a.aux_out = d.auxiliary ? aux_logp : nullptr;
    a.stamps = probe_stamps_enabled() ? w + ws.stamps : nullptr;
    TRY(decoder_run(false, B, H, cond, a, st));
    return 0;
}

// One training step's forward pass, loss and backward pass (train.py:96-110) as one call: the prelude, the decoder's two
// recurrences, the rest of the backward pass.  (Round 3 also ran the two recurrences as ONE launch behind this entry:
// correct, and slower — one register allocation for two loops that each fill the file; DESIGN.md 6.  Removed.)
int step_train_nll(const gscan_dims &d, const gscan_params &p, const gscan_batch &bt, const gscan_masks &mk, float *w,
                   float *logp, float *aux_logp, const NllSeed &nll, const gscan_params &g, hipStream_t st) {
    TRY(step_forward(d, p, bt, mk, w, logp, aux_logp, st));
    return step_backward(d, p, bt, mk, w, nullptr, nullptr, nullptr, &nll, g, st);
}

// --------------------------------------------------------------------------------------
// greedy decoding (predict.py:82-115): encode once, then one decoder step per call with the caller's token
// --------------------------------------------------------------------------------------
int step_encode(const gscan_dims &d, const gscan_params &p, const gscan_batch &bt, const gscan_masks &mk, float *w,
                hipStream_t st) {
    TRY(check_dims(d));
    GSCAN_CHECK(d.T == 1, "encode: dims.T must be 1 (the workspace is laid out for single decoder steps), got %d", d.T);
    Workspace ws;
    TRY(workspace_layout(d, &ws));
    return encode_branches(d, p, bt, mk, w, ws, false, st);
}

int step_decode_one(const gscan_dims &d, const gscan_params &p, const gscan_batch &bt, const int64_t *tokens,
                    const float *h_in, const float *c_in, float *w, float *logits, float *h_out, float *c_out,
                    float *alpha_text, float *alpha_vis, hipStream_t st) {
    TRY(check_dims(d));
    GSCAN_CHECK(d.T == 1, "decode_step: dims.T must be 1, got %d", d.T);
    Workspace ws;
    TRY(workspace_layout(d, &ws));
    const int B = d.B, H = d.H, V = d.V;
    // e = Emb_dec[token] (eval mode: no dropout), then its part of the gate pre-activations
    TRY(embed_rows(tokens, p.dec_emb, V, nullptr, B, H, w + ws.S, 4 * H, st));
    TRY(gemm_f32(B, 4 * H, H, 1.f, w + ws.S, 4 * H, 1, p.dec_w_ih, 1, 3 * H, 0.f, w + ws.ge, 4 * H, w + ws.bsum, 0,
                 nullptr, 1, st));
    gscan_batch step_batch = bt;
    step_batch.targets = tokens;
    DecoderArgs a = decoder_args(d, p, step_batch, w, ws);
    a.w_image = w + ws.dec_w_fwd;
    a.hprev = const_cast<float *>(h_in);       // T = 1: row 0 is read, nothing is written
    a.c0 = c_in;
    a.h_last = h_out;
    a.cells = c_out;
    a.logits = logits;
    a.alpha_c = alpha_text;
    a.alpha_s = alpha_vis;
    a.logp_out = w + ws.logp_saved;
    a.aux_saved = nullptr; a.aux_out = nullptr; a.row_stats = nullptr;
    a.stamps = nullptr;
    return decoder_run(false, B, H, d.conditional != 0, a, st);
}

// Greedy decoding of a whole batch in ONE launch (predict.py:82-115): encode, two tiny table products, then the
// persistent decoder with the argmax fed back in-kernel, every row until its own <EOS> or max_steps steps.
int step_greedy(const gscan_dims &d, int max_steps, const gscan_params &p, const gscan_batch &bt, float *w, int sos,
                int eos, int64_t *tokens, int32_t *steps, float *alpha_text, float *alpha_vis, float *att_sum,
                hipStream_t st) {
    TRY(check_dims(d));
    GSCAN_CHECK(d.T == 1, "greedy_decode: dims.T must be 1 (the workspace holds no per-step activations), got %d", d.T);
    GSCAN_CHECK(max_steps > 0 && tokens && steps && alpha_text && alpha_vis && att_sum, "greedy_decode: bad arguments");
    GSCAN_CHECK(sos >= 0 && sos < d.V && eos >= 0 && eos < d.V, "greedy_decode: <SOS>/<EOS> outside the vocabulary");
    Workspace ws;
    TRY(workspace_layout(d, &ws));
    gscan_masks none{};
    TRY(encode_branches(d, p, bt, none, w, ws, false, st));
    const int B = d.B, H = d.H, V = d.V;
    {
        GemmBatch g;
        g.add(V, 4 * H, H, p.dec_emb, H, 1, p.dec_w_ih, 1, 3 * H, w + ws.ge_table, 4 * H, 0.f, w + ws.bsum);
        TRY(g.launch(st));
    }
    DecoderArgs a = decoder_args(d, p, bt, w, ws);
    a.T = max_steps;
    a.w_image = w + ws.dec_w_fwd;
    a.ge = w + ws.ge_table;
    a.dec_emb = p.dec_emb;
    a.sos = sos; a.eos = eos;
    a.tokens_out = tokens; a.steps_out = steps;
    a.alpha_c = alpha_text; a.alpha_s = alpha_vis; a.att_sum = att_sum;
    a.aux_saved = nullptr; a.aux_out = nullptr; a.row_stats = nullptr; a.stamps = nullptr;
    return decoder_run(false, B, H, d.conditional != 0, a, st);
}

// --------------------------------------------------------------------------------------
// backward: main stream = chain of data gradients; side streams = weight-gradient leaves (the convolution's
// gradient product + fold on its own stream: it is long and nothing but the optimiser waits for it)
// --------------------------------------------------------------------------------------
int step_backward(const gscan_dims &d, const gscan_params &p, const gscan_batch &bt, const gscan_masks &mk, float *w,
                  const float *dlogp, const float *daux, const float *seeds, const NllSeed *nll, const gscan_params &g,
                  hipStream_t st) {
    TRY(check_dims(d));
    Workspace ws;
    TRY(workspace_layout(d, &ws));
    const int B = d.B, L = d.L, T = d.T, M = d.G * d.G, C = d.C, Co = d.Co, F = 3 * Co, E = d.E, He = d.He, H = d.H,
              V = d.V, D = d.bidirectional ? 2 : 1;
    const int BT = B * T, BL = B * L, BM_ = B * M;
    const bool cond = d.conditional != 0;
    const bool reduced = attention_time_reduced(T, L, M);      // long targets: sums over time per memory (above workspace_layout)
    const int gt_w = cond ? 5 * H : 4 * H;                     // columns of G_text: [delta | dzq]
    GSCAN_CHECK(dlogp || nll, "backward: dlogp is NULL");
    TRY(check_drop(mk));
    const DropSpec drop_cnn = drop_of(mk, kDropSegCnn), drop_enc = drop_of(mk, kDropSegEnc), drop_dec = drop_of(mk, kDropSegDec);
    // deterministic mode (GSCAN_DETERMINISTIC=1): every sum formed across workgroups in a fixed order — split-K partial
    // tiles through slabs (gemm_mt.hip), embedding and convolution-bias gradients by their ordered kernels
    const bool ordered_sums = gemm_macro_tile_mode() > 0;
    TRY(side_init());
    hipStream_t sd = g_side.single ? st : g_side.stream, sd2 = g_side.single ? st : g_side.stream2;
    float *S = w + ws.S, *dS = w + ws.dS;
    const float *delta = w + ws.delta, *hprev = w + ws.hprev;

    // ---- reverse recurrence (occupies every CU: nothing overlaps it).  Its prologue is the backward of the head
    // (log_softmax, hidden_to_output, output_to_hidden) for the row's T steps; seeds (optional, device) multiply
    // dlogp by seeds[0] and daux by seeds[1]
    // The convolution gradients' lists of non-zeros depend on the batch alone and their leaf ends the step (it starts
    // behind keys_backward and runs longest): built here on the second leaf stream, forked BEFORE the reverse
    // recurrence, they run whenever the chip has room — normally as the recurrence drains — and are long done when
    // the convolution gradients start.  (Built on the caller's stream, or as passengers of the forward world encoder,
    // they cost the critical chain what they saved: DESIGN.md 6.)
    static const int early_lists = [] { const char *e = getenv("GSCAN_EARLY_LISTS"); return e ? atoi(e) : 1; }();
    if (early_lists) {
        TRY(order_after(sd2, st));
        TRY(world_conv_lists(bt.world_u8 ? (const void *)bt.world_u8 : (const void *)bt.world, bt.world_u8 != nullptr, B, d.G,
                             C, w + ws.conv_lists, sd2));
    }
    const bool use_aux = d.auxiliary && daux;
    DecoderArgs a = decoder_args(d, p, bt, w, ws);
    a.dlogp = dlogp; a.daux = use_aux ? daux : nullptr; a.seeds = seeds;
    if (nll) { a.nll_mode = nll->sum ? 2 : 1; a.w_aux = nll->w_aux; a.stats_out = nll->stats_out; a.seeds_out = nll->seeds_out; }
    a.dlogits = w + ws.dlogits; a.ds = dS;
    a.delta = w + ws.delta; a.dqt = w + ws.dqt; a.dqv = w + ws.dqv;
    a.dpk_t = w + ws.dpk_t; a.dpk_v = w + ws.dpk_v; a.dv_t = w + ws.dv_t; a.dv_v = w + ws.dv_v;
    a.dh0 = w + ws.dh0;
    a.w_image = w + ws.dec_w_bwd;
    a.stamps = probe_stamps_enabled() ? w + ws.stamps + 16 : nullptr;
    TRY(decoder_run(true, B, H, cond, a, st));
    // where the decoder's weight-gradient leaves fork off the chain (GSCAN_LEAVES_FORK, A/B): 0 behind the recurrence,
    // 1 behind the dS += product, 2 behind keys_backward (one fork event in the whole backward pass)
    static const int leaves_fork = [] { const char *e = getenv("GSCAN_LEAVES_FORK"); return e ? atoi(e) : 0; }();
    // GSCAN_LEAVES_SPLIT=1 (A/B, off): the leaves as TWO launches, the second one held back until the command encoder's
    // reverse recurrence has run.  One launch of K = B T products is thousands of 128-VGPR workgroups that take every freed
    // slot, and the recurrence — 254 VGPRs, a whole CU's registers per pair of workgroups — does not get onto the chip before
    // they have drained (S3 timelines, profiles/r05_time_reduced_attention_gradients_S3_ab.txt).  Measured at S3: the chain's
    // kernels run undisturbed (keys backward 101 -> 66 us, reverse recurrence 35 -> 28) and the step is SLOWER, 1.664 -> 1.685
    // ms: the held-back half (2.1 GMAC) becomes the tail — this stretch is bound by the GEMM work, not by who runs first.
    // part: 0 = everything, 1 = first half, 2 = second half.
    static const bool leaves_split = [] { const char *e = getenv("GSCAN_LEAVES_SPLIT"); return e && atoi(e) > 0; }();
    auto decoder_leaves = [&](int part = 0) -> int {
        if (part == 0 && leaves_split) part = 1;
        const bool early = part != 2, late = part != 1;
        {   // leaves: head weights (the permuted W_o2h gradient is scattered back below) and the decoder parameter
            // gradients (dense products over the B*T saved rows)
            GemmBatch b;
            if (gemm_macro_tile_mode() > 0) b.scratch(w + ws.gemm_slabs_side, (size_t)kGemmSlabs * gemm_slab_floats());
            // (the leaves on the macro tiles, GSCAN_LEAVES_MT=<rows>, measured for S3: the family's time drops 0.78 ->
            // 0.65 ms and the step gets SLOWER, 1.777 -> 1.822 ms — their workgroups crowd the chain's kernels)
            static const int leaves_mt = [] { const char *e = getenv("GSCAN_LEAVES_MT"); return e ? atoi(e) : 0; }();
            if (leaves_mt > 0 && BT >= leaves_mt) b.prefer_macro_tiles();
            static const int dec_split = [] { const char *e = getenv("GSCAN_SPLIT_DEC"); return e ? atoi(e) : 0; }();
            g_split_override = dec_split;
            // the head as one matrix Wc = W_h2o . W_o2h (decoder.hip): d Wc = dlogits^T . S, and both Linears'
            // gradients follow from it in head_grad_finish below.  These V x 4H x BT multiply-adds stand in for the
            // reference's two products (H x 4H and V x H over the same BT rows), whose algorithmic flops (SURVEY.md
            // 8d) the launch is credited with.
            if (early) {
            add_grad(b, V, 4 * H, BT, w + ws.dlogits, 1, V, S, 4 * H, 1, w + ws.dwc, 4 * H);
            b.credit(2.0 * BT * ((double)H * 4 * H + (double)V * H) - 2.0 * BT * (double)V * 4 * H);
            if (reduced) {     // context columns of dW_ih (and of dW_q2k below) from the sums over time: K = B (L + G^2), not B T
                add_grad(b, 4 * H, H, BT, delta, 1, 5 * H, S, 4 * H, 1, g.dec_w_ih, 3 * H);
                add_grad(b, 4 * H, H, BL, w + ws.g_t, 1, gt_w, w + ws.pkt, H, 1, g.dec_w_ih + H, 3 * H);
                add_grad(b, 4 * H, H, BM_, w + ws.g_v, 1, 4 * H, w + ws.pkv, H, 1, g.dec_w_ih + 2 * H, 3 * H);
                b.credit(2.0 * 4 * H * 2 * H * (double)BT - 2.0 * 4 * H * H * (double)(BL + BM_));
            } else {
                add_grad(b, 4 * H, 3 * H, BT, delta, 1, 5 * H, S, 4 * H, 1, g.dec_w_ih, 3 * H);
            }
            add_grad(b, 4 * H, H, BT, delta, 1, 5 * H, hprev, H, 1, g.dec_w_hh, H, g.dec_b_ih, g.dec_b_hh);
            if (cond && reduced) {
                add_grad(b, H, H, BL, w + ws.g_t + 4 * H, 1, gt_w, w + ws.pkt, H, 1, g.q2k_w + H, 2 * H);
                b.credit(2.0 * H * H * (double)(BT - BL));
            }
            }
            if (late) {
            add_grad(b, H, H, BT, w + ws.dqt, 1, H, hprev, H, 1, g.txt_query_w, H);
            if (cond) {
                add_grad(b, H, H, BT, delta + 4 * H, 1, 5 * H, hprev, H, 1, g.q2k_w, 2 * H, g.q2k_b);
                if (!reduced) add_grad(b, H, H, BT, delta + 4 * H, 1, 5 * H, S + H, 4 * H, 1, g.q2k_w + H, 2 * H);
                add_grad(b, H, H, BT, w + ws.dqv, 1, H, w + ws.q2, H, 1, g.vis_query_w, H);
            } else {
                add_grad(b, H, H, BT, w + ws.dqv, 1, H, hprev, H, 1, g.vis_query_w, H);
            }
            // gradient wrt the embedded target token (the e columns of dS): only the embedding table consumes it, so
            // this third of the LSTM-input back-propagation is a leaf too
            g_split_override = 0;
            b.add(BT, H, 4 * H, delta, 5 * H, 1, w + ws.wcat5, 3 * H, 1, dS, 4 * H, 1.f);
            }
            g_split_override = 0;
            TRY(b.launch(sd));
            if (early)
                TRY(head_grad_finish(w + ws.dwc, p.hid2out_w, p.out2hid_w, g.out2hid_w, g.hid2out_w, H, V, sd, w + ws.dv_t,
                                     w + ws.dv_v, B, g.txt_energy_w, g.vis_energy_w));
            // (dropout drawn in the kernels: the gather of the forward pass left its keep values in the workspace)
            if (late)
                TRY(embed_grad(bt.targets, dS, 4 * H, mk.in_kernel ? (drop_dec.on ? w + ws.drawn_mask_dec : nullptr) : mk.dec, BT, H,
                               V, d.pad_tgt, g.dec_emb, sd, ordered_sums ? w + ws.embed_part_dec : nullptr));
        }
        return 0;
    };
    KeysBackwardArgs k{};
    k.T = T; k.L = L; k.M = M; k.He = He; k.F = F;
    k.alpha_c = w + ws.alpha_c; k.alpha_s = w + ws.alpha_s; k.ds = dS;
    k.dpk_t = w + ws.dpk_t; k.dpk_v = w + ws.dpk_v; k.dh0 = w + ws.dh0;
    k.w_kt = p.txt_key_w; k.w_kv = p.vis_key_w; k.w_b = p.bridge_w;
    k.feat = w + ws.feat; k.mask = mk.in_kernel ? nullptr : mk.cnn;
    k.mask_scale = drop_cnn.on ? drop_cnn.scale : 0.f;
    k.denc = w + ws.denc; k.dhN = w + ws.dhN; k.dfeat = w + ws.dfeat;
    float *const gw[3] = {g.conv1_w, g.conv2_w, g.conv3_w};
    float *const gb[3] = {g.conv1_b, g.conv2_b, g.conv3_b};
    const void *world = bt.world_u8 ? (const void *)bt.world_u8 : (const void *)bt.world;
    if (reduced) {
        // long targets: sums over time per memory first (the leaves need them too, so they fork behind this launch), then
        // d PK += G . [W_ih[:, ctx] ; W_q2k[:, ctx_text]] — B (L + G^2) rows through the weights instead of B T; keys_backward
        // below adds what the output head hands to the contexts (dS from the reverse kernel's prologue) as before
        AlphaReduceArgs r{};
        r.T = T; r.L = L; r.M = M; r.wt = gt_w; r.wv = 4 * H; r.ldx = 5 * H;
        r.alpha_c = w + ws.alpha_c; r.alpha_s = w + ws.alpha_s; r.x = delta; r.g_t = w + ws.g_t; r.g_v = w + ws.g_v;
        TRY(alpha_reduce(B, r, st));
        if (leaves_fork == 0) { TRY(order_after(sd, st)); TRY(decoder_leaves()); }
        GemmBatch b;
        b.add(BL, H, gt_w, w + ws.g_t, gt_w, 1, w + ws.wcat5 + H, 3 * H, 1, w + ws.dpk_t, H, 1.f);
        b.add(BM_, H, 4 * H, w + ws.g_v, 4 * H, 1, w + ws.wcat5 + 2 * H, 3 * H, 1, w + ws.dpk_v, H, 1.f);
        b.credit(2.0 * BT * 2 * H * (double)gt_w - 2.0 * H * ((double)BL * gt_w + (double)BM_ * 4 * H));
        TRY(b.launch(st));
    } else {
    if (leaves_fork == 0) { TRY(order_after(sd, st)); TRY(decoder_leaves()); }
    // chain: gradient wrt [ctx_text | ctx_vis] through the LSTM input and the conditional query
    // (one product: [delta | dzq] . [W_ih[:, ctx] ; (W_q2k[:, ctx_text] | 0)], K = 5H when conditional)
    {
        // From ~10 000 decoder rows on (S3: 30 720; S1 from 512 rows per GPU) this product — alone in its launch, on the
        // critical chain, 500 deep — is faster on the macro tiles of gemm_mt.hip: S3 1.777 -> 1.750 ms per step, S1 at
        // 512 / 1 024 rows 0.909 -> 0.891 / 1.740 -> 1.713; at 5 120 / 7 680 rows it loses 1 %
        // (profiles/r04_ds_product_macro_tiles_ab.txt).  GSCAN_DS_MT=<rows> moves the threshold, 0 = never.
        static const int ds_mt = [] { const char *e = getenv("GSCAN_DS_MT"); return e ? atoi(e) : 10000; }();
        GemmBatch b;
        if (ds_mt > 0 && BT >= ds_mt) b.prefer_macro_tiles();
        b.add(BT, 2 * H, cond ? 5 * H : 4 * H, delta, 5 * H, 1, w + ws.wcat5 + H, 3 * H, 1, dS + H, 4 * H, 1.f);
        TRY(b.launch(st));
    }
    }
    if (leaves_fork == 1) { TRY(order_after(sd, st)); TRY(decoder_leaves()); }
    // chain: value path of both attentions (dPK[b,m,:] += sum_t alpha[b,t,m] * dctx[b,t,:]), then through the
    // key layers and the bridge to the encoder outputs / final state / conv features — one launch, row per WG
    // GSCAN_KEYS_GEMM=1 (round 6 A/B): the kernel stops behind the value path (dPK totals written out) and the key layers
    // and the bridge follow as ONE grouped-GEMM launch on the chain — d feat = (dPK_vis . W_key_vis) with the ReLU gate and
    // the dropout scale as the epilogue, d enc_out = dPK_text . W_key_text, d h_N = d h0 . W_bridge.  The fused kernel
    // re-reads W_key (60 KB) per (row, 16 memories): 82 MB of L2 -> CU traffic per launch for 0.26 GMAC.
    static const int keys_gemm = [] { const char *e = getenv("GSCAN_KEYS_GEMM"); return e ? atoi(e) : 0; }();
    const bool keys_split = keys_gemm != 0 && k.mask == nullptr;      // a mask in memory stays on the fused path
    k.value_path_only = keys_split ? 1 : 0;
    TRY(keys_backward(B, H, k, st));
    if (keys_split) {
        GemmBatch b;
        const float scale = k.mask_scale != 0.f ? k.mask_scale : 1.f;
        b.add(BM_, F, H, w + ws.dpk_v, H, 1, p.vis_key_w, F, 1, w + ws.dfeat, F, 0.f, nullptr, 3, nullptr, 1, nullptr, nullptr,
              w + ws.feat, scale);
        b.add(BL, He, H, w + ws.dpk_t, H, 1, p.txt_key_w, He, 1, w + ws.denc, He);
        b.add(B, He, H, w + ws.dh0, H, 1, p.bridge_w, He, 1, w + ws.dhN, He);
        TRY(b.launch(st));
    }
    TRY(order_after(sd, st, sd2));     // one event releases both leaf streams
    if (leaves_fork == 2) TRY(decoder_leaves());
    {   // leaves: key and bridge weights
        GemmBatch b;
        if (gemm_macro_tile_mode() > 0)      // same stream as the decoder leaves: the same region
            b.scratch(w + ws.gemm_slabs_side, (size_t)kGemmSlabs * gemm_slab_floats());
        add_grad(b, H, He, BL, w + ws.dpk_t, 1, H, w + ws.enc_out, He, 1, g.txt_key_w, He);
        add_grad(b, H, He, B, w + ws.dh0, 1, H, w + ws.hN, He, 1, g.bridge_w, He, g.bridge_b);
        add_grad(b, H, F, BM_, w + ws.dpk_v, 1, H, w + ws.feat, F, 1, g.vis_key_w, F);
        TRY(b.launch(sd));
    }
    {   // leaf: convolution kernel and bias gradients from the non-zeros of the world and d(features) (conv.hip)
        // (measured twice: building the lists earlier does not pay.  Right behind the recurrence on this stream: 0.547
        // -> 0.572 ms.  As passenger workgroups of the forward pass's world-encoder launch: that launch 10.6 -> 15.2 us,
        // and the 17 us this leaf starts earlier are lost again because it then overlaps the encoder's weight-gradient
        // launch, which stretches 19 -> 41 us: this stretch of the step is throughput-bound, 0.541 vs 0.538 ms.)
        if (!early_lists) TRY(world_conv_lists(world, bt.world_u8 != nullptr, B, d.G, C, w + ws.conv_lists, sd2));
        TRY(world_conv_backward(w + ws.dfeat, B, d.G, C, Co, d.K3, w + ws.conv_lists, gw, gb, sd2,
                                ordered_sums ? w + ws.conv_bias_part : nullptr));
    }
    // ---- command encoder BPTT (chain), last layer first.  Per layer: the reverse recurrence, then ONE launch on the
    // caller's stream (the leaf streams are still busy with the key / conv gradients and would finish last
    // otherwise) with the layer's weight gradients for both directions and the gradient wrt its input: the embedded
    // command for layer 0 (K = 8He split eight ways onto the zeroed buffer), the layer below's [B,L,D*He] output
    // otherwise (one product per direction, both accumulating with atomics onto the zeroed buffer).
    const int NL = enc_layers(d);
    const int64_t ldd = (int64_t)D * 4 * He, lay_h = (int64_t)B * L * D * He, lay_g = 4 * lay_h;
    if (NL > 1) GSCAN_HIP(hipMemsetAsync(w + ws.deep_dy, 0, (size_t)(NL - 1) * lay_h * sizeof(float), st));
    for (int l = NL - 1; l >= 0; --l) {
        const EncLayer q = enc_layer(p, l), gq = enc_layer(g, l);
        const bool last = l == NL - 1;
        const float *lgates = last ? w + ws.enc_gates : w + ws.deep_gates + l * lay_g;
        const float *lcells = last ? w + ws.enc_cells : w + ws.deep_cells + l * lay_h;
        const float *lhprev = last ? w + ws.enc_hprev : w + ws.deep_hprev + l * lay_h;
        float *ldelta = last ? w + ws.enc_delta : w + ws.deep_delta + l * lay_g;
        if (last) {
            TRY(encoder_lstm_backward(B, L, He, D, bt.cmd_lengths, q.w_hh, q.w_hh_rev, lgates, lcells, w + ws.denc,
                                      w + ws.dhN, ldelta, st));
            if (leaves_split) { TRY(order_after(sd, st)); TRY(decoder_leaves(2)); }     // the held-back half of the decoder's leaves
        } else {
            TRY(encoder_lstm_backward(B, L, He, D, bt.cmd_lengths, q.w_hh, q.w_hh_rev, lgates, lcells,
                                      w + ws.deep_dy + l * lay_h, nullptr, ldelta, st, D * He, He,
                                      mk.enc_deep ? mk.enc_deep + l * lay_h : nullptr));
        }
        const int Din = l == 0 ? E : D * He;
        const float *x = l == 0 ? w + ws.xe : w + ws.deep_y + (l - 1) * lay_h;
        GemmBatch b;
        if (gemm_macro_tile_mode() > 0) b.scratch(w + ws.gemm_slabs_main, (size_t)kGemmSlabs * gemm_slab_floats());
        // this launch ends the step on an otherwise idle chip: its split is its own knob (GSCAN_SPLIT_ENC)
        static const int enc_split = [] { const char *e = getenv("GSCAN_SPLIT_ENC"); return e ? atoi(e) : 0; }();
        g_split_override = enc_split;
        for (int dir = 0; dir < D; ++dir) {
            const float *dl = ldelta + dir * 4 * He;
            float *gw_ih = dir ? gq.w_ih_rev : gq.w_ih, *gw_hh = dir ? gq.w_hh_rev : gq.w_hh;
            float *gb_ih = dir ? gq.b_ih_rev : gq.b_ih, *gb_hh = dir ? gq.b_hh_rev : gq.b_hh;
            add_grad(b, 4 * He, He, BL, dl, 1, ldd, lhprev + dir * He, (int64_t)D * He, 1, gw_hh, He, gb_ih, gb_hh);
            add_grad(b, 4 * He, Din, BL, dl, 1, ldd, x, Din, 1, gw_ih, Din);
            if (l > 0)
                // both directions add into the same buffer from one launch: atomics even when K = 4He is too short to split
                b.add(BL, Din, 4 * He, dl, ldd, 1, dir ? q.w_ih_rev : q.w_ih, Din, 1, w + ws.deep_dy + (l - 1) * lay_h, Din,
                      1.f, nullptr, 0, nullptr, -2);
        }
        if (l == 0)
            b.add(BL, E, D * 4 * He, ldelta, ldd, 1, w + ws.wih_stack, E, 1, w + ws.dxe, E, 1.f, nullptr, 0, nullptr, 8);
        g_split_override = 0;
        TRY(b.launch(st));
    }
    TRY(embed_grad(bt.commands, w + ws.dxe, E, mk.in_kernel ? (drop_enc.on ? w + ws.drawn_mask_enc : nullptr) : mk.enc, BL, E,
                   d.Vi, d.pad_in, g.enc_emb, st, ordered_sums ? w + ws.embed_part_enc : nullptr));
    // join: every gradient is complete when the caller's stream continues.  Side 1 (done long before) waits for side 2,
    // the caller's stream for side 1: ONE wait packet in front of the optimiser instead of two (each costs the queue
    // ~2.5 us even when its event completed long ago)
    // the first leaf stream's gradients (decoder, attentions, bridge, conditional query: everything but the convolution
    // kernels and the command encoder) are complete where this event sits: a data-parallel caller's communication stream
    // waits for it (gscan_early_gradients_wait) and starts its first all-reduce under the tail of this pass
    if (g_early_comm && g_early_n) TRY(comm_allreduce_f32(g_early_comm, g_early_buf, g_early_n, sd));
    if (sd != st) {
        GSCAN_HIP(hipEventRecord(g_side.early, sd));
        g_side.early_recorded = true;
    } else {
        g_side.early_recorded = false;       // single-stream mode: nothing completes early
    }
    TRY(order_after(sd, sd2));
    TRY(order_after(st, sd));
    return 0;
}

int set_early_allreduce(void *comm, float *buf, size_t n) {
    GSCAN_CHECK(comm == nullptr || (buf != nullptr && n > 0), "set_early_allreduce: a communicator needs a buffer range");
    g_early_comm = comm; g_early_buf = buf; g_early_n = comm ? n : 0;
    return 0;
}

int early_gradients_wait(hipStream_t stream) {
    TRY(side_init());
    GSCAN_CHECK(g_side.early_recorded || g_side.single, "early_gradients_wait: no backward pass has been issued in this process yet");
    if (g_side.early_recorded) GSCAN_HIP(hipStreamWaitEvent(stream, g_side.early, 0));
    return 0;
}

}  // namespace gscan

// Step-level driver: sequences the op-level launchers for one training / evaluation step from C++ (a few microseconds of
// host time per launch instead of a Python round trip) on the caller's stream plus, when the context carries one, an
// aux stream joined by events — no host synchronisation, no allocation.  tcar_train_step == sess.run([loss, global_step, train_op]) (model_combine.py:231);
// tcar_eval_step == sess.run([softmax_input, cross_loss]) + util.cau_metrics + top-k (model_combine.py:283,296,301).
#include "tcar_common.h"
#include <stdlib.h>

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}
// ONE table: name, field, shipped default (a switch cannot be added out of order)
namespace {
struct Switch { const char* name; int32_t TcarTuning::*field; int dflt; };
const Switch kSwitches[] = {
    {"TCAR_BF16_TILE", &TcarTuning::bf16_tile, 0},          {"TCAR_BF16_KS", &TcarTuning::bf16_ks, 2},
    {"TCAR_WGRAD_KS", &TcarTuning::wgrad_ks, 1536},         {"TCAR_GATHER_BIG_ROWS", &TcarTuning::gather_big_rows, 16384},
    {"TCAR_MHA_MFMA", &TcarTuning::mha_mfma, 1},            {"TCAR_SORT_SCATTER", &TcarTuning::sort_scatter, 1},
    {"TCAR_DET_SMALL", &TcarTuning::det_small, 1},          {"TCAR_FUSED_CE", &TcarTuning::fused_ce, 2},
    {"TCAR_ONEHOT_TIME", &TcarTuning::onehot_time, 2},      {"TCAR_FLAG_FORK", &TcarTuning::flag_fork, 4095},
    {"TCAR_CE_FOLD", &TcarTuning::ce_fold, 1024},           {"TCAR_PROJ_SPLIT_ROWS", &TcarTuning::proj_split_rows, 1024},
};
static_assert(sizeof(kSwitches) / sizeof(kSwitches[0]) == 12 && sizeof(TcarTuning) == 12 * sizeof(int32_t), "twelve switches, one table");
}  // namespace
// the process snapshot: written once by the initialiser of this function-local static, const ever after
const TcarTuning& tcar_tuning() {
  static const TcarTuning t = [] {
    TcarTuning x{};
    for (const Switch& sw : kSwitches) x.*(sw.field) = env_int(sw.name, sw.dflt);
    return x;
  }();
  return t;
}
extern "C" int tcar_tuning_defaults(tcar_tuning_t* out) {
  if (!out) return TCAR_E_ARG;
  *out = tcar_tuning();
  return TCAR_OK;
}
// sets one switch in the CALLER's copy (tests and tools build a context with it); returns the previous value, INT_MIN for an
// unknown name
extern "C" int tcar_tuning_set(tcar_tuning_t* t, const char* name, int value) {
  if (!t || !name) return -2147483647 - 1;
  for (const Switch& sw : kSwitches) {
    bool same = true;
    for (int i = 0; same; ++i) {
      if (sw.name[i] != name[i]) same = false;
      else if (!sw.name[i]) break;
    }
    if (same) { const int old = t->*(sw.field); t->*(sw.field) = value; return old; }
  }
  return -2147483647 - 1;
}

namespace {

#define RET(x)            \
  do {                    \
    int rc__ = (x);       \
    if (rc__) return rc__; \
  } while (0)

struct Geo {
  int N, Npad, H, ldh, ldt, ic, pt, ct, ek;
  explicit Geo(const tcar_dims_t& d)
      : N(d.n_items), Npad((d.n_items + 127) / 128 * 128), H(d.H), ldh(d.ldh), ldt(d.ldt), ic(2 * d.ldh), pt(5 * d.ldt),
        ct(2 * d.ldt), ek(2 * d.ldh + 5 * d.ldt) {}
};

inline float* W(const tcar_ctx_t* c, int v) { return c->W + c->off[v]; }
inline int units(int K) { return (K + 127) / 128; }     // 128-deep K chunks of a split small GEMM (two 64-deep stages each)
inline float* G(const tcar_ctx_t* c, int v) { return c->Gx + c->off[v]; }

tcar_gemm_desc_t prob(int M, int N, float* C, int64_t ldc, const float* bias = nullptr, int act = 0, int beta = 0,
                      int splitk = 1, int atomic = 0) {
  tcar_gemm_desc_t d = {};
  d.nseg = 0; d.C = C; d.ldc = ldc; d.bias = bias; d.M = M; d.N = N; d.act = act; d.beta = beta; d.splitk = splitk;
  d.atomic = atomic;
  return d;
}
void seg(tcar_gemm_desc_t& d, const float* A, int64_t lda, const float* B, int64_t ldb, int K) {
  const int i = d.nseg++;
  d.A[i] = A; d.lda[i] = lda; d.B[i] = B; d.ldb[i] = ldb; d.K[i] = K;
}
tcar_gemm_desc_t prob1(int M, int N, const float* A, int64_t lda, const float* B, int64_t ldb, int K, float* C,
                       int64_t ldc, const float* bias = nullptr, int act = 0, int beta = 0, int splitk = 1,
                       int atomic = 0) {
  tcar_gemm_desc_t d = prob(M, N, C, ldc, bias, act, beta, splitk, atomic);
  seg(d, A, lda, B, ldb, K);
  return d;
}

void tables_of(const tcar_ctx_t* c, tcar_tables_t& t) {
  t.E = c->E; t.pos = W(c, TCAR_V_POS);
  for (int k = 0; k < 5; ++k) t.time[k] = W(c, TCAR_V_MONTH + k);
  t.dur = W(c, TCAR_V_DUR);
}
void grads_of(const tcar_ctx_t* c, tcar_grads_t& g) {
  g.g_item = c->big; g.g_pos = G(c, TCAR_V_POS);
  for (int k = 0; k < 5; ++k) { g.g_time[k] = G(c, TCAR_V_MONTH + k); g.slot_time[k] = c->slot_of[TCAR_V_MONTH + k]; }
  g.g_dur = G(c, TCAR_V_DUR);
  g.sqn = c->Gx + c->arena_n;
  g.slot_item = c->slot_item; g.slot_pos = c->slot_of[TCAR_V_POS]; g.slot_dur = c->slot_of[TCAR_V_DUR];
  g.rows_out = nullptr; g.norms_out = nullptr; g.rows_ld = 0; g.skip_small = 0;
}

// the small contractions follow the scoring precision: exact fp32 MFMA in "f32" mode, split-bf16 otherwise
// `o`: launch options (context tuning; the completion flag this launch is to carry — split-bf16 form only)
inline int small_gemm(const tcar_ctx_t* c, int layout, int n, const tcar_gemm_desc_t* p, void* stream, TcarOpt* o = nullptr) {
  TcarOpt plain;
  if (!o) { plain.tune = c->tune; o = &plain; }
  return c->scoring ? tcar_gemm_x3_grouped_o(layout, n, p, stream, o) : tcar_gemm_f32_grouped(layout, n, p, stream);
}

// softmax epilogue of the logits GEMM (training steps of the hi-only backward precision): workspace [rowstat 2B | label score B |
// group stats]; the logits are not materialised
struct CeWs { float* rowstat; float* lab; float* stats; int64_t stats_floats; };
inline bool fused_ce(const tcar_ctx_t* c, int B, CeWs* w) {
  if (!c->scoring || c->scoring_bwd != 1 || !c->ce_ws || !(c->tune ? c->tune->fused_ce : tcar_tuning().fused_ce)) return false;
  const int64_t head = 2L * B + ((B + 1) & ~1);
  const int64_t need = (int64_t)B * ((c->d.n_items + 63) / 64 + 8) * 2;
  if (c->ce_ws_floats < head + need) return false;
  if (w) { w->rowstat = c->ce_ws; w->lab = c->ce_ws + 2L * B; w->stats = c->ce_ws + head; w->stats_floats = c->ce_ws_floats - head; }
  return true;
}

inline hipStream_t aux_stream(const tcar_ctx_t* c) {
  return (c->stream2 && c->ev[0] && c->ev[1] && c->ev[2] && c->ev[3] && c->ev[4]) ? (hipStream_t)c->stream2 : nullptr;
}

// ---- forks and joins of the step's streams without an event on the producer's stream ----------------------------------------------
// sig = fork_arm(c, slot) right before the launch whose END the other stream has to wait for; the signal is handed to THAT launch
// explicitly (TcarOpt.sig of its `_o` form) and fork_commit(c, slot, opt) records whether the kernel form it chose carries the
// flag; fork_go(slot, ...) where the event record + wait used to be: a one-wave kernel on the consumer's stream polls the flag.
// A launch that did not carry the flag, a slot masked off in TCAR_FLAG_FORK or a context without the flag words: the event
// pair, as before.  ALL of this state lives in the context's own host block (tcar_ctx_t.fork_host): nothing per thread, nothing
// per process — two contexts stepped alternately, or from two host threads, cannot see each other's forks.
// The poll gives up after POLL_TICKS of the 100-MHz wall clock (a profiler that serialises kernels would otherwise hang) and
// counts the time-out in sig_dev[TCAR_SIG_ERR] AND, with a system-scope atomic, in the context's host-visible word
// (sig_err_host): the engine tests that word after every step without synchronising and raises.
// Visibility across the eight private L2s is the PRODUCER's business: whatever the consumer reads of the flagged launch's own
// output is stored write-through (sc1 / agent-scope atomic stores) or with atomics, everything older was released when its
// launch ended; the consumer kernels behind the poll start with the runtime's usual acquire.  Which kernels may carry a flag,
// and what each stores write-through: TcarOpt in tcar_common.h; a small-GEMM launch with bf16 plane outputs (plain stores)
// REFUSES a flag (TCAR_E_ARG).
constexpr int TCAR_SIG_SLOTS = 16, TCAR_SIG_ERR = 2 * TCAR_SIG_SLOTS;
constexpr long long POLL_TICKS = 100000000LL;      // 1 s of the 100-MHz wall clock
__global__ __launch_bounds__(64) void poll_flag_kernel(const unsigned* flag, unsigned epoch, unsigned* err, unsigned* err_host,
                                                       long long ticks = POLL_TICKS, const unsigned* flag2 = nullptr,
                                                       unsigned epoch2 = 0, int delay_ticks = 0) {
  if (threadIdx.x != 0) return;
  const long long t0 = wall_clock64();
  bool ok = true;
  if (flag2) {                        // a join of two streams: the second flag first, the usual loop below for the first
    while ((int)(__hip_atomic_load(flag2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch2) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > ticks) { ok = false; break; }
    }
  }
  while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
    __builtin_amdgcn_s_sleep(2);
    if (wall_clock64() - t0 > ticks) { ok = false; break; }
  }
  if (delay_ticks > 0) {             // DELAYED fork: hold the consumer back as the event it replaces did
    const long long t1 = wall_clock64();
    while (wall_clock64() - t1 < delay_ticks) __builtin_amdgcn_s_sleep(8);
  }
  if (!ok) {
    atomicAdd(err, 1u);
    if (err_host) __hip_atomic_fetch_add(err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// slots (bits of TCAR_FLAG_FORK).  The early Adam -> candidate refresh fork stays an event: with a flag the step is slower in every
// form tried (write-through producers, delayed polls: profiles/r03_ab_experiments.txt).  The softmax -> dE fork is the START flag of
// the dX GEMM since round 6 (slot 8: neutral in round 3's longer step, -6.5 us per step now: profiles/r06_ab_experiments.txt section 10).  The logits -> arena-zero fork is a DELAYED flag fork: its consumers read nothing the logits
// GEMM writes, and held back TCAR_FORK_DELAY us behind the GEMM's end they start when the event released them, while the main
// stream records nothing.
// (slots 8, 10 and 11 — negative term -> slab reduce, pool backward -> fused click-query backward and its join — were retired in round 6
//  with the forms that used them; the numbering of the others, which TCAR_FLAG_FORK masks, is unchanged)
// (slot 8 since: the dX GEMM's START flag — softmax gradient -> dE's stream without an event record between the softmax finish and dX)
enum { FK_TAIL2 = 0, FK_PROJ = 1, FK_TAIL3 = 2, FK_REDUCE = 3, FK_INGRAD = 4, FK_DCLICK = 5, FK_GATHER = 6, FK_QUERY = 7, FK_DXSTART = 8,
       FK_LOGITS = 9 };
// (a thirteenth slot, softmax gradient -> dE's stream with the plane stored write-through, measured +17 .. +21 us in rounds 3 and 4: removed)
// host-side fork state of ONE context (tcar_ctx_t.fork_host: caller-owned, zeroed, tcar_fork_state_bytes() bytes)
struct ForkSlot { TcarSignal sig; uint32_t live; uint32_t pad; };     // live: the launch armed last for this slot carries sig
struct ForkHost { uint32_t epoch; uint32_t pad[3]; ForkSlot slot[TCAR_SIG_SLOTS]; };
inline ForkHost* fork_host(const tcar_ctx_t* c) { return c->sig_dev ? static_cast<ForkHost*>(c->fork_host) : nullptr; }
inline const TcarTuning& tn(const tcar_ctx_t* c) { return c->tune ? *c->tune : tcar_tuning(); }
inline TcarOpt opt_of(const tcar_ctx_t* c) { TcarOpt o; o.tune = c->tune; return o; }
// the flag the producing launch is asked to carry; empty: this fork is an event fork
inline TcarSignal fork_arm(const tcar_ctx_t* c, int slot) {
  ForkHost* f = fork_host(c);
  if (!f) return TcarSignal{};
  f->slot[slot].live = 0;
  if (!((tn(c).flag_fork >> slot) & 1)) return TcarSignal{};      // the switch is a mask over the slots
  f->slot[slot].sig = TcarSignal{c->sig_dev + slot, c->sig_dev + TCAR_SIG_SLOTS + slot, ++f->epoch, (unsigned)slot};
  return f->slot[slot].sig;
}
// behind the producing launch: does it really carry the flag armed for `slot`?  (false: fork_go records an event)
inline bool fork_commit(const tcar_ctx_t* c, int slot, const TcarOpt& o) {
  ForkHost* f = fork_host(c);
  if (!f) return false;
  ForkSlot& s = f->slot[slot];
  s.live = (o.carried && o.sig.cnt && o.sig.cnt == s.sig.cnt && o.sig.epoch == s.sig.epoch) ? 1u : 0u;
  return s.live != 0;
}
inline void fork_disarm(const tcar_ctx_t* c, int slot) {
  if (ForkHost* f = fork_host(c)) f->slot[slot].live = 0;
}
inline const ForkSlot* fork_live(const tcar_ctx_t* c, int slot) {
  ForkHost* f = fork_host(c);
  return (f && f->slot[slot].live) ? &f->slot[slot] : nullptr;
}
inline int fork_go(const tcar_ctx_t* c, int slot, hipStream_t from, hipStream_t to, void* ev, int delay_us = 0) {
  if (const ForkSlot* s = fork_live(c, slot)) {
    TCAR_LAUNCH(poll_flag_kernel, dim3(1), dim3(64), 0, to, (const unsigned*)s->sig.flag, s->sig.epoch, c->sig_dev + TCAR_SIG_ERR,
                c->sig_err_host, POLL_TICKS, (const unsigned*)nullptr, 0u, delay_us * 100);
    TCAR_CHECK_LAUNCH();
    return TCAR_OK;
  }
  if (hipEventRecord((hipEvent_t)ev, from) != hipSuccess || hipStreamWaitEvent(to, (hipEvent_t)ev, 0) != hipSuccess) return TCAR_E_LAUNCH;
  return TCAR_OK;
}

// join of TWO producers into `to`: one polling kernel when both forks are flag forks, else each on its own (poll or event)
inline int fork_go2(const tcar_ctx_t* c, int slot_a, hipStream_t from_a, void* ev_a, int slot_b, hipStream_t from_b, void* ev_b, hipStream_t to) {
  const ForkSlot* a = fork_live(c, slot_a);
  const ForkSlot* b = fork_live(c, slot_b);
  if (a && b) {
    TCAR_LAUNCH(poll_flag_kernel, dim3(1), dim3(64), 0, to, (const unsigned*)a->sig.flag, a->sig.epoch, c->sig_dev + TCAR_SIG_ERR,
                c->sig_err_host, POLL_TICKS, (const unsigned*)b->sig.flag, b->sig.epoch, 0);
    TCAR_CHECK_LAUNCH();
    return TCAR_OK;
  }
  RET(fork_go(c, slot_a, from_a, to, ev_a));
  return fork_go(c, slot_b, from_b, to, ev_b);
}

__global__ void set_flag_kernel(unsigned* flag, unsigned epoch) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace

// Do a polling kernel on `side` and a kernel enqueued BEHIND it on `main` run concurrently?  They do not when a tool serialises
// kernel execution (counter collection, AMD_SERIALIZE_KERNEL) or when both streams map to one hardware queue: every flag fork
// would then sit out its time-out.  One probe with a 20-ms time-out; *concurrent = 0 means "leave sig_dev NULL" (events).
// Synchronises both streams; the words of slot 15 and the error count are left as they were found.
extern "C" int tcar_flag_fork_selftest(uint32_t* sig_dev, void* main_stream, void* side_stream, int32_t* concurrent) {
  if (!sig_dev || !side_stream || !concurrent || main_stream == side_stream) return TCAR_E_ARG;
  hipStream_t sm = (hipStream_t)main_stream, ss = (hipStream_t)side_stream;
  unsigned before[2] = {0, 0}, after = 0;
  if (hipStreamSynchronize(sm) != hipSuccess || hipStreamSynchronize(ss) != hipSuccess) return TCAR_E_LAUNCH;
  if (hipMemcpy(&before[0], sig_dev + TCAR_SIG_ERR, 4, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(&before[1], sig_dev + TCAR_SIG_SLOTS + 15, 4, hipMemcpyDeviceToHost) != hipSuccess)
    return TCAR_E_LAUNCH;
  const unsigned epoch = before[1] + 1u;
  TCAR_LAUNCH(poll_flag_kernel, dim3(1), dim3(64), 0, ss, (const unsigned*)(sig_dev + TCAR_SIG_SLOTS + 15), epoch, sig_dev + TCAR_SIG_ERR,
              (unsigned*)nullptr, 2000000LL);
  TCAR_CHECK_LAUNCH();
  TCAR_LAUNCH(set_flag_kernel, dim3(1), dim3(64), 0, sm, sig_dev + TCAR_SIG_SLOTS + 15, epoch);
  TCAR_CHECK_LAUNCH();
  if (hipStreamSynchronize(sm) != hipSuccess || hipStreamSynchronize(ss) != hipSuccess) return TCAR_E_LAUNCH;
  if (hipMemcpy(&after, sig_dev + TCAR_SIG_ERR, 4, hipMemcpyDeviceToHost) != hipSuccess) return TCAR_E_LAUNCH;
  *concurrent = (after == before[0]) ? 1 : 0;
  if (hipMemcpy(sig_dev + TCAR_SIG_ERR, &before[0], 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(sig_dev + TCAR_SIG_SLOTS + 15, &before[1], 4, hipMemcpyHostToDevice) != hipSuccess)
    return TCAR_E_LAUNCH;
  return TCAR_OK;
}

extern "C" int64_t tcar_fork_state_bytes(void) { return (int64_t)sizeof(ForkHost); }
extern "C" int64_t tcar_ctx_bytes(void) { return (int64_t)sizeof(tcar_ctx_t); }

// Diagnostic: ONE polling kernel on `stream` that waits ~10 us for an epoch of slot 15 nobody will publish, i.e. a poll that
// gives up — exactly what a step leaves behind when its streams do not overlap: sig_dev[32] += 1 and, with err_host, the
// host-visible mirror += 1 (system-scope atomic).  Engines test their fail-fast path with it.
extern "C" int tcar_flag_poll_expire(uint32_t* sig_dev, uint32_t* err_host, void* stream) {
  if (!sig_dev) return TCAR_E_ARG;
  TCAR_LAUNCH(poll_flag_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned*)(sig_dev + TCAR_SIG_SLOTS + 15), 0x7fffffffu,
              sig_dev + TCAR_SIG_ERR, err_host, 1000LL, (const unsigned*)nullptr, 0u, 0);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

namespace {
int check_ctx(const tcar_ctx_t* c, const tcar_batch_t* bt) {
  if (!c || !bt || bt->B <= 0 || bt->T <= 0 || bt->T > TCAR_POS_VOCAB) return TCAR_E_ARG;
  if (!c->E || !c->W || !c->Gx || !c->M || !c->V || !c->big || !c->Mi || !c->Vi) return TCAR_E_ARG;
  return TCAR_OK;
}

}  // namespace

namespace {
// `rest_lr` >= 0: the item rows the early pass of a split update left out are updated on the aux stream, behind the
// candidate-time refresh and in front of the join that the logits GEMM waits for
// `train_index`: the fused training step follows — the sort index of its deterministic item-row sum (segsum.hip) depends on
// the feed only and is built on the aux stream under the forward pass
int forward_impl(const tcar_ctx_t* c, const tcar_batch_t* bt, int refresh_time, void* stream, float rest_lr, bool train_index);

// does the fused step add the item-row gradients through the sorted segmented sum?
bool sorted_rows(const tcar_ctx_t* c, const tcar_batch_t* bt) {
  if (!aux_stream(c) || !c->segsum_ws || !tn(c).sort_scatter || c->d.ldh > 512) return false;
  const bool has_neg = bt->K > 0 && bt->neg && c->neg_coef && c->negpart;
  return c->segsum_bytes >= tcar_segsum_ws_bytes(&c->d, (int64_t)bt->B * (bt->T + (has_neg ? bt->K : 0)));
}
}  // namespace

namespace {
// One-hot form of the candidate-side time columns in a fused training step.  Forward: the logits GEMM contracts
// 2 ldh + 160 columns (P OH^T).  Backward (TCAR_ONEHOT_TIME >= 2, ldt = 64, the order-fixed schedule): dX = dlogits [E_ic | OH],
// dE keeps only its item block + per-candidate (q, z) pairs, and no launch of the step reads the time planes of E — they are
// not refreshed by such a step.  Both halves of a step evaluate these predicates on the same (context, batch).
bool onehot_fwd(const tcar_ctx_t* c, int B) {
  return fused_ce(c, B, nullptr) && c->oh16 && c->p16h && c->p16l && c->scoring == 3 && tn(c).onehot_time != 0;
}
bool onehot_bwd(const tcar_ctx_t* c, const tcar_batch_t* bt) {
  return onehot_fwd(c, bt->B) && tn(c).onehot_time >= 2 && c->tclip && c->dP && c->qz && c->d.ldt == 64 && c->et_perm && c->inv_off &&
         c->ct_ws && c->stream3 && c->ev3 && c->gw_rows && sorted_rows(c, bt) && tn(c).det_small != 0;
}
// anchored softmax form (score.hip: ce_anchor_fold_kernel): the one-hot step with the buffers of the form and at most eight 64-column
// anchor partials per row (the anchor columns of the one-hot K segment: embed.hip)
bool ce_anchored(const tcar_ctx_t* c, const tcar_batch_t* bt) {
  return onehot_bwd(c, bt) && tn(c).fused_ce >= 2 && c->ce_rowscale && c->aps16h && c->ce_form && 2 * c->d.ldh / 64 <= TCAR_ANCHOR_COLS;
}
}  // namespace

extern "C" int tcar_step_forward(const tcar_ctx_t* c, const tcar_batch_t* bt, int refresh_time, void* stream) {
  return forward_impl(c, bt, refresh_time, stream, -1.f, false);
}

namespace {
// model_combine.py:52-132 for the sessions of `bt`: gather + clip, the three input projections, the click query, both
// attention pools and the output transforms -> c->attout [B, ek] (+ its bf16 planes when `planes`)
// `hook(stage, o)`: called in front of the projection launch (0: it may put a completion flag into o->sig for that launch to carry),
// behind it (1: o->carried says whether it did) and behind the click-query launch (2) — forward_impl forks the rest pass of a
// pending split update there
// `so`: where the one-hot time scores of the logits GEMM go (forward_impl of a training step); when the output transforms run in
// their split form the finishing launch computes them too and sets so->done — the caller then skips tcar_time_scores_clip
struct ScoreOut {
  void* p_hi; void* p_lo; float* tclip; bool done;
  // anchored softmax form: the finishing launch also leaves minus the anchor partials in the anchor columns of P (label rows of E: see
  // forward_impl); anchor_done says so
  const int32_t* label = nullptr; const float* E = nullptr; int64_t ldE = 0; bool anchor_done = false;
};
template <class Hook>
int session_forward(const tcar_ctx_t* c, const tcar_batch_t* bt, const Geo& g, void* stream, bool planes, int ei, Hook hook,
                    ScoreOut* so = nullptr) {
  const int B = bt->B, BT = bt->B * bt->T;
  tcar_tables_t tab;
  tables_of(c, tab);
  // The click-query MLP (q1, q: modules.py:138-139) needs only the gathered click-time rows: with flag forks it runs as ONE
  // launch (query.hip) on the third stream BESIDE the projections instead of as two small GEMMs between them and the pools
  // (22 + 30 us on the main chain).  The gather publishes a flag for it, it publishes one for the pools; both producers store
  // what the other stream reads write-through, so the polling kernels write back no L2.
  const int fmask = tn(c).flag_fork;
  hipStream_t sq = (g.ldh == 256 && g.ldt == 64 && c->stream3 && c->ev3 && fork_host(c) && ((fmask >> FK_GATHER) & 1) &&
                    ((fmask >> FK_QUERY) & 1))
                       ? (hipStream_t)c->stream3 : nullptr;
  TcarOpt og = opt_of(c);
  if (sq) og.sig = fork_arm(c, FK_GATHER);
  if (ei >= 0) (void)hipEventRecord((hipEvent_t)c->ev_start[4 * c->ev_n + ei], (hipStream_t)stream);     // kind 4: the step's own gather
  RET(tcar_gather_clip_fwd_o(&c->d, &tab, bt, c->x_icp, c->x_pt, c->x_act, c->click_t, stream, &og));
  if (ei >= 0) (void)hipEventRecord((hipEvent_t)c->ev_stop[4 * c->ev_n + ei], (hipStream_t)stream);
  const bool qside = sq && fork_commit(c, FK_GATHER, og);       // (the throughput form of the gather carries no flag)
  if (qside) {
    RET(fork_go(c, FK_GATHER, (hipStream_t)stream, sq, c->ev3));
    TcarOpt oq = opt_of(c);
    oq.sig = fork_arm(c, FK_QUERY);
    RET(tcar_query_mlp_o(&c->d, B, c->click_t, W(c, TCAR_V_Q1_W), W(c, TCAR_V_Q1_B), W(c, TCAR_V_Q2_W), W(c, TCAR_V_Q2_B), c->q1,
                         c->q, (void*)sq, &oq));
    if (!fork_commit(c, FK_QUERY, oq)) return TCAR_E_LAUNCH;
  }
  // without the side stream (the replica engine, contexts without the flag words): the same ONE launch on this stream in
  // place of the two small GEMMs of the split-bf16 modes
  const bool qfused = qside || (g.ldh == 256 && g.ldt == 64 && c->scoring != 0);
  if (qfused && !qside)
    RET(tcar_query_mlp(&c->d, B, c->click_t, W(c, TCAR_V_Q1_W), W(c, TCAR_V_Q1_B), W(c, TCAR_V_Q2_W), W(c, TCAR_V_Q2_B), c->q1,
                       c->q, stream));
  const float* x_c = c->x_icp + g.ldh;
  // pre1, pre2, q1 (modules.py:126-131, 94-96, 138).  With the slab workspace (split-bf16 modes) every (operand pair, 128-deep K
  // chunk) of the two projections is its OWN problem of the grouped launch writing its own slab: 7 + 5 chunks x 32-72 tiles
  // instead of 2 x 32-72 workgroups walking 13 / 9 serial stages — a workgroup pays one global-memory round trip — and the pool
  // kernel folds the slabs in slab order while it reads them (one writer per element: order-fixed)
  const int n1 = units(g.ic) + units(g.ldh) + units(g.ldt), n2 = units(g.pt) + units(g.ldh);
  const int64_t stride = (int64_t)BT * g.ldh;
  const bool split = c->scoring && c->proj_slabs && c->proj_slab_floats >= (n1 + n2) * stride;
  // (the slab form of the PROJECTIONS only up to TCAR_PROJ_SPLIT_ROWS rows: from there the launch has workgroups enough without the
  //  split, and 12 slabs of [BT, ldh] written and folded again cost more than the serial stages they save; the output transforms
  //  below are B-row problems and keep their split at every T)
  const bool psplit = split && BT <= tn(c).proj_split_rows;
  TcarOpt op = opt_of(c);          // the projection launch: the hook may hand it a completion flag
  if (psplit) {
    float* s1 = c->proj_slabs;
    float* s2 = c->proj_slabs + n1 * stride;
    tcar_gemm_desc_t p[6];
    p[0] = prob1(BT, g.ldh, c->x_icp, g.ic, W(c, TCAR_V_M_WIN), g.ldh, g.ic, s1, g.ldh, nullptr, 0, 0, units(g.ic));
    p[1] = prob1(BT, g.ldh, x_c, g.ic, W(c, TCAR_V_M_WC), g.ldh, g.ldh, s1 + units(g.ic) * stride, g.ldh, nullptr, 0, 0, units(g.ldh));
    p[2] = prob1(BT, g.ldh, c->x_act, g.ldt, W(c, TCAR_V_M_WINT), g.ldh, g.ldt, s1 + (units(g.ic) + units(g.ldh)) * stride, g.ldh,
                 nullptr, 0, 0, units(g.ldt));
    p[3] = prob1(BT, g.ldh, c->x_pt, g.pt, W(c, TCAR_V_S_WIN), g.ldh, g.pt, s2, g.ldh, nullptr, 0, 0, units(g.pt));
    p[4] = prob1(BT, g.ldh, x_c, g.ic, W(c, TCAR_V_S_WC), g.ldh, g.ldh, s2 + units(g.pt) * stride, g.ldh, nullptr, 0, 0, units(g.ldh));
    p[5] = prob1(B, g.ldh, c->click_t, g.ct, W(c, TCAR_V_Q1_W), g.ldh, g.ct, c->q1, g.ldh, W(c, TCAR_V_Q1_B), 1);
    // optional HIP events around exactly this launch (kind 3 of ev_start / ev_stop: the largest of the session-side small GEMMs)
    if (ei >= 0) (void)hipEventRecord((hipEvent_t)c->ev_start[3 * c->ev_n + ei], (hipStream_t)stream);
    RET(hook(0, &op));
    RET(small_gemm(c, 0, qfused ? 5 : 6, p, stream, &op));
    if (ei >= 0) (void)hipEventRecord((hipEvent_t)c->ev_stop[3 * c->ev_n + ei], (hipStream_t)stream);
  } else {
    tcar_gemm_desc_t p[3];
    p[0] = prob(BT, g.ldh, c->pre1, g.ldh);
    seg(p[0], c->x_icp, g.ic, W(c, TCAR_V_M_WIN), g.ldh, g.ic);
    seg(p[0], x_c, g.ic, W(c, TCAR_V_M_WC), g.ldh, g.ldh);
    seg(p[0], c->x_act, g.ldt, W(c, TCAR_V_M_WINT), g.ldh, g.ldt);
    p[1] = prob(BT, g.ldh, c->pre2, g.ldh);
    seg(p[1], c->x_pt, g.pt, W(c, TCAR_V_S_WIN), g.ldh, g.pt);
    seg(p[1], x_c, g.ic, W(c, TCAR_V_S_WC), g.ldh, g.ldh);
    p[2] = prob1(B, g.ldh, c->click_t, g.ct, W(c, TCAR_V_Q1_W), g.ldh, g.ct, c->q1, g.ldh, W(c, TCAR_V_Q1_B), 1);
    if (ei >= 0) (void)hipEventRecord((hipEvent_t)c->ev_start[3 * c->ev_n + ei], (hipStream_t)stream);      // (kind 3 in the un-split form too)
    RET(hook(0, &op));
    RET(small_gemm(c, 0, qfused ? 2 : 3, p, stream, &op));
    if (ei >= 0) (void)hipEventRecord((hipEvent_t)c->ev_stop[3 * c->ev_n + ei], (hipStream_t)stream);
  }
  RET(hook(1, &op));
  // the pools wait for q: a polling kernel behind the query MLP's flag, or an event.  (The pool kernel can also wait for that flag
  // ITSELF, behind its slab fold — tcar_attn_pool_fwd_slabs_w's wait argument; measured no faster in rounds 4 and 5, not used here.)
  const TcarWait wq{};
  if (qside) {
    RET(fork_go(c, FK_QUERY, sq, (hipStream_t)stream, c->ev3));
  } else if (!qside && !qfused) {       // q = tanh(q1 Wq2 + b) (modules.py:139)
    tcar_gemm_desc_t p = prob1(B, g.ic, c->q1, g.ldh, W(c, TCAR_V_Q2_W), g.ic, g.ldh, c->q, g.ic, W(c, TCAR_V_Q2_B), 2);
    RET(small_gemm(c, 0, 1, &p, stream));
  }
  RET(hook(2, &op));
  if (psplit)
    RET(tcar_attn_pool_fwd_slabs_w(&c->d, B, bt->T, c->x_icp, c->x_pt, c->proj_slabs, n1, c->proj_slabs + n1 * stride, n2, stride,
                                   c->pre1, c->pre2, c->q, W(c, TCAR_V_M_WRES), W(c, TCAR_V_S_WRES), c->pooled, c->alpha, stream, wq));
  else
    RET(tcar_attn_pool_fwd(&c->d, B, bt->T, c->x_icp, c->x_pt, c->pre1, c->pre2, c->q, W(c, TCAR_V_M_WRES),
                           W(c, TCAR_V_S_WRES), c->pooled, c->alpha, stream));
  // attout (model_combine.py:119,127,132).  Split form (slab workspace, ldt = 64): every 128-deep K chunk of the two output
  // transforms is its own set of workgroups (376 instead of 104 walking 8 / 5 serial stages), and ONE launch folds the slabs, adds
  // the bias, applies tanh, writes attout + its planes and goes on to the one-hot time scores (embed.hip: attout_finish_kernel)
  const int na_ic = units(g.ic), na_pt = units(g.pt);
  const int64_t astride = (int64_t)B * g.ek;
  if (split && g.ldt == 64 && (g.ldh & 63) == 0 &&
      c->proj_slab_floats >= (na_ic > na_pt ? na_ic : na_pt) * astride) {
    tcar_gemm_desc_t p[2];
    p[0] = prob1(B, g.ic, c->pooled, g.ek, W(c, TCAR_V_O_W), g.ic, g.ic, c->proj_slabs, g.ek, nullptr, 0, 0, na_ic);
    p[1] = prob1(B, g.pt, c->pooled + g.ic, g.ek, W(c, TCAR_V_OT_W), g.pt, g.pt, c->proj_slabs + g.ic, g.ek, nullptr, 0, 0, na_pt);
    RET(small_gemm(c, 0, 2, p, stream));
    const float* tt[5];
    for (int k = 0; k < 5; ++k) tt[k] = W(c, TCAR_V_MONTH + k);
    RET(tcar_attout_finish_scores_a(&c->d, tt, B, c->proj_slabs, na_ic, na_pt, astride, W(c, TCAR_V_O_B), W(c, TCAR_V_OT_B), c->attout,
                                    g.ek, planes ? c->a16h : nullptr, planes ? c->a16l : nullptr, g.ek, planes ? c->ap16h : nullptr,
                                    planes ? c->ap16l : nullptr, g.ldh + g.pt, so ? so->p_hi : nullptr, so ? so->p_lo : nullptr, 160,
                                    so ? so->tclip : nullptr, so ? so->label : nullptr, so ? so->E : nullptr, so ? so->ldE : 0, stream));
    if (so) { so->done = true; so->anchor_done = so->label != nullptr; }
    return TCAR_OK;
  }
  {
    tcar_gemm_desc_t p[2];
    p[0] = prob1(B, g.ic, c->pooled, g.ek, W(c, TCAR_V_O_W), g.ic, g.ic, c->attout, g.ek, W(c, TCAR_V_O_B), 2);
    p[1] = prob1(B, g.pt, c->pooled + g.ic, g.ek, W(c, TCAR_V_OT_W), g.pt, g.pt, c->attout + g.ic, g.ek,
                 W(c, TCAR_V_OT_B), 2);
    if (planes) {
      // split-bf16 scoring: the epilogue also writes attout's hi / lo planes (operand of the logits GEMM) and the packed
      // item | time planes (operand of dE) — tcar_split_bf16 without its own launch
      for (int i = 0; i < 2; ++i) {
        p[i].plane_hi = c->a16h; p[i].plane_lo = c->a16l; p[i].plane_inner = g.ek; p[i].plane_col0 = i ? g.ic : 0;
        p[i].pack_hi = c->ap16h; p[i].pack_lo = c->ap16l; p[i].pack_inner = g.ldh + g.pt; p[i].pack_c0 = g.ldh; p[i].pack_c1 = g.ic;
      }
    }
    RET(small_gemm(c, 0, 2, p, stream));
  }
  return TCAR_OK;
}
}  // namespace

namespace {
// the gradient arena (+ norm pieces) and the dense norm slots in ONE launch (two hipMemsetAsync calls cost more host time
// and two blit kernels)
__global__ __launch_bounds__(256) void zero2_kernel(float4* __restrict__ a, long na4, float* __restrict__ b, int nb) {
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < na4; i += (long)gridDim.x * 256) a[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < nb) b[threadIdx.x] = 0.f;
}
int zero_arena(const tcar_ctx_t* c, hipStream_t s) {
  const long n = (long)c->arena_n + TCAR_NSLOT;
  if ((n & 3) || !tcar_aligned16(c->Gx) || TCAR_NSLOT > 256) {
    if (hipMemsetAsync(c->Gx, 0, (size_t)n * sizeof(float), s) != hipSuccess) return TCAR_E_LAUNCH;
    if (hipMemsetAsync(c->sqn_dense, 0, TCAR_NSLOT * sizeof(float), s) != hipSuccess) return TCAR_E_LAUNCH;
    return TCAR_OK;
  }
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  TCAR_LAUNCH(zero2_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (float4*)c->Gx, n / 4, c->sqn_dense, (int)TCAR_NSLOT);
  TCAR_CHECK_LAUNCH();
  return TCAR_OK;
}

// zero the gradient arena and the norm slots, and the forward part of the sampled negative term (it needs only attout and E):
// on the aux stream `sz`, ordered behind everything already on the main stream (the previous update read Gx)
// (Round 3 A/B: saving this fork's event — arena zeroed on the already-ordered aux stream, the negative term's forward behind the
// dE fork — and joining the three streams through ONE wait at the end of the step measured SLOWER, 0.635 vs 0.613 ms per step:
// dE then starts behind the negative term and the final join becomes two hops.)
int backward_prologue(const tcar_ctx_t* c, const tcar_batch_t* bt, hipStream_t st, hipStream_t sz) {
  // (armed by forward_impl in front of the logits GEMM of a softmax-epilogue step; an event otherwise.  The arm must be THIS
  // context's latest fork action — nothing else forks between that launch and here —, else it is a leftover: event)
  if (const ForkSlot* f = fork_live(c, FK_LOGITS))
    if (f->sig.epoch != fork_host(c)->epoch) fork_disarm(c, FK_LOGITS);
  if (sz != st) RET(fork_go(c, FK_LOGITS, st, sz, c->ev[0], tcar_fixed::fork_delay));
  RET(zero_arena(c, sz));
  if (bt->K > 0 && bt->neg && c->neg_coef && c->negpart) {
    TcarOpt on = opt_of(c);
    RET(tcar_neg_fwd_o(&c->d, bt->B, bt->K, c->E, bt->neg, c->attout, c->neg_weight, c->neg_fb, c->neg_coef, c->negpart, (void*)sz, &on));
  }
  return TCAR_OK;
}

int forward_impl(const tcar_ctx_t* c, const tcar_batch_t* bt, int refresh_time, void* stream, float rest_lr, bool train_index) {
  RET(check_ctx(c, bt));
  const Geo g(c->d);
  const int B = bt->B, BT = bt->B * bt->T;
  // The candidate-side time block of E depends only on the time tables: it is rebuilt on the auxiliary stream while
  // the session side (gather, projections, pools) runs on the main one; the logits GEMM joins them.
  hipStream_t s1 = (hipStream_t)stream, s2 = aux_stream(c);
  bool joined = true;
  int rest_stage = -1;
  // a fused training step in the one-hot form reads the candidate-side time planes of E nowhere: their refresh is left to the
  // next entry point that does (evaluation, the op-level paths: the engine keeps refresh_time set until one of them ran)
  const bool oh_bwd = train_index && onehot_bwd(c, bt);
  // anchored softmax form: the finishing launch of the session forward reads the LABEL rows of E (item | content columns) for the
  // anchor — of a pending split update they join the early part, so that the rest pass (aux stream, beside the forward) never
  // writes a row this step's forward reads: no race, the same bits in every run
  const bool anchored = oh_bwd && ce_anchored(c, bt);
  // REST pass of a pending split update, on the aux stream (behind whatever the main stream has enqueued so far when it is
  // forked late); ev[1] = "aux stream ready for the logits GEMM"
  auto launch_rest = [&]() -> int {
    const float* pieces = c->Gx + c->arena_n;
    RET(tcar_clip_adam_rest_keep_o(c->E, g.ek, c->big, c->Mi, c->Vi, g.N, g.ldh, c->slot_item, c->sqn_dense, pieces, c->use_dense,
                                   c->clip, rest_lr, c->b1, c->b2, c->eps, c->scoring ? c->e16h : nullptr,
                                   c->scoring ? c->e16l : nullptr, g.ek, c->adam_bitmap, (void*)s2, tcar_fixed::rest_grid));
    if (hipEventRecord((hipEvent_t)c->ev[1], s2) != hipSuccess) return TCAR_E_LAUNCH;
    // the marks are cleared BEHIND the event the logits GEMM waits for (the clear is a launch of its own)
    // (the map is allocated in whole 64-byte units, tcar_hip.h: ONE fill kernel — a size that is no multiple of 16 bytes costs a second one)
    if (hipMemsetAsync(c->adam_bitmap, 0, (size_t)(((g.N + 31) / 32 + 15) & ~15) * sizeof(uint32_t), s2) != hipSuccess) return TCAR_E_LAUNCH;
    return TCAR_OK;
  };
  if (refresh_time) {
    const float* tt[5];
    for (int k = 0; k < 5; ++k) tt[k] = W(c, TCAR_V_MONTH + k);
    if (s2 && rest_lr >= 0.f) {
      // The pending split update (tcar_train_step_deferred): EARLY pass on the main stream (arena + the item rows this batch
      // gathers), then — on the aux stream, beside the session forward — the time refresh and the REST pass over every other
      // item row; the logits GEMM joins.  (Round 3 tried to start the rest pass at once, from row marks made by a small
      // kernel, with the time refresh on the third stream: the 376-MB pass then runs beside the gather and the projections
      // as well and slows them by more than it gains — 0.626 vs 0.618 ms per step)
      const float* pieces = c->Gx + c->arena_n;
      RET(tcar_clip_adam_early_2(c->W, c->Gx, c->M, c->V, &c->segs_all, c->E, g.ek, c->big, c->Mi, c->Vi, g.N, g.ldh,
                                 c->slot_item, c->sqn_dense, pieces, c->use_dense, c->clip, rest_lr, c->b1, c->b2, c->eps,
                                 c->scoring ? c->e16h : nullptr, c->scoring ? c->e16l : nullptr, g.ek, bt->seq, (int64_t)BT,
                                 anchored ? bt->label : nullptr, anchored ? (int64_t)B : 0, c->adam_bitmap, stream));
      // (one-hot form: nothing is launched on the aux stream here — the rest pass is forked behind the projection launch, which is
      //  behind the early pass on this stream: no event on the main chain at the top of the step)
      if (!oh_bwd && (hipEventRecord((hipEvent_t)c->ev[0], s1) != hipSuccess || hipStreamWaitEvent(s2, (hipEvent_t)c->ev[0], 0) != hipSuccess))
        return TCAR_E_LAUNCH;
      if (!oh_bwd)
        RET(tcar_cand_time_fwd_bf16(&c->d, tt, c->mwdhm, c->scoring ? nullptr : c->E, c->scoring ? c->e16h : nullptr, c->scoring ? c->e16l : nullptr, (void*)s2));
      // the rest pass is forked BEHIND the projection launch: gather and projections run without the 376-MB stream beside
      // them (21 instead of 40 us for the projections), the pass still ends before the output transforms do.  Measured over
      // three interleaved rounds: 0.6158 ms per step against 0.6213 forked at once and 0.628 forked behind the query MLP.
      rest_stage = 1;
      joined = false;
      // (Round 4 re-measured the placement on the shorter head — behind the gather through a second poll of its flag: +11 us; at once
      //  behind an event: equal — and round 5 removed those two forms: profiles/r04_ab_experiments.txt.)
    } else {
      if (s2 && (hipEventRecord((hipEvent_t)c->ev[0], s1) != hipSuccess || hipStreamWaitEvent(s2, (hipEvent_t)c->ev[0], 0) != hipSuccess))
        return TCAR_E_LAUNCH;
      if (!oh_bwd)
        RET(tcar_cand_time_fwd_bf16(&c->d, tt, c->mwdhm, c->scoring ? nullptr : c->E, c->scoring ? c->e16h : nullptr, c->scoring ? c->e16l : nullptr,
                                    s2 ? (void*)s2 : stream));
      if (s2) {
        if (hipEventRecord((hipEvent_t)c->ev[1], s2) != hipSuccess) return TCAR_E_LAUNCH;
        joined = false;
      }
    }
  }
  // optional device timing (bench.py roofline): one slot per step, chosen here; the backward pass of this step uses the same slot
  int ei = -1;
  if (c->ev_n > 0 && c->ev_start && c->ev_stop && c->ev_cursor) {
    ei = c->ev_cursor[0]++ % c->ev_n;
    c->ev_cursor[1] = ei;
  }
  CeWs w;
  const bool ce_epi = c->scoring && train_index && fused_ce(c, B, &w);
  const bool onehot = ce_epi && onehot_fwd(c, B);
  ScoreOut so{c->p16h, c->p16l, oh_bwd ? c->tclip : nullptr, false};
  if (anchored) { so.label = bt->label; so.E = c->E; so.ldE = g.ek; }
  RET(session_forward(c, bt, g, stream, c->scoring != 0, ei, [&](int stage, TcarOpt* o) -> int {
    if (rest_stage == 1 && stage == 0) o->sig = fork_arm(c, FK_PROJ);        // the projection launch carries the flag
    if (stage != rest_stage) return TCAR_OK;
    // late fork: the HBM-bound rest pass starts only now, so the launches before this point ran without it
    (void)fork_commit(c, FK_PROJ, *o);
    RET(fork_go(c, FK_PROJ, s1, s2, c->ev[0]));      // (the rest pass reads nothing the projections write: timing only)
    return launch_rest();
  }, onehot ? &so : nullptr));
  // sort index of the item rows (feed only): on the aux stream BEHIND the rest pass — the logits GEMM does not wait for it (ev[1]
  // was recorded in front of it), the backward does
  if (train_index && sorted_rows(c, bt)) {
    // (with a time refresh the aux stream was already forked from this step's main stream: it is behind the previous backward)
    if (!refresh_time &&
        (hipEventRecord((hipEvent_t)c->ev[0], s1) != hipSuccess || hipStreamWaitEvent(s2, (hipEvent_t)c->ev[0], 0) != hipSuccess))
      return TCAR_E_LAUNCH;
    RET(tcar_segsum_index(&c->d, bt, c->segsum_ws, c->segsum_bytes, (void*)s2));
  }
  // candidate-side time scores through the one-hot contraction (embed.hip: tcar_time_scores / tcar_time_onehot): the logits
  // GEMM of a training step then runs over 2 ldh + 160 columns instead of 2 ldh + 5 ldt, the one-hot block at two MFMAs and one
  // B plane per product.  The scores need attout and the time tables (early part of the update, this stream) only: launched
  // AHEAD of the join with the aux stream.
  if (onehot && !so.done) {
    const float* tt[5];
    for (int k = 0; k < 5; ++k) tt[k] = W(c, TCAR_V_MONTH + k);
    RET(tcar_time_scores_clip(&c->d, tt, B, c->attout, g.ek, c->p16h, c->p16l, 160, oh_bwd ? c->tclip : nullptr, stream));
  }
  if (!joined && hipStreamWaitEvent(s1, (hipEvent_t)c->ev[1], 0) != hipSuccess) return TCAR_E_LAUNCH;
  // logits = attout E^T (model_combine.py:138).  Optional HIP events bracket exactly the GEMM launch (bench.py roofline).
  auto start_timer = [&]() {
    if (ei >= 0) (void)hipEventRecord((hipEvent_t)c->ev_start[ei], (hipStream_t)stream);
  };
  int rc;
  fork_disarm(c, FK_LOGITS);           // (EVERY forward pass: the slot is the one fork whose arm and go sit in different calls)
  TcarOpt ol = opt_of(c);
  if (c->scoring) {
    // split-bf16 path: the planes of attout were written by the output-transform GEMM's epilogue
    start_timer();
    if (ce_epi && s2) ol.sig = fork_arm(c, FK_LOGITS);       // backward_prologue releases the aux stream behind this launch
    // the form the backward half of this step finds: anchored only when the finishing launch did leave the anchor partials
    const bool anch = anchored && onehot && so.anchor_done;
    if (c->ce_form) *c->ce_form = anch ? 1 : 0;
    ol.anchored = anch;
    if (ce_epi) {
      // training step, hi-only backward: the GEMM's softmax epilogue writes exp(x - group max) as the bf16 plane that becomes
      // dlogits, plus per-group (max, sum) — no [B, N] fp32 logits; backward_impl finishes with tcar_ce_finish
      int32_t gw = 0, ng = 0;
      if (onehot) {
        rc = tcar_gemm_bf16_ce_o(B, g.N, g.ic + 160, c->a16h, c->a16l, g.ek, B, c->e16h, c->e16l, g.ek, g.Npad, g.ic, c->p16h,
                                 c->p16l, c->oh16, 160, c->dl16h, g.Npad, (B + 127) & ~127, w.stats, w.stats_floats, bt->label,
                                 w.lab, c->scoring, &gw, &ng, stream, &ol);
      } else {
        rc = tcar_gemm_bf16_ce_o(B, g.N, g.ek, c->a16h, c->a16l, g.ek, B, c->e16h, c->e16l, g.ek, g.Npad, g.ek, nullptr, nullptr,
                                 nullptr, 0, c->dl16h, g.Npad, (B + 127) & ~127, w.stats, w.stats_floats, bt->label, w.lab,
                                 c->scoring, &gw, &ng, stream, &ol);
      }
      if (!rc) (void)fork_commit(c, FK_LOGITS, ol);
      if (!rc && c->ce_geo) { c->ce_geo[0] = gw; c->ce_geo[1] = ng; }
      else if (!rc) rc = TCAR_E_ARG;
    } else {
      rc = tcar_gemm_bf16_perm_o(1, B, g.N, g.ek, c->a16h, c->a16l, g.ek, B, c->e16h, c->e16l, g.ek, g.Npad, c->logits, g.Npad,
                                 nullptr, 0, 0, nullptr, 0, c->scoring, 1, stream, &ol);
    }
  } else {
    start_timer();
    rc = tcar_gemm_f32(1, B, g.N, g.ek, c->attout, g.ek, c->E, g.ek, c->logits, g.Npad, nullptr, 0, 0, 1, stream);
  }
  if (ei >= 0) (void)hipEventRecord((hipEvent_t)c->ev_stop[ei], (hipStream_t)stream);
  return rc;
}
}  // namespace

namespace {
// The nine weight gradients x^T dy (K = batch rows, model_combine.py:156): ONE grouped launch.  K is not split up to wgrad_ks
// (1,536) rows; longer batches split it — into slabs folded in split order when the context has the workspace (order-fixed:
// tcar_fold_slabs), else with float atomics into the zeroed arena.
// tile code of the dE (q, z) launcher: a forced one (TCAR_BF16_TILE), else by catalog size.  Up to ~1 M rows the dlogits plane sits in
// the Infinity Cache and the double-buffered 192-row tile is fastest (50.9 us alone at 46 k rows; the three-stage ring: 57.7).  Beyond,
// the plane streams from HBM and the deeper ring pays: 10 M rows alone 11.1 ms (192 x 192 double buffer) / 9.65 (192-row ring) /
// 9.27 (128-row ring) — profiles/r06_ab_experiments.txt
int de_tile(const tcar_ctx_t* c, long n_rows) {
  const int f = tn(c).bf16_tile;
  if (f == 256 || f == 128 || f == 64 || f == 1922 || f == 1923 || f == 1283 || f == 2562) return f;
  return n_rows >= (1L << 20) ? 1283 : 0;
}

int weight_grads(const tcar_ctx_t* c, const Geo& g, int B, int BT, void* stream, TcarOpt* o = nullptr) {
  const int ksdiv = tn(c).wgrad_ks > 0 ? tn(c).wgrad_ks : 1536;
  auto ks = [ksdiv](int K) { int s = (K + ksdiv - 1) / ksdiv; return s < 1 ? 1 : (s > 16 ? 16 : s); };
  const int kb = ks(B), kr = ks(BT);
  const float* x_c = c->x_icp + g.ldh;
  tcar_gemm_desc_t p[9];
  p[0] = prob1(g.ic, g.ic, c->pooled, g.ek, c->dattout, g.ek, B, G(c, TCAR_V_O_W), g.ic, nullptr, 0, 0, kb, 1);
  p[1] = prob1(g.pt, g.pt, c->pooled + g.ic, g.ek, c->dattout + g.ic, g.ek, B, G(c, TCAR_V_OT_W), g.pt, nullptr, 0, 0, kb, 1);
  p[2] = prob1(g.ldh, g.ic, c->q1, g.ldh, c->dq, g.ic, B, G(c, TCAR_V_Q2_W), g.ic, nullptr, 0, 0, kb, 1);
  p[3] = prob1(g.ct, g.ldh, c->click_t, g.ct, c->dq1, g.ldh, B, G(c, TCAR_V_Q1_W), g.ldh, nullptr, 0, 0, kb, 1);
  p[4] = prob1(g.ic, g.ldh, c->x_icp, g.ic, c->dpre1, g.ldh, BT, G(c, TCAR_V_M_WIN), g.ldh, nullptr, 0, 0, kr, 1);
  p[5] = prob1(g.ldh, g.ldh, x_c, g.ic, c->dpre1, g.ldh, BT, G(c, TCAR_V_M_WC), g.ldh, nullptr, 0, 0, kr, 1);
  p[6] = prob1(g.ldt, g.ldh, c->x_act, g.ldt, c->dpre1, g.ldh, BT, G(c, TCAR_V_M_WINT), g.ldh, nullptr, 0, 0, kr, 1);
  p[7] = prob1(g.pt, g.ldh, c->x_pt, g.pt, c->dpre2, g.ldh, BT, G(c, TCAR_V_S_WIN), g.ldh, nullptr, 0, 0, kr, 1);
  p[8] = prob1(g.ldh, g.ldh, x_c, g.ic, c->dpre2, g.ldh, BT, G(c, TCAR_V_S_WC), g.ldh, nullptr, 0, 0, kr, 1);
  // slab form: problem i writes its splits to slabs_i [split][M][N]
  tcar_fold_t f[9];
  int nf = 0;
  int64_t need = 0;
  for (int i = 0; i < 9; ++i)
    if (p[i].splitk > 1) need += (int64_t)p[i].splitk * p[i].M * p[i].N;
  if (need > 0 && c->wgrad_slabs && c->wgrad_slab_floats >= need) {
    float* at = c->wgrad_slabs;
    for (int i = 0; i < 9; ++i) {
      if (p[i].splitk <= 1) continue;
      const int K = (i < 4) ? B : BT;
      f[nf].dst = p[i].C; f[nf].slabs = at; f[nf].n = (int64_t)p[i].M * p[i].N;
      f[nf].ks = tcar_gemm_splitk_effective(K, p[i].splitk); f[nf].stride = (int64_t)p[i].M * p[i].N;
      ++nf;
      p[i].C = at; p[i].atomic = 0;
      at += (int64_t)p[i].splitk * p[i].M * p[i].N;
    }
  }
  RET(small_gemm(c, 2, 9, p, stream, o));
  if (nf) RET(tcar_fold_slabs(nf, f, stream));
  return TCAR_OK;
}

// bias gradients of the four linear_2d layers and the two residual-weight gradients as column sums in a fixed order
void det_colsum_list(const tcar_ctx_t* c, const Geo& g, int B, tcar_colsum_t (&cs)[6]) {
  const tcar_colsum_t v[6] = {{c->dattout, g.ek, B, g.ic, G(c, TCAR_V_O_B)},
                              {c->dattout + g.ic, g.ek, B, g.pt, G(c, TCAR_V_OT_B)},
                              {c->dq, g.ic, B, g.ic, G(c, TCAR_V_Q2_B)},
                              {c->dq1, g.ldh, B, g.ldh, G(c, TCAR_V_Q1_B)},
                              {c->gw_rows, g.ic, B, g.ldh, G(c, TCAR_V_M_WRES)},
                              {c->gw_rows + g.ldh, g.ic, B, g.ldh, G(c, TCAR_V_S_WRES)}};
  for (int i = 0; i < 6; ++i) cs[i] = v[i];
}
int det_colsums(const tcar_ctx_t* c, const Geo& g, int B, void* stream) {
  tcar_colsum_t cs[6];
  det_colsum_list(c, g, B, cs);
  return tcar_colsum_det(6, cs, stream);
}
int finish_dense_side(const tcar_ctx_t* c, const Geo& g, void* stream);
int item_norm(const tcar_ctx_t* c, const Geo& g, void* stream);
int cand_time_backward(const tcar_ctx_t* c, const Geo& g, void* stream);

// Backward pass.  Three chains follow the softmax gradient (fused single-rank step; `main` is the caller's stream, which the
// engine makes a high-priority one so that its workgroups are dispatched first):
//   main:   softmax -> dX = dlogits E -> slab reduce + tanh' -> attention / projection / query backward -> input gradients
//           -> negative rows (sorted sum) + loss -> dense-norm partials -> item-row gradients -> session rows (sorted sum)
//           + norm folds -> [join] -> (update)
//   aux:    zero arena, negative-term forward (beside the softmax) -> dE = dlogits^T attout (time block in inverted-index
//           order) -> candidate-side time backward -> position / time / dwell tables of the session side (order-fixed)
//   third:  weight-gradient GEMM (+ slab fold) -> bias / residual-weight column sums -> dense-weight norms
// No sum in these chains depends on arrival order (determinism); the atomic forms stay behind switches.
// Every cross-stream join costs ~10 us of launch latency behind an event, so there are as few as the data flow allows.
// Rank-local backward of the data-parallel step (fuse_finish = false): dE and the negative rows run FIRST on the main
// stream (their all-reduce then overlaps everything else, dp.py); the finish is tcar_step_finish after the exchange.
int backward_impl(const tcar_ctx_t* c, const tcar_batch_t* bt, void* stream, bool fuse_finish, bool ce_epilogue = false) {
  RET(check_ctx(c, bt));
  const Geo g(c->d);
  const int B = bt->B, T = bt->T, BT = B * T, K = bt->K;
  hipStream_t st = (hipStream_t)stream, s2 = aux_stream(c);
  // chain B runs on the aux stream in the fused single-rank step; in the rank-local backward of the data-parallel step
  // it runs FIRST on the main stream, so that dE is complete early and its all-reduce overlaps chain A (dp.py)
  void* sB = (s2 && fuse_finish) ? (void*)s2 : stream;
  void* sW = s2 ? (void*)s2 : stream;          // the weight-gradient GEMM (third stream below, when there is one)
  const bool has_neg = K > 0 && bt->neg && c->neg_coef && c->negpart;
  // zero the gradient arena and the norm slots; with an aux stream this happens beside the softmax, not before it
  // (the aux stream is first ordered behind everything already on the main stream: the previous update read Gx)
  hipStream_t sz = s2 ? s2 : st;
  RET(backward_prologue(c, bt, st, sz));
  // sorted segmented sum of the item-row gradients (deterministic); its index was built under the forward pass, on the aux
  // stream — ev[1] (recorded behind the prologue) orders the main stream behind it
  const bool sorted = fuse_finish && sorted_rows(c, bt);
  // one-hot form of the two gradient GEMMs (the forward half of this step made the same decision: onehot_bwd)
  const bool ohb = ce_epilogue && fuse_finish && onehot_bwd(c, bt);
  // ev[1] = "aux prologue done" (arena zeroed, negative term's forward): in the one-hot schedule its only reader on the main chain is
  // the slab reduce (negpart).  Everything else that needs the zeroed arena sits behind dE on its own stream (stream order / ev[4]).
  // (The slab reduce can also wait for the negative term's flag in-kernel — tcar_reduce_dact_onehot_o's wait —: measured no faster.)
  if (s2 && hipEventRecord((hipEvent_t)c->ev[1], s2) != hipSuccess) return TCAR_E_LAUNCH;
  float* Gi = c->big;
  float* d_et = c->big + (size_t)g.N * g.ldh;
  const int S = tcar_gemm_splitk_effective(g.Npad, c->splitk);
  // backward precision: scoring_bwd (1 = hi planes only) or the forward precision
  const int nsb = c->scoring_bwd ? c->scoring_bwd : c->scoring;
  // hi-only backward (bf16x3-mixed, bf16): the lo plane of dlogits is never read — and not written
  CeWs cw;
  // anchored softmax form (decided by the forward half of this step): no pass over the plane — a fold launch leaves the row scales,
  // the label's -1 inside the plane and the scaled attout plane of dE; dX is scaled in its slab reduce
  const bool anch = ohb && c->ce_form && *c->ce_form == 1;
  if (ce_epilogue && fused_ce(c, B, &cw)) { // the forward pass of THIS step ran the softmax epilogue (same predicate)
    TcarOpt os = opt_of(c);
    if (anch)
      RET(tcar_ce_anchor_fold_o(B, g.N, c->ce_geo[0], c->ce_geo[1], cw.stats, cw.lab, bt->label, cw.rowstat, c->ce, c->ce_rowscale,
                                c->dl16h, g.Npad, c->ap16h, c->ap16l, c->aps16h, g.ldh + g.pt, g.ldh + g.pt, stream, &os));
    else
      RET(tcar_ce_finish_o(B, g.N, c->ce_geo[0], c->ce_geo[1], cw.stats, cw.lab, bt->label, cw.rowstat, c->ce, c->dl16h, g.Npad, stream, &os));
  }
  else if (c->scoring) RET(tcar_softmax_ce_bf16_o(B, g.N, c->logits, g.Npad, bt->label, c->ce, c->dl16h, nsb == 1 ? nullptr : c->dl16l, stream));
  else RET(tcar_softmax_ce(B, g.N, c->logits, g.Npad, bt->label, c->ce, stream));
  // (forking dE behind dX instead — dX then runs without dE beside it — was re-measured in round 4: dX is no faster alone, dE ends
  //  13 us later: 0.529 vs 0.516 ms per step, profiles/r04_ab_experiments.txt)
  // dE (aux stream) reads the softmax gradient the launch above has just written.  One-hot schedule with flag forks: dX is launched
  // FIRST and carries a START flag — a dX workgroup that runs has the softmax finish behind it, complete and released — and the aux
  // stream polls that word: no event record between the softmax finish and dX on the main chain (round 6: ~7 us of it, median, in
  // profiles/r05_main_stream_gaps.txt), no write-through plane (the round-3/4 form that lost: slot 12).  Else: an event.
  TcarSignal dx_start{};
  if (ohb && s2 && sB != stream) dx_start = fork_arm(c, FK_DXSTART);
  if (!dx_start.flag && s2 &&
      (hipEventRecord((hipEvent_t)c->ev[2], st) != hipSuccess || hipStreamWaitEvent(s2, (hipEvent_t)c->ev[2], 0) != hipSuccess))
    return TCAR_E_LAUNCH;
  // ---- chain B  (when it runs on the main stream it is the first user of the aux stream's prologue there)
  if (s2 && sB == stream && hipStreamWaitEvent(st, (hipEvent_t)c->ev[1], 0) != hipSuccess) return TCAR_E_LAUNCH;
  // optional HIP events around exactly the dE and dX launches (slot chosen by the forward pass; kind 1 = dX, 2 = dE)
  const int ei = (c->ev_n > 0 && c->ev_start && c->ev_stop && c->ev_cursor) ? c->ev_cursor[1] : -1;
  auto tick = [&](int kind, bool stop, void* s) {
    if (ei >= 0) (void)hipEventRecord((hipEvent_t)(stop ? c->ev_stop : c->ev_start)[kind * c->ev_n + ei], (hipStream_t)s);
  };
  const bool split_finish = fuse_finish && s2;
  auto chain_b = [&]() -> int {
    tick(2, false, sB);
    if (ohb) {
      // item block only; the time block leaves as per-candidate (||gy||^2, x . gy) pairs in the order of the inverted index
      TcarOpt ob = opt_of(c);
      RET(tcar_gemm_bf16_de_qz_o(g.N, (B + 31) & ~31, c->dl16h, g.Npad, (B + 127) & ~127, anch ? c->aps16h : c->ap16h, g.ldh + g.pt,
                                 (B + 127) & ~127, g.ldh, Gi, g.ldh, c->mwdhm, c->et_perm, c->tclip, c->qz, de_tile(c, g.N), sB, &ob));
    } else if (c->scoring) {
      // the time block goes out in the order of the inverted index (et_perm) so that its backward streams it
      TcarOpt ob = opt_of(c);
      RET(tcar_gemm_bf16_perm_o(2, g.N, g.ldh + g.pt, (B + 31) & ~31, c->dl16h, c->dl16l, g.Npad, (B + 127) & ~127, c->ap16h,
                                c->ap16l, g.ldh + g.pt, (B + 127) & ~127, Gi, g.ldh, d_et, g.pt, g.ldh, c->et_perm, g.ldt, nsb, 1,
                                sB, &ob));
    } else {  // dE = dlogits^T attout: item block and time block (content is frozen)
      tcar_gemm_desc_t p[2];
      p[0] = prob1(g.N, g.ldh, c->logits, g.Npad, c->attout, g.ek, B, Gi, g.ldh);
      p[1] = prob1(g.N, g.pt, c->logits, g.Npad, c->attout + g.ic, g.ek, B, d_et, g.pt);
      RET(small_gemm(c, 2, 2, p, sB));
    }
    tick(2, true, sB);
    // Fused single-rank step: the aux stream goes straight on to the candidate-side time backward (it needs only d_et),
    // while the negative rows and the dense item norm (they need Gi) are appended to the MAIN chain, which has slack
    // once dX has been given priority.  Rank-local backward: negative rows here, the rest in tcar_step_finish.
    if (split_finish && hipEventRecord((hipEvent_t)c->ev[4], s2) != hipSuccess) return TCAR_E_LAUNCH;         // dE done
    if (has_neg && !split_finish)
      RET(tcar_neg_scatter(&c->d, B, K, bt->neg, c->attout, c->neg_coef, Gi, c->neg_fb, c->ce, c->neg_weight, c->loss, sB));
    // (one-hot form: the candidate-side table gradients need dP of the dX chain too — they follow on this stream further down)
    if (fuse_finish && !ohb) RET(split_finish ? cand_time_backward(c, g, sB) : finish_dense_side(c, g, sB));
    // chain B done (read only where the main stream waits for the whole chain: not in the fused step's split finish — every
    // event record between two kernels costs its stream ~7 us)
    if (s2 && !split_finish && hipEventRecord((hipEvent_t)c->ev[3], (hipStream_t)sB) != hipSuccess) return TCAR_E_LAUNCH;
    return TCAR_OK;
  };
  auto chain_a_dx = [&]() -> int {
    tick(1, false, stream);
    if (ohb) {
      TcarOpt ox = opt_of(c);
      ox.start = dx_start;
      RET(tcar_gemm_bf16_dx_onehot_o(B, g.ic, g.Npad, c->dl16h, g.Npad, B, c->e16h, g.ek, g.Npad, c->oh16, 160, c->slabs, g.ic + 160,
                                     c->splitk, stream, &ox));
      if (dx_start.flag && !ox.started) return TCAR_E_ARG;      // (every form of this launcher publishes the start flag)
    } else if (c->scoring) {
      TcarOpt ox = opt_of(c);
      RET(tcar_gemm_bf16_perm_o(0, B, g.ek, g.Npad, c->dl16h, c->dl16l, g.Npad, B, c->e16h, c->e16l, g.ek, g.Npad, c->slabs, g.ek,
                                nullptr, 0, 0, nullptr, 0, nsb, c->splitk, stream, &ox));
    } else {
      RET(tcar_gemm_f32(0, B, g.ek, g.Npad, c->logits, g.Npad, c->E, g.ek, c->slabs, g.ek, nullptr, 0, 0, c->splitk, stream));
    }
    tick(1, true, stream);
    return TCAR_OK;
  };
  if (dx_start.flag) {
    // dX first (it carries the start flag), then the aux stream's poll in front of dE
    RET(chain_a_dx());
    TCAR_LAUNCH(poll_flag_kernel, dim3(1), dim3(64), 0, s2, (const unsigned*)dx_start.flag, dx_start.epoch, c->sig_dev + TCAR_SIG_ERR,
                c->sig_err_host, POLL_TICKS, (const unsigned*)nullptr, 0u, 0);
    TCAR_CHECK_LAUNCH();
  }
  RET(chain_b());
  // negative rows of the item gradient (sorted sum) + the loss: they need dE's item block and nothing of the main chain — on
  // the third stream (idle until the weight gradients) the moment dE lands, instead of on the main chain behind its small GEMMs
  const bool neg_s3 = split_finish && has_neg && sorted && c->stream3 && c->ev3;
  if (neg_s3) {
    hipStream_t s3n = (hipStream_t)c->stream3;
    if (hipStreamWaitEvent(s3n, (hipEvent_t)c->ev[4], 0) != hipSuccess) return TCAR_E_LAUNCH;
    RET(tcar_segsum_apply(&c->d, bt, c->segsum_ws, c->segsum_bytes, 1, nullptr, c->neg_coef, c->attout, g.ek, Gi, nullptr, nullptr,
                          c->ce, c->neg_fb, c->neg_weight, c->loss, (void*)s3n));
    if (hipEventRecord((hipEvent_t)c->ev[3], s3n) != hipSuccess) return TCAR_E_LAUNCH;
  }
  // ---- chain A
  if (!dx_start.flag) RET(chain_a_dx());
  // first use of the zeroed arena and of the negative term's forward outputs on the main stream
  if (s2 && hipStreamWaitEvent(st, (hipEvent_t)c->ev[1], 0) != hipSuccess) return TCAR_E_LAUNCH;
  // (Round 3 A/B: letting the chain of small kernels behind dX wait until dE has finished — every one of them runs ~2x slower
  // beside dE's 106-MB write stream — loses more in idle time than the faster kernels give back: 0.647 vs 0.620 ms per step.)
  // dattout = slabs summed + the negative term's part, through tanh' of both output transforms, + their bias gradients
  // order-fixed bias / residual-weight gradients (split-bf16 modes with the fused query chain and a row workspace): the
  // producers below leave the column sums to ONE tcar_colsum_det launch behind the weight-gradient GEMM
  const bool fusedq = c->scoring != 0;
  const bool detc = fusedq && c->gw_rows != nullptr;
  if (ohb) {    // ... and the one-hot columns: dP = their slab sum, expanded to the time columns of dattout on the spot
    TcarOpt orr = opt_of(c);
    orr.sig = fork_arm(c, FK_REDUCE);          // (dP leaves write-through when the launch carries the flag)
    const TcarRowFix fix{c->ce_rowscale, bt->label, c->E, (long)g.ek, c->mwdhm, g.N};
    if (anch) orr.rowfix = &fix;
    RET(tcar_reduce_dact_onehot_o(c->slabs, S, B, g.ic, g.ic + 160, has_neg ? c->negpart : nullptr, g.ic, c->attout, g.ek, c->tclip,
                                  c->dattout, g.ek, c->dP, detc ? nullptr : G(c, TCAR_V_O_B), detc ? nullptr : G(c, TCAR_V_OT_B), stream,
                                  &orr));
    (void)fork_commit(c, FK_REDUCE, orr);
    // candidate-side time-table gradients: on the aux stream behind dE (its (q, z) pairs: stream order) and behind this launch
    // (dP: a flag, no event on the main chain) — beside the session backward, ahead of the small tables' order-fixed pass
    // (its per-table norm pieces are folded by the small tables' last launch: no launch of their own)
    const TcarWait wdp{};
    RET(fork_go(c, FK_REDUCE, st, s2, c->ev[5]));
    tcar_grads_t gr;
    grads_of(c, gr);
    RET(tcar_cand_time_bwd_onehot_w(&c->d, B, c->inv_off, c->qz, c->dP, c->attout, g.ek, c->tclip, c->ct_ws, &gr, (void*)s2, wdp, 0));
  } else
    RET(tcar_splitk_reduce_dact(c->slabs, S, B, g.ek, g.ek, has_neg ? c->negpart : nullptr, g.ic, g.ic, c->attout, g.ek, 2,
                                c->dattout, detc ? nullptr : G(c, TCAR_V_O_B), g.ic, detc ? nullptr : G(c, TCAR_V_OT_B), stream));
  if (!has_neg && hipMemsetAsync(c->neg_fb, 0, (size_t)B * sizeof(float), st) != hipSuccess) return TCAR_E_LAUNCH;
  // dpooled = dattout W_o^T (both output transforms).  Split form (slab workspace + order-fixed pool backward): every 128-deep K
  // chunk is its own set of workgroups writing its own slab, the pool backward folds them while it loads dpooled
  const int nd_ic = units(g.ic), nd_pt = units(g.pt);
  const int64_t dstride = (int64_t)B * g.ek;
  // (diagnostic builds: the session-side backward GEMMs on plain bf16 operands under the hi-only backward precision)
  const bool bwd_hi = tcar_fixed::bwd_small_hi != 0 && c->scoring_bwd == 1 && fuse_finish;
  const bool dsplit = detc && c->proj_slabs && c->proj_slab_floats >= (nd_ic > nd_pt ? nd_ic : nd_pt) * dstride;
  {
    tcar_gemm_desc_t p[2];
    float* dp = dsplit ? c->proj_slabs : c->dpooled;
    p[0] = prob1(B, g.ic, c->dattout, g.ek, W(c, TCAR_V_O_W), g.ic, g.ic, dp, g.ek, nullptr, 0, 0, dsplit ? nd_ic : 1);
    p[1] = prob1(B, g.pt, c->dattout + g.ic, g.ek, W(c, TCAR_V_OT_W), g.pt, g.pt, dp + g.ic, g.ek, nullptr, 0, 0, dsplit ? nd_pt : 1);
    TcarOpt odp = opt_of(c);
    odp.hi_only = bwd_hi;
    RET(small_gemm(c, 1, 2, p, stream, &odp));
  }
  // query MLP backward (modules.py:138-139).  Split-bf16 modes: tanh' + bias gradient of query_trans2 ride in the pool
  // backward, relu' + bias gradient of query_trans1 in the epilogue of the GEMM that produces dq1, and that GEMM shares ONE
  // launch with the three input-gradient GEMMs of the projections (all four need only the pool backward's outputs); the
  // click-query input gradient (needs dq1) follows.  fp32 mode: the op-level sequence.
  // (The whole click-query MLP backward, dq -> dq1 -> dclick, as ONE launch on the third stream behind the pool backward's flag —
  //  tcar_query_mlp_bwd with both layers — measured slower in rounds 4 and 5; the driver runs only its layer-1 half, further down.)
  // (Weight gradients in two launches — eight of the nine problems behind the pool backward's flag — were measured twice, 5 us and
  //  15 us SLOWER per step: the early launch runs beside the main chain's input-gradient GEMM.  Removed in round 5.)
  if (detc) {
    TcarOpt opb = opt_of(c);
    RET(tcar_attn_pool_bwd_slabs_o(&c->d, B, T, c->x_icp, c->x_pt, c->pre1, c->pre2, c->q, W(c, TCAR_V_M_WRES), W(c, TCAR_V_S_WRES),
                                   c->alpha, dsplit ? c->proj_slabs : c->dpooled, dsplit ? nd_ic : 1, dsplit ? nd_pt : 1, dstride,
                                   c->dx_icp, c->dx_pt, c->dq, c->dpre1, c->dpre2, c->gw_rows, stream, &opb));
  }
  else
    RET(tcar_attn_pool_bwd_q(&c->d, B, T, c->x_icp, c->x_pt, c->pre1, c->pre2, c->q, W(c, TCAR_V_M_WRES),
                             W(c, TCAR_V_S_WRES), c->alpha, c->dpooled, c->dx_icp, c->dx_pt, c->dq, c->dpre1, c->dpre2,
                             G(c, TCAR_V_M_WRES), G(c, TCAR_V_S_WRES), fusedq ? G(c, TCAR_V_Q2_B) : nullptr, stream));
  if (fusedq) {
    tcar_gemm_desc_t p[4];
    p[0] = prob1(B, g.ldh, c->dq, g.ic, W(c, TCAR_V_Q2_W), g.ic, g.ic, c->dq1, g.ldh);
    p[0].dact = 1; p[0].dact_y = c->q1; p[0].ld_dact_y = g.ldh; p[0].colsum = detc ? nullptr : G(c, TCAR_V_Q1_B);
    // input gradients (only the ITEM half of dX_ic: content is frozen)
    p[1] = prob1(BT, g.ldh, c->dpre1, g.ldh, W(c, TCAR_V_M_WIN), g.ldh, g.ldh, c->dx_icp, g.ic, nullptr, 0, 1);
    p[2] = prob1(BT, g.ldt, c->dpre1, g.ldh, W(c, TCAR_V_M_WINT), g.ldh, g.ldh, c->dx_act, g.ldt);
    p[3] = prob1(BT, g.pt, c->dpre2, g.ldh, W(c, TCAR_V_S_WIN), g.ldh, g.ldh, c->dx_pt, g.pt, nullptr, 0, 1);
    TcarOpt oi = opt_of(c);
    oi.hi_only = bwd_hi;
    if (s2 && fuse_finish && c->stream3 && c->ev3) oi.sig = fork_arm(c, FK_INGRAD);     // the third stream's fork below
    else fork_disarm(c, FK_INGRAD);
    RET(small_gemm(c, 1, 4, p, stream, &oi));
    (void)fork_commit(c, FK_INGRAD, oi);
  } else {
    fork_disarm(c, FK_INGRAD);
    RET(tcar_dact_colsum(B, g.ic, g.ic, c->q, c->dq, G(c, TCAR_V_Q2_B), 2, stream));
    {
      tcar_gemm_desc_t p = prob1(B, g.ldh, c->dq, g.ic, W(c, TCAR_V_Q2_W), g.ic, g.ic, c->dq1, g.ldh);
      RET(small_gemm(c, 1, 1, &p, stream));
    }
    RET(tcar_dact_colsum(B, g.ldh, g.ldh, c->q1, c->dq1, G(c, TCAR_V_Q1_B), 1, stream));
  }
  // Everything the weight gradients need exists now; they run beside the main stream's input gradients and row scatter:
  // on the third stream when the context has one (fused step: the aux stream is still busy with the candidate-time
  // backward, which only became ready when dE finished), else on the aux stream behind chain B.
  hipStream_t s3 = (s2 && fuse_finish && c->stream3 && c->ev3) ? (hipStream_t)c->stream3 : nullptr;
  TcarOpt ow = opt_of(c);
  ow.hi_only = bwd_hi;
  if (s3) {
    sW = (void*)s3;      // ordered behind the main chain so far AND behind the aux stream's arena memset (ev[1])
    RET(fork_go(c, FK_INGRAD, st, s3, c->ev[0]));        // (flagged small-GEMM launches store write-through)
    // (with the negative rows on this stream it already waited for dE — ev[4], recorded on the aux stream BEHIND the arena zero —
    // and a wait for a completed event still costs the stream a ~6-us barrier packet)
    if (!neg_s3 && hipStreamWaitEvent(s3, (hipEvent_t)c->ev[1], 0) != hipSuccess) return TCAR_E_LAUNCH;
  } else if (s2 && (hipEventRecord((hipEvent_t)c->ev[0], st) != hipSuccess || hipStreamWaitEvent(s2, (hipEvent_t)c->ev[0], 0) != hipSuccess))
    return TCAR_E_LAUNCH;
  // Order-fixed small tables (sorted mode, which implies an aux stream): they — and the click-query input gradient, which only
  // they consume — run on the aux stream behind the candidate-time backward, beside the rest of the main chain
  const bool det_small = sorted && tn(c).det_small != 0;
  // The click-query input gradient dclick = dq1 Wq1^T is consumed by the small tables' pass (aux stream) ONLY: with the
  // order-fixed small tables it runs THERE, right in front of them, behind a second poll of the input-gradient launch's flag
  // (it needs dq1 of that launch) — off the main chain (round 4: 17 us) and off the third stream, whose weight gradients, column
  // sums and norms are the step's last chain
  const bool dclick_aux = fusedq && det_small && s3 != nullptr;
  RET(weight_grads(c, g, B, BT, sW, &ow));
  // column sums and dense norms are the last two launches of the step's last chain: ONE launch when the context has the fold scratch
  // (optim.hip: colsum_sqnorm_kernel)
  const bool cs_fused = detc && fuse_finish && s2 && c->fold_scratch;
  if (detc && !cs_fused) RET(det_colsums(c, g, B, sW));
  // the dense-weight norms need nothing from the row scatter (tables are normed through their row pieces, S5): with an
  // aux stream they follow the weight gradients there, beside the scatter
  // The step's LAST join (aux + third stream into the main one, in front of the next update): two event waits cost the main
  // stream two barrier packets behind whichever chain ends last (16-20 us); with flag forks the last launch of either side
  // stream publishes a flag and ONE poll on the main stream waits for both.
  const int tmask = tn(c).flag_fork;
  const bool tail_flags = s3 && fuse_finish && det_small && ((tmask >> FK_TAIL2) & 1) && ((tmask >> FK_TAIL3) & 1);
  bool tail3 = false, tail2 = false;
  if (fuse_finish && s2) {
    TcarOpt o3 = opt_of(c);
    if (tail_flags) o3.sig = fork_arm(c, FK_TAIL3);
    if (c->fold_scratch) { o3.scratch = c->fold_scratch; o3.scratch_words = c->fold_scratch_words; }
    bool fused_done = false;
    if (cs_fused) {
      tcar_colsum_t cs[6];
      det_colsum_list(c, g, B, cs);
      const int rc = tcar_colsum_sqnorm_o(c->Gx, &c->segs_dense, 6, cs, c->sqn_dense, sW, &o3);
      if (rc == TCAR_OK) fused_done = true;
      else if (rc != TCAR_E_ARG) return rc;
      else { o3.carried = false; RET(det_colsums(c, g, B, sW)); }       // (a layout the fused form does not take: the two launches)
    }
    if (!fused_done) RET(tcar_sqnorm_o(c->Gx, &c->segs_dense, c->sqn_dense, sW, &o3));
    if (tail_flags) tail3 = fork_commit(c, FK_TAIL3, o3);
  }
  if (s3 && hipEventRecord((hipEvent_t)c->ev3, s3) != hipSuccess) return TCAR_E_LAUNCH;
  // (with the order-fixed small tables the aux stream's "done" event is recorded behind them, below)
  if (s2 && !det_small && hipEventRecord((hipEvent_t)c->ev[2], s2) != hipSuccess) return TCAR_E_LAUNCH;
  if (dclick_aux) {
    // (launched on the aux stream below, in front of the small tables)
  } else if (fusedq) {   // the click-query input gradient (the projections' input gradients went with dq1)
    tcar_gemm_desc_t p = prob1(B, g.ct, c->dq1, g.ldh, W(c, TCAR_V_Q1_W), g.ldh, g.ldh, c->dclick, g.ct);
    TcarOpt od = opt_of(c);
    if (det_small) od.sig = fork_arm(c, FK_DCLICK);
    else fork_disarm(c, FK_DCLICK);
    RET(small_gemm(c, 1, 1, &p, stream, &od));
    (void)fork_commit(c, FK_DCLICK, od);
  } else {  // input gradients (only the ITEM half of dX_ic: content is frozen)
    fork_disarm(c, FK_DCLICK);
    tcar_gemm_desc_t p[4];
    p[0] = prob1(B, g.ct, c->dq1, g.ldh, W(c, TCAR_V_Q1_W), g.ldh, g.ldh, c->dclick, g.ct);
    p[1] = prob1(BT, g.ldh, c->dpre1, g.ldh, W(c, TCAR_V_M_WIN), g.ldh, g.ldh, c->dx_icp, g.ic, nullptr, 0, 1);
    p[2] = prob1(BT, g.ldt, c->dpre1, g.ldh, W(c, TCAR_V_M_WINT), g.ldh, g.ldh, c->dx_act, g.ldt);
    p[3] = prob1(BT, g.pt, c->dpre2, g.ldh, W(c, TCAR_V_S_WIN), g.ldh, g.ldh, c->dx_pt, g.pt, nullptr, 0, 1);
    RET(small_gemm(c, 1, 4, p, stream));
  }
  if (det_small) {
    tcar_tables_t tab;
    tcar_grads_t gr;
    tables_of(c, tab);
    grads_of(c, gr);
    // row pieces (+ chunk partials of long buckets) in the context's own workspace, else the second half of the segsum tail
    float* rowq = c->small_det_ws ? c->small_det_ws : (float*)((char*)c->segsum_ws + c->segsum_bytes - 2048);
    const int64_t rowq_floats = c->small_det_ws ? c->small_det_ws_floats : 512;
    // on the AUX stream: it is idle once the candidate-time backward is through (the third stream still holds the weight
    // gradients, the column sums and the dense norms); the final join below waits for ev[2], re-recorded here
    if (dclick_aux) {
      RET(fork_go(c, FK_INGRAD, st, s2, c->ev[5]));       // (the same flag the third stream polled: dq1, dx_* of that launch)
      if (g.ldh == 256 && g.ldt == 64) {
        // dclick = dq1 Wq1^T as ONE fp32 launch of whole-row dots (query.hip: the layer-1 half of the click-query backward): 7 us
        // where the 16-workgroup small GEMM walks four serial 64-deep stages (20 us), on the chain that ends the step
        RET(tcar_query_mlp_bwd_o(&c->d, B, nullptr, nullptr, W(c, TCAR_V_Q1_W), nullptr, c->dq1, c->dclick, (void*)s2, nullptr));
      } else {
        tcar_gemm_desc_t p = prob1(B, g.ct, c->dq1, g.ldh, W(c, TCAR_V_Q1_W), g.ldh, g.ldh, c->dclick, g.ct);
        RET(small_gemm(c, 1, 1, &p, (void*)s2));
      }
    } else {
      RET(fork_go(c, FK_DCLICK, st, s2, c->ev[5]));
    }
    TcarOpt o2 = opt_of(c);
    if (tail3) o2.sig = fork_arm(c, FK_TAIL2);
    RET(tcar_small_tables_bwd_det_o(&c->d, &tab, bt, c->dx_icp, c->dx_pt, c->dx_act, c->dclick, &gr, rowq, (void*)s2, &o2,
                                    ohb ? tcar_cand_pieces(&c->d, c->ct_ws) : nullptr, rowq_floats));
    if (tail3) tail2 = fork_commit(c, FK_TAIL2, o2);
    if (hipEventRecord((hipEvent_t)c->ev[2], s2) != hipSuccess) return TCAR_E_LAUNCH;
  }
  if (split_finish) {   // Gi is complete once dE has landed: negative rows (+ loss), then its norm BEFORE any row scatter (S5)
    if (hipStreamWaitEvent(st, (hipEvent_t)(neg_s3 ? c->ev[3] : c->ev[4]), 0) != hipSuccess) return TCAR_E_LAUNCH;
    if (neg_s3) {
      // (done on the third stream)
    } else if (has_neg && sorted) {
      RET(tcar_segsum_apply(&c->d, bt, c->segsum_ws, c->segsum_bytes, 1, nullptr, c->neg_coef, c->attout, g.ek, Gi, nullptr, nullptr,
                            c->ce, c->neg_fb, c->neg_weight, c->loss, stream));
    } else if (has_neg) {
      RET(tcar_neg_scatter(&c->d, B, K, bt->neg, c->attout, c->neg_coef, Gi, c->neg_fb, c->ce, c->neg_weight, c->loss, stream));
    }
    // sorted: the block partials of ||Gi||^2 (S5: BEFORE any session row lands) ride in the launch of the item-row gradients
    // below (tcar_gather_clip_bwd_sqnorm); the session-list pass folds them
    if (!sorted) RET(item_norm(c, g, stream));
  }
  // The row scatter needs the item norm (same stream) but NOT the candidate-time backward: both only add (atomically) into
  // the time-table gradients, and the final join below covers the whole aux stream.  Without the split, wait for chain B.
  if (s2 && !split_finish && hipStreamWaitEvent(st, (hipEvent_t)c->ev[3], 0) != hipSuccess) return TCAR_E_LAUNCH;
  if (fuse_finish) {
    // sparse rows and per-row norm pieces (after the dense item norm of chain B), then the dense-weight norms
    tcar_tables_t tab;
    tcar_grads_t gr;
    tables_of(c, tab);
    grads_of(c, gr);
    if (sorted) {
      // the gather backward WRITES the session sources' item rows and their squared norms (plain stores); the segmented sum
      // adds the rows into Gi in sorted order and folds the norms — and the dense norm's partials — in a fixed order
      gr.rows_out = tcar_segsum_rows_buffer(&c->d, bt, c->segsum_ws);
      gr.norms_out = tcar_segsum_norms_buffer(&c->d, bt, c->segsum_ws);
      gr.skip_small = det_small ? 1 : 0;
      if (split_finish)
        RET(tcar_gather_clip_bwd_sqnorm(&c->d, &tab, bt, c->dx_icp, c->dx_pt, c->dx_act, c->dclick, &gr, Gi, (int64_t)g.N * g.ldh,
                                        c->segsum_ws, c->segsum_bytes, stream));
      else
        RET(tcar_gather_clip_bwd(&c->d, &tab, bt, c->dx_icp, c->dx_pt, c->dx_act, c->dclick, &gr, stream));
      RET(tcar_segsum_apply(&c->d, bt, c->segsum_ws, c->segsum_bytes, 0, gr.rows_out, nullptr, nullptr, 0, Gi,
                            c->Gx + c->arena_n + c->slot_item, c->sqn_dense + c->slot_item, nullptr, nullptr, 0.f, nullptr, stream));
    } else {
      RET(tcar_gather_clip_bwd(&c->d, &tab, bt, c->dx_icp, c->dx_pt, c->dx_act, c->dclick, &gr, stream));
    }
  }
  if (tail2 && tail3) {
    const ForkSlot& f2 = fork_host(c)->slot[FK_TAIL2];
    const ForkSlot& f3 = fork_host(c)->slot[FK_TAIL3];
    // light poll (no L2 write-back): the flagged launches publish their OWN results with atomics (norm slots; the small tables'
    // rows) or write-through stores (colsum_sqnorm_kernel: the six bias / residual-weight gradients), and everything else the
    // update reads was written by EARLIER launches of those streams, released when they ended
    TCAR_LAUNCH(poll_flag_kernel, dim3(1), dim3(64), 0, st, (const unsigned*)f2.sig.flag, f2.sig.epoch, c->sig_dev + TCAR_SIG_ERR,
                c->sig_err_host, POLL_TICKS, (const unsigned*)f3.sig.flag, f3.sig.epoch, 0);
    TCAR_CHECK_LAUNCH();
  } else {
    if (s2 && hipStreamWaitEvent(st, (hipEvent_t)c->ev[2], 0) != hipSuccess) return TCAR_E_LAUNCH;    // the aux stream is done
    if (s3 && hipStreamWaitEvent(st, (hipEvent_t)c->ev3, 0) != hipSuccess) return TCAR_E_LAUNCH;       // weight gradients are in
  }
  if (fuse_finish && !s2) RET(tcar_sqnorm(c->Gx, &c->segs_dense, c->sqn_dense, stream));
  return TCAR_OK;
}

// clip norm of the dense item block BEFORE the sparse rows are scattered in, then the candidate-side
// time backward (static inverted index)
int item_norm(const tcar_ctx_t* c, const Geo& g, void* stream) {
  tcar_segments_t one = {};
  one.nseg = 1; one.off[0] = 0; one.len[0] = (int64_t)g.N * g.ldh; one.slot[0] = c->slot_item;
  return tcar_sqnorm(c->big, &one, c->sqn_dense, stream);
}
int cand_time_backward(const tcar_ctx_t* c, const Geo& g, void* stream) {
  tcar_grads_t gr;
  grads_of(c, gr);
  const float* tt[5];
  for (int k = 0; k < 5; ++k) tt[k] = W(c, TCAR_V_MONTH + k);
  return tcar_cand_time_bwd_indexed(&c->d, tt, c->inv_n, c->inv_off, c->big + (size_t)g.N * g.ldh,
                                    (c->scoring && c->et_perm) ? 1 : 0, c->ct_ws, &gr, stream);
}
int finish_dense_side(const tcar_ctx_t* c, const Geo& g, void* stream) {
  RET(item_norm(c, g, stream));
  tcar_grads_t gr;
  grads_of(c, gr);
  const float* tt[5];
  for (int k = 0; k < 5; ++k) tt[k] = W(c, TCAR_V_MONTH + k);
  return tcar_cand_time_bwd_indexed(&c->d, tt, c->inv_n, c->inv_off, c->big + (size_t)g.N * g.ldh,
                                    (c->scoring && c->et_perm) ? 1 : 0, c->ct_ws, &gr, stream);
}
}  // namespace

extern "C" int tcar_step_backward_local(const tcar_ctx_t* c, const tcar_batch_t* bt, void* stream) {
  return backward_impl(c, bt, stream, false);
}

extern "C" int tcar_step_finish(const tcar_ctx_t* c, const tcar_batch_t* bt, void* stream) {
  RET(check_ctx(c, bt));
  const Geo g(c->d);
  RET(finish_dense_side(c, g, stream));
  tcar_tables_t tab;
  tcar_grads_t gr;
  tables_of(c, tab);
  grads_of(c, gr);
  RET(tcar_gather_clip_bwd(&c->d, &tab, bt, c->dx_icp, c->dx_pt, c->dx_act, c->dclick, &gr, stream));
  return tcar_sqnorm(c->Gx, &c->segs_dense, c->sqn_dense, stream);
}

extern "C" int tcar_step_update(const tcar_ctx_t* c, float lr_t, void* stream) {
  if (!c) return TCAR_E_ARG;
  const Geo g(c->d);
  const float* pieces = c->Gx + c->arena_n;
  return tcar_clip_adam_all(c->W, c->Gx, c->M, c->V, &c->segs_all, c->E, g.ek, c->big, c->Mi, c->Vi, g.N, g.ldh, c->slot_item,
                            c->sqn_dense, pieces, c->use_dense, c->clip, lr_t, c->b1, c->b2, c->eps,
                            c->scoring ? c->e16h : nullptr, c->scoring ? c->e16l : nullptr, g.ek, stream);
}

// Which form would a fused training step of `bt` take on this context?  The predicates the driver itself evaluates, for tools
// that label measurements (bench.py's roofline entries): form[0] softmax epilogue in the logits GEMM, [1] one-hot time segment of
// the logits GEMM, [2] one-hot form of the two gradient GEMMs, [3] sorted (order-fixed) item-row sum, [4] anchored softmax form (no
// rescale pass over the plane).  Launches nothing.
extern "C" int tcar_step_form(const tcar_ctx_t* c, const tcar_batch_t* bt, int32_t* form /*host, 5 ints*/) {
  if (!form) return TCAR_E_ARG;
  RET(check_ctx(c, bt));
  form[0] = (c->scoring && fused_ce(c, bt->B, nullptr)) ? 1 : 0;
  form[1] = (form[0] && onehot_fwd(c, bt->B)) ? 1 : 0;
  form[2] = onehot_bwd(c, bt) ? 1 : 0;
  form[3] = sorted_rows(c, bt) ? 1 : 0;
  form[4] = ce_anchored(c, bt) ? 1 : 0;
  return TCAR_OK;
}

extern "C" int tcar_train_step(const tcar_ctx_t* c, const tcar_batch_t* bt, int refresh_time, float lr_t, void* stream) {
  RET(forward_impl(c, bt, refresh_time, stream, -1.f, true));
  RET(backward_impl(c, bt, stream, true, true));
  return tcar_step_update(c, lr_t, stream);
}

extern "C" int tcar_train_step_deferred(const tcar_ctx_t* c, const tcar_batch_t* bt, int refresh_time, int pending,
                                        float lr_pending, void* stream) {
  RET(check_ctx(c, bt));
  float rest_lr = -1.f;
  if (pending) {
    hipStream_t s2 = aux_stream(c);
    if (s2 && refresh_time && c->adam_bitmap) {
      // split update, issued by forward_impl: EARLY part on the main stream (arena + the item rows this batch gathers), the
      // REST on the aux stream beside it and the session forward
      rest_lr = lr_pending;
    } else {
      RET(tcar_step_update(c, lr_pending, stream));
    }
  }
  RET(forward_impl(c, bt, refresh_time, stream, rest_lr, true));
  return backward_impl(c, bt, stream, true, true);
}

extern "C" int tcar_eval_step(const tcar_ctx_t* c, const tcar_batch_t* bt, int refresh_time, int k, void* stream) {
  RET(tcar_step_forward(c, bt, refresh_time, stream));
  const Geo g(c->d);
  if (g.Npad <= 512L * 4 * 24)      // one read of the score matrix: rank, top-k and CE from the row-resident kernel
    return tcar_eval_rows(bt->B, g.N, c->logits, g.Npad, bt->label, k, c->rank, c->topk, c->ce, stream);
  RET(tcar_rank_topk(bt->B, g.N, c->logits, g.Npad, bt->label, k, c->rank, c->topk, stream));
  return tcar_softmax_ce(bt->B, g.N, c->logits, g.Npad, bt->label, c->ce, stream);
}


// ---------------------------------------------------------------------------------------------------------------------
// Catalog-sharded step (sharded.py): the same op-level launchers, cut at the points where the ranks exchange data.
extern "C" int tcar_step_session_forward(const tcar_ctx_t* c, const tcar_batch_t* bt, void* stream) {
  RET(check_ctx(c, bt));
  if (!c->scoring) return TCAR_E_ARG;                   // split-bf16 modes only
  const Geo g(c->d);
  RET(session_forward(c, bt, g, stream, false, -1, [](int, TcarOpt*) { return (int)TCAR_OK; }));
  if (bt->K > 0 && bt->neg && c->neg_coef && c->negpart)
    RET(tcar_neg_fwd(&c->d, bt->B, bt->K, c->E, bt->neg, c->attout, c->neg_weight, c->neg_fb, c->neg_coef, c->negpart, stream));
  return TCAR_OK;
}

extern "C" int tcar_shard_begin(const tcar_ctx_t* c, const tcar_batch_t* bt, int cap, int Kc, float* head, int64_t ld_head,
                                int refresh_time, int n_loc, float lr_pending, void* stream) {
  if (!c || !c->scoring || !c->Gx || !c->sqn_dense || !head || cap <= 0 || n_loc <= 0) return TCAR_E_ARG;
  if (bt) RET(check_ctx(c, bt));
  const Geo g(c->d);
  hipStream_t st = (hipStream_t)stream, s2 = aux_stream(c);
  // A pending SPLIT update (one rank: the shard is the whole table, nothing is exchanged behind the update): the single-GPU step's
  // early pass here — arena + the item rows this batch gathers —, the REST pass over every other row, then the arena zero, on the aux
  // stream beside the session forward; ev[5] = "aux stream done" orders the negative term (it reads rows of the rest pass) and
  // everything after it behind both.  With more ranks the owned rows are exchanged before the next gather: the caller updates
  // inside the step (tcar_step_update) and passes lr_pending < 0.
  const bool split_update = lr_pending >= 0.f;
  if (split_update) {
    if (!bt || !s2 || !c->adam_bitmap || n_loc != (int)g.N) return TCAR_E_ARG;
    const float* pieces = c->Gx + c->arena_n;
    RET(tcar_clip_adam_early(c->W, c->Gx, c->M, c->V, &c->segs_all, c->E, g.ek, c->big, c->Mi, c->Vi, g.N, g.ldh, c->slot_item,
                             c->sqn_dense, pieces, c->use_dense, c->clip, lr_pending, c->b1, c->b2, c->eps, c->e16h, c->e16l, g.ek,
                             bt->seq, (int64_t)bt->B * bt->T, c->adam_bitmap, stream));
  }
  if ((refresh_time || split_update) && s2) {
    // the candidate-side time planes of the shard depend on the time tables only: rebuilt on the aux stream beside the session
    // forward pass and the first all-gather (tcar_shard_score joins)
    if (hipEventRecord((hipEvent_t)c->ev[0], st) != hipSuccess || hipStreamWaitEvent(s2, (hipEvent_t)c->ev[0], 0) != hipSuccess)
      return TCAR_E_LAUNCH;
    if (refresh_time) {
      tcar_dims_t dc = c->d;
      dc.n_items = n_loc;
      const float* tt[5];
      for (int k = 0; k < 5; ++k) tt[k] = W(c, TCAR_V_MONTH + k);
      RET(tcar_cand_time_fwd_bf16(&dc, tt, c->mwdhm, nullptr, c->e16h, c->e16l, (void*)s2));
    }
    if (split_update) {
      const float* pieces = c->Gx + c->arena_n;
      RET(tcar_clip_adam_rest_keep_o(c->E, g.ek, c->big, c->Mi, c->Vi, g.N, g.ldh, c->slot_item, c->sqn_dense, pieces, c->use_dense,
                                     c->clip, lr_pending, c->b1, c->b2, c->eps, c->e16h, c->e16l, g.ek, c->adam_bitmap, (void*)s2,
                                     tcar_fixed::rest_grid));
      RET(zero_arena(c, s2));              // (behind the rest pass: it reads the norm slots and pieces the zero clears)
    }
    if (hipEventRecord((hipEvent_t)c->ev[5], s2) != hipSuccess) return TCAR_E_LAUNCH;
    if (split_update &&        // the marks are cleared BEHIND the event the main stream waits for (the clear is a launch of its own)
        hipMemsetAsync(c->adam_bitmap, 0, (size_t)(((g.N + 31) / 32 + 15) & ~15) * sizeof(uint32_t), s2) != hipSuccess)
      return TCAR_E_LAUNCH;
  }
  if (!split_update) RET(zero_arena(c, st));
  if (!bt) return tcar_shard_pack_head(0, cap, g.ek, 0, Kc, nullptr, nullptr, nullptr, nullptr, head, ld_head, stream);
  RET(session_forward(c, bt, g, stream, false, -1, [](int, TcarOpt*) { return (int)TCAR_OK; }));
  // (split update: the negative term gathers item rows the rest pass may still be writing — and the arena is zeroed there)
  if (split_update && hipStreamWaitEvent(st, (hipEvent_t)c->ev[5], 0) != hipSuccess) return TCAR_E_LAUNCH;
  const bool has_neg = bt->K > 0 && bt->neg && c->neg_coef && c->negpart;
  if (has_neg)
    RET(tcar_neg_fwd(&c->d, bt->B, bt->K, c->E, bt->neg, c->attout, c->neg_weight, c->neg_fb, c->neg_coef, c->negpart, stream));
  return tcar_shard_pack_head(bt->B, cap, g.ek, has_neg ? bt->K : 0, Kc, c->attout, bt->label, has_neg ? c->neg_coef : nullptr,
                              has_neg ? bt->neg : nullptr, head, ld_head, stream);
}

namespace {
// The shard runs the benchmarked single-GPU schedule (softmax epilogue + one-hot forms) when the context carries its workspaces,
// sized for the world*cap session rows of the exchange, and the precision is the benchmarked one (bf16x3 forward, hi-only backward)
bool shard_onehot(const tcar_ctx_t* c, const tcar_shard_t* s, CeWs* w) {
  tcar_ctx_t cc = *c;
  cc.d.n_items = s->n_loc;
  return c->scoring == 3 && c->d.ldt == 64 && fused_ce(&cc, s->world * s->cap, w) && c->ce_geo && c->oh16 && c->p16h && c->p16l &&
         c->tclip && c->dP && c->qz && c->inv_off && c->ct_ws && c->mwdhm && tn(c).onehot_time >= 2;
}
// anchored softmax form on the shard (score.hip: ce_anchor_apply_kernel): the shard's one-hot schedule with the two buffers of the form
// in the shard descriptor
bool shard_anchored(const tcar_ctx_t* c, const tcar_shard_t* s) {
  return tn(c).fused_ce >= 2 && s->aps16h && s->scale2 && s->n_total >= s->n0 + s->n_loc;
}
int check_shard(const tcar_ctx_t* c, const tcar_shard_t* s) {
  if (!c || !s || !c->scoring || s->world <= 0 || s->cap <= 0 || s->n_loc <= 0 || s->n0 < 0) return TCAR_E_ARG;
  if (!s->att_all || !s->lab_all || !s->stats || !s->lse || !s->ce || !s->a16h || !s->a16l || !s->ap16h ||
      !s->ap16l || !s->dl16h || !s->dl16l || !s->slabs || !s->dx || !c->e16h || !c->e16l || !c->big || !c->et_perm)
    return TCAR_E_ARG;
  if (!s->logits && !shard_onehot(c, s, nullptr)) return TCAR_E_ARG;
  return TCAR_OK;
}
}  // namespace

// Which form do the shard pieces take for this context / descriptor (the predicates tcar_shard_score / _backward evaluate; launches
// nothing): form[0] the one-hot schedule with the softmax epilogue, form[1] the anchored softmax form (no rescale pass)
extern "C" int tcar_shard_form(const tcar_ctx_t* c, const tcar_shard_t* s, int32_t* form /*host, 2 ints*/) {
  if (!form) return TCAR_E_ARG;
  RET(check_shard(c, s));
  form[0] = shard_onehot(c, s, nullptr) ? 1 : 0;
  form[1] = (form[0] && shard_anchored(c, s)) ? 1 : 0;
  return TCAR_OK;
}

extern "C" int tcar_shard_score(const tcar_ctx_t* c, const tcar_shard_t* s, int refresh_time, void* stream) {
  RET(check_shard(c, s));
  const Geo g(c->d);
  const int Bq = s->world * s->cap, nl = s->n_loc, nlpad = (nl + 127) & ~127;
  tcar_dims_t dc = c->d;
  dc.n_items = nl;
  if (refresh_time) {
    if (hipStream_t s2 = aux_stream(c)) {      // tcar_shard_begin issued the refresh on the aux stream
      (void)s2;
      if (hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)c->ev[5], 0) != hipSuccess) return TCAR_E_LAUNCH;
    } else {
      const float* tt[5];
      for (int k = 0; k < 5; ++k) tt[k] = W(c, TCAR_V_MONTH + k);
      RET(tcar_cand_time_fwd_bf16(&dc, tt, c->mwdhm, nullptr, c->e16h, c->e16l, stream));
    }
  }
  if (s->ld_att > g.ek)     // packed exchange rows: label / negatives / coefficient ride behind attout
    RET(tcar_shard_unpack_head(Bq, g.ek, s->head_K, s->att_all, s->ld_att, const_cast<int32_t*>(s->lab_all), s->coef_all, s->neg_all,
                               stream));
  RET(tcar_split_bf16(s->att_all, s->ld_att ? s->ld_att : g.ek, Bq, g.ek, s->a16h, s->a16l, g.ek, s->ap16h, s->ap16l, g.ldh + g.pt, g.ldh, g.ic, stream));
  CeWs w;
  if (shard_onehot(c, s, &w)) {
    // the benchmarked schedule on the shard: time scores through the one-hot contraction (+ the clipped table rows for the
    // backward), ONE GEMM over 2 ldh + 160 columns whose epilogue leaves exp(x - group max) as the plane that becomes dlogits and
    // per-group (max, sum) pairs — no [W*cap, n_loc] fp32 logits —, then the shard's row of the statistics exchange
    const float* tt[5];
    for (int k = 0; k < 5; ++k) tt[k] = W(c, TCAR_V_MONTH + k);
    const int64_t lda = s->ld_att ? s->ld_att : g.ek;
    RET(tcar_time_scores_clip(&dc, tt, Bq, s->att_all, lda, c->p16h, c->p16l, 160, c->tclip, stream));
    TcarOpt ol = opt_of(c);
    ol.lab_off = s->n0;
    ol.lab_window = 1;
    if (shard_anchored(c, s)) {
      // anchored form: every shard computes the SAME anchor of every session — attout . E[label] over the item | content columns,
      // from the gathered attout rows and its own copy of the whole fp32 table (identical on every rank after the row exchange) — and
      // puts it into the anchor column of the time-score planes: the GEMM subtracts it, all shards' exponentials share one reference
      // per row, and the statistics exchange adds plain sums.  `c` is the SHARD's context here (c->E = its first row) or the rank's
      const float* e_all = c->E - ((int)g.N == nl ? (int64_t)s->n0 * g.ek : 0);
      RET(tcar_anchor_scores(g.ldh, Bq, s->att_all, lda, s->lab_all, e_all, g.ek, s->n_total, c->p16h, c->p16l, 160, stream));
      ol.anchored = true;
    }
    int32_t gw = 0, ng = 0;
    RET(tcar_gemm_bf16_ce_o(Bq, nl, g.ic + 160, s->a16h, s->a16l, g.ek, Bq, c->e16h, c->e16l, g.ek, nlpad, g.ic, c->p16h, c->p16l,
                            c->oh16, 160, s->dl16h, nlpad, (Bq + 127) & ~127, w.stats, w.stats_floats, s->lab_all, w.lab, c->scoring,
                            &gw, &ng, stream, &ol));
    c->ce_geo[0] = gw; c->ce_geo[1] = ng;
    return tcar_ce_shard_stats_a(Bq, ng, w.stats, w.lab, s->lab_all, s->n0, nl, s->stats, ol.anchored ? 1 : 0, stream);
  }
  RET(tcar_gemm_bf16(1, Bq, nl, g.ek, s->a16h, s->a16l, g.ek, Bq, c->e16h, c->e16l, g.ek, nlpad, s->logits, nlpad, nullptr, 0, 0,
                     c->scoring, 1, stream));
  return tcar_softmax_stats(Bq, nl, s->logits, nlpad, s->lab_all, s->n0, s->stats, stream);
}

extern "C" int tcar_shard_backward(const tcar_ctx_t* c, const tcar_shard_t* s, const float* stats_all, void* stream) {
  RET(check_shard(c, s));
  if (!stats_all) return TCAR_E_ARG;
  const Geo g(c->d);
  const int Bq = s->world * s->cap, nl = s->n_loc, nlpad = (nl + 127) & ~127, Bp = (Bq + 127) & ~127;
  const int nsb = c->scoring_bwd ? c->scoring_bwd : c->scoring;
  CeWs w;
  fork_disarm(c, FK_REDUCE);          // (tcar_shard_finish asks this slot: never a leftover of another step's form)
  if (shard_onehot(c, s, &w)) {
    // lse from the exchanged statistics, the exp plane rescaled in place to dlogits of the shard's columns; then the one-hot forms:
    // dE (aux stream) keeps its item block and leaves (||gy||^2, x . gy) pairs for the time block; dX contracts the shard against
    // [E_item | E_content | OH]; the slab reduce expands dP to the time columns WITHOUT tanh' (it follows the exchange)
    const bool anch = shard_anchored(c, s);
    if (anch) RET(tcar_softmax_combine_anchored(s->world, Bq, stats_all, s->lab_all, s->lse, s->ce, w.rowstat, stream));
    else RET(tcar_softmax_combine_rowstat(s->world, Bq, stats_all, s->lab_all, s->lse, s->ce, w.rowstat, stream));
    if (anch)      // no pass over the plane: row scales, the label's -1 where the label lives here, the scaled attout plane of dE
      RET(tcar_ce_anchor_apply_o(Bq, nl, w.rowstat, s->lab_all, s->n0, s->dl16h, nlpad, s->ap16h, s->ap16l, s->aps16h, g.ldh + g.pt,
                                 g.ldh + g.pt, s->scale2, stream));
    else
      RET(tcar_ce_rescale(Bq, nl, c->ce_geo[0], c->ce_geo[1], w.stats, w.rowstat, s->lab_all, s->n0, 1, s->dl16h, nlpad, stream));
    hipStream_t st = (hipStream_t)stream, s2 = aux_stream(c);
    // dE (aux stream) behind the rescale through an EVENT, dE launched first.  (backward_impl orders dE behind dX's START flag instead;
    // here that form — dX first, the poll, then dE — measured 14 us per step SLOWER on the one-rank shard, 0.548 against 0.534 ms in
    // 4 of 4 interleaved rounds, and neutral at the 8-rank shape: profiles/r06_ab_experiments.txt section 13.)
    if (s2 && (hipEventRecord((hipEvent_t)c->ev[0], st) != hipSuccess || hipStreamWaitEvent(s2, (hipEvent_t)c->ev[0], 0) != hipSuccess))
      return TCAR_E_LAUNCH;
    TcarOpt ox = opt_of(c);
    TcarOpt ob = opt_of(c);
    // (a short shard leaves the 192-row tiles too few workgroups for the chip: 128-row tiles then)
    const int forced = tn(c).bf16_tile;
    // (64-row tiles — 270 workgroups at the 8-rank shape — measured slower than 128-row ones there: 0.697 vs 0.653 ms per step)
    const int tile = (forced == 256 || forced == 128 || forced == 64) ? forced : (((nl + 191) / 192) * 3 < 200 ? 128 : 0);
    RET(tcar_gemm_bf16_de_qz_o(nl, (Bq + 31) & ~31, s->dl16h, nlpad, Bp, anch ? s->aps16h : s->ap16h, g.ldh + g.pt, Bp, g.ldh, c->big, g.ldh,
                               c->mwdhm, c->et_perm, c->tclip, c->qz, tile, s2 ? (void*)s2 : stream, &ob));
    RET(tcar_gemm_bf16_dx_onehot_o(Bq, g.ic, nlpad, s->dl16h, nlpad, Bq, c->e16h, g.ek, nlpad, c->oh16, 160, s->slabs, g.ic + 160,
                                   c->splitk, stream, &ox));
    const int S1 = tcar_gemm_splitk_effective(nlpad, c->splitk);
    TcarOpt orr = opt_of(c);
    if (s2) orr.sig = fork_arm(c, FK_REDUCE);          // (dP leaves write-through when the launch carries the flag)
    // (anchored: E / mwdhm of the SHARD — the residual is non-zero only where the label lives here)
    TcarRowFix fix{s->scale2, s->lab_all, c->E + ((int)g.N == nl ? 0 : (int64_t)s->n0 * g.ek), (long)g.ek, c->mwdhm, nl};
    fix.lab_off = s->n0;
    if (anch) orr.rowfix = &fix;
    RET(tcar_reduce_dact_onehot_o(s->slabs, S1, Bq, g.ic, g.ic + 160, nullptr, 0, nullptr, 0, c->tclip, s->dx, g.ek, c->dP, nullptr,
                                  nullptr, stream, &orr));
    // dP is complete: tcar_shard_finish lets the candidate-side table gradients (aux stream, behind dE) wait for this point — for
    // the launch's own flag when it carries one (a context with flag forks: no event record on this chain, which heads for the dX
    // exchange), else for ev[2]
    if (s2 && !fork_commit(c, FK_REDUCE, orr) && hipEventRecord((hipEvent_t)c->ev[2], st) != hipSuccess) return TCAR_E_LAUNCH;
    return TCAR_OK;
  }
  RET(tcar_softmax_combine(s->world, Bq, stats_all, s->lab_all, s->lse, s->ce, stream));
  RET(tcar_softmax_grad(Bq, nl, s->logits, nlpad, s->lse, s->lab_all, s->n0, s->dl16h, nsb == 1 ? nullptr : s->dl16l, stream));
  float* Gi = c->big;
  float* d_et = c->big + (size_t)nl * g.ldh;
  // dE of the shard on the aux stream beside dX (the caller's stream: dX heads for the reduce-scatter, the critical path)
  hipStream_t st = (hipStream_t)stream, s2 = aux_stream(c);
  if (s2 && (hipEventRecord((hipEvent_t)c->ev[0], st) != hipSuccess || hipStreamWaitEvent(s2, (hipEvent_t)c->ev[0], 0) != hipSuccess))
    return TCAR_E_LAUNCH;
  RET(tcar_gemm_bf16_perm(2, nl, g.ldh + g.pt, (Bq + 31) & ~31, s->dl16h, s->dl16l, nlpad, Bp, s->ap16h, s->ap16l, g.ldh + g.pt, Bp,
                          Gi, g.ldh, d_et, g.pt, g.ldh, c->et_perm, g.ldt, nsb, 1, s2 ? (void*)s2 : stream));
  if (s2 && hipEventRecord((hipEvent_t)c->ev[1], s2) != hipSuccess) return TCAR_E_LAUNCH;
  const int S = tcar_gemm_splitk_effective(nlpad, c->splitk);
  RET(tcar_gemm_bf16(0, Bq, g.ek, nlpad, s->dl16h, s->dl16l, nlpad, Bp, c->e16h, c->e16l, g.ek, nlpad, s->slabs, g.ek, nullptr, 0,
                     0, nsb, c->splitk, stream));
  RET(tcar_splitk_reduce(s->slabs, S, Bq, g.ek, g.ek, s->dx, stream));
  return TCAR_OK;      // dE is still running on the aux stream: tcar_shard_finish queues behind it there, tcar_shard_join joins
}

extern "C" int tcar_shard_finish(const tcar_ctx_t* c, const tcar_shard_t* s, int K, const int32_t* neg_all, const float* coef_all,
                                 void* stream) {
  RET(check_shard(c, s));
  const Geo g(c->d);
  const int nl = s->n_loc;
  tcar_dims_t dc = c->d;
  dc.n_items = nl;
  // Everything here needs dE (aux stream, tcar_shard_backward) and nothing of the dX exchange: it queues behind dE on the aux
  // stream and runs beside the reduce-scatter and the session backward; tcar_shard_join orders the main stream behind it.
  hipStream_t s2 = aux_stream(c);
  void* sf = s2 ? (void*)s2 : stream;
  if (K > 0) {
    if (!neg_all || !coef_all) return TCAR_E_ARG;
    RET(tcar_neg_scatter_range(&c->d, (int64_t)s->world * s->cap, K, s->n0, nl, neg_all, s->att_all, s->ld_att ? s->ld_att : g.ek, coef_all, c->big, sf));
  }
  // the shard's dense item norm, BEFORE any gathered row is scattered in (S5), straight into the item slot of the pieces
  tcar_segments_t one = {};
  one.nseg = 1; one.off[0] = 0; one.len[0] = (int64_t)nl * g.ldh; one.slot[0] = c->slot_item;
  RET(tcar_sqnorm(c->big, &one, c->Gx + c->arena_n, sf));
  tcar_grads_t gr;
  grads_of(c, gr);
  const float* tt[5];
  for (int k = 0; k < 5; ++k) tt[k] = W(c, TCAR_V_MONTH + k);
  if (shard_onehot(c, s, nullptr)) {
    // from the (q, z) pairs of dE (this stream) and dP of the slab reduce (main stream: its flag or ev[2], tcar_shard_backward)
    if (s2) {
      if (fork_live(c, FK_REDUCE)) RET(fork_go(c, FK_REDUCE, (hipStream_t)stream, s2, c->ev[2]));
      else if (hipStreamWaitEvent(s2, (hipEvent_t)c->ev[2], 0) != hipSuccess) return TCAR_E_LAUNCH;
    }
    RET(tcar_cand_time_bwd_onehot_w(&dc, s->world * s->cap, c->inv_off, c->qz, c->dP, s->att_all, s->ld_att ? s->ld_att : g.ek,
                                    c->tclip, c->ct_ws, &gr, sf, TcarWait{}, 1));
  } else {
    RET(tcar_cand_time_bwd_indexed(&dc, tt, c->inv_n, c->inv_off, c->big + (size_t)nl * g.ldh, 1, c->ct_ws, &gr, sf));
  }
  if (s2 && hipEventRecord((hipEvent_t)c->ev[3], s2) != hipSuccess) return TCAR_E_LAUNCH;
  return TCAR_OK;
}

// the main stream behind everything the aux stream still holds of this step (dE, negative rows, shard norm, candidate-time
// backward, weight gradients): before the gathered rows are scattered into the shard's gradient and the arena is exchanged
extern "C" int tcar_shard_join(const tcar_ctx_t* c, void* stream) {
  if (!c) return TCAR_E_ARG;
  if (aux_stream(c) && hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)c->ev[3], 0) != hipSuccess) return TCAR_E_LAUNCH;
  return TCAR_OK;
}

extern "C" int tcar_step_session_backward(const tcar_ctx_t* c, const tcar_batch_t* bt, const float* dx_rows, float* rows_out,
                                          int64_t rows_ld, int64_t rows_total, const float* ce_rows, void* stream) {
  RET(check_ctx(c, bt));
  if (!c->scoring || !dx_rows || !rows_out || (rows_ld && (rows_ld < c->d.ldh || (rows_ld & 3)))) return TCAR_E_ARG;
  const Geo g(c->d);
  const int B = bt->B, T = bt->T, BT = B * T;
  const bool has_neg = bt->K > 0 && bt->neg && c->neg_coef && c->negpart;
  // dattout = (dX + the negative term's part) * tanh'(attout), + the bias gradients of both output transforms
  const bool detc = c->gw_rows != nullptr;       // order-fixed bias / residual-weight gradients (see backward_impl)
  RET(tcar_splitk_reduce_dact(dx_rows, 1, B, g.ek, g.ek, has_neg ? c->negpart : nullptr, g.ic, g.ic, c->attout, g.ek, 2,
                              c->dattout, detc ? nullptr : G(c, TCAR_V_O_B), g.ic, detc ? nullptr : G(c, TCAR_V_OT_B), stream));
  // dpooled = dattout W_o^T in the split form of the fused step (backward_impl): every 128-deep K chunk is its own set of workgroups
  // writing its own slab, the pool backward folds them in slab order while it loads dpooled (round 5: 376 workgroups and one global
  // round trip instead of 104 walking 8 / 5 serial stages — 30 -> 19 us on this chain)
  const int nd_ic = units(g.ic), nd_pt = units(g.pt);
  const int64_t dstride = (int64_t)B * g.ek;
  const bool dsplit = detc && c->proj_slabs && c->proj_slab_floats >= (nd_ic > nd_pt ? nd_ic : nd_pt) * dstride;
  {
    tcar_gemm_desc_t p[2];
    float* dp = dsplit ? c->proj_slabs : c->dpooled;
    p[0] = prob1(B, g.ic, c->dattout, g.ek, W(c, TCAR_V_O_W), g.ic, g.ic, dp, g.ek, nullptr, 0, 0, dsplit ? nd_ic : 1);
    p[1] = prob1(B, g.pt, c->dattout + g.ic, g.ek, W(c, TCAR_V_OT_W), g.pt, g.pt, dp + g.ic, g.ek, nullptr, 0, 0, dsplit ? nd_pt : 1);
    RET(small_gemm(c, 1, 2, p, stream));
  }
  if (detc)
    RET(tcar_attn_pool_bwd_slabs_o(&c->d, B, T, c->x_icp, c->x_pt, c->pre1, c->pre2, c->q, W(c, TCAR_V_M_WRES), W(c, TCAR_V_S_WRES),
                                   c->alpha, dsplit ? c->proj_slabs : c->dpooled, dsplit ? nd_ic : 1, dsplit ? nd_pt : 1, dstride,
                                   c->dx_icp, c->dx_pt, c->dq, c->dpre1, c->dpre2, c->gw_rows, stream, nullptr));
  else
    RET(tcar_attn_pool_bwd_q(&c->d, B, T, c->x_icp, c->x_pt, c->pre1, c->pre2, c->q, W(c, TCAR_V_M_WRES), W(c, TCAR_V_S_WRES),
                             c->alpha, c->dpooled, c->dx_icp, c->dx_pt, c->dq, c->dpre1, c->dpre2, G(c, TCAR_V_M_WRES),
                             G(c, TCAR_V_S_WRES), G(c, TCAR_V_Q2_B), stream));
  {
    tcar_gemm_desc_t p[4];
    p[0] = prob1(B, g.ldh, c->dq, g.ic, W(c, TCAR_V_Q2_W), g.ic, g.ic, c->dq1, g.ldh);
    p[0].dact = 1; p[0].dact_y = c->q1; p[0].ld_dact_y = g.ldh; p[0].colsum = detc ? nullptr : G(c, TCAR_V_Q1_B);
    p[1] = prob1(BT, g.ldh, c->dpre1, g.ldh, W(c, TCAR_V_M_WIN), g.ldh, g.ldh, c->dx_icp, g.ic, nullptr, 0, 1);
    p[2] = prob1(BT, g.ldt, c->dpre1, g.ldh, W(c, TCAR_V_M_WINT), g.ldh, g.ldh, c->dx_act, g.ldt);
    p[3] = prob1(BT, g.pt, c->dpre2, g.ldh, W(c, TCAR_V_S_WIN), g.ldh, g.ldh, c->dx_pt, g.pt, nullptr, 0, 1);
    TcarOpt oi = opt_of(c);
    if (aux_stream(c)) oi.sig = fork_arm(c, FK_INGRAD);      // the weight-gradient fork below (flagged launches store write-through)
    else fork_disarm(c, FK_INGRAD);
    RET(small_gemm(c, 1, 4, p, stream, &oi));
    (void)fork_commit(c, FK_INGRAD, oi);
  }
  {
    // beside the row gradients on the aux stream (behind tcar_shard_finish there); tcar_shard_join covers it.  Forked IN FRONT of
    // the click-query input gradient: the weight gradients need dq1 of the launch above, not dclick (round 4 timeline: the aux
    // chain — weight gradients, column sums — ends the piece; 20 us earlier here is 20 us off the join).  Through the launch's flag
    // where the context has flag forks (round 5: no event record on the main chain, the aux stream starts ~10 us earlier)
    hipStream_t st = (hipStream_t)stream, s2 = aux_stream(c);
    if (s2) RET(fork_go(c, FK_INGRAD, st, s2, c->ev[0]));
    RET(weight_grads(c, g, B, BT, s2 ? (void*)s2 : stream));
    if (detc) RET(det_colsums(c, g, B, s2 ? (void*)s2 : stream));
  }
  if (g.ldh == 256 && g.ldt == 64) {
    // dclick = dq1 Wq1^T as ONE fp32 launch of whole-row dots (query.hip: the layer-1 half of the click-query backward, as in the
    // fused step): 7 us where the 16-workgroup small GEMM walks four serial 64-deep stages (16-20 us) on this chain
    RET(tcar_query_mlp_bwd_o(&c->d, B, nullptr, nullptr, W(c, TCAR_V_Q1_W), nullptr, c->dq1, c->dclick, stream, nullptr));
  } else {
    tcar_gemm_desc_t p = prob1(B, g.ct, c->dq1, g.ldh, W(c, TCAR_V_Q1_W), g.ldh, g.ldh, c->dclick, g.ct);
    RET(small_gemm(c, 1, 1, &p, stream));
  }
  tcar_tables_t tab;
  tcar_grads_t gr;
  tables_of(c, tab);
  grads_of(c, gr);
  // (the small tables keep their LDS + atomic form here: the aux stream of this path already carries the dE tail, the weight
  // gradients and the column sums, and the order-fixed kernel behind them delays the join: 0.69 -> 0.72+ ms on one rank)
  if (aux_stream(c) && hipEventRecord((hipEvent_t)c->ev[3], aux_stream(c)) != hipSuccess) return TCAR_E_LAUNCH;
  gr.rows_out = rows_out;
  gr.rows_ld = rows_ld;
  RET(tcar_gather_clip_bwd(&c->d, &tab, bt, c->dx_icp, c->dx_pt, c->dx_act, c->dclick, &gr, stream));
  if (rows_ld > g.ldh && rows_total >= BT)       // packed exchange rows: the ids behind the rows, loss beside them
    RET(tcar_shard_pack_ids(BT, rows_total, g.ldh, bt->seq, rows_out, rows_ld, B, ce_rows, c->neg_fb, c->neg_weight,
                            (ce_rows && has_neg) ? c->loss : nullptr, stream));
  return TCAR_OK;
}


// dense-weight norms of the arena gradients (after the arena exchange of the catalog-sharded step), summed in a fixed order: several
// workgroups per variable + the order-fixed fold of the last arrival when the context carries the fold scratch (one workgroup per
// variable otherwise) — identical gradients give identical norms on every rank
extern "C" int tcar_step_dense_norms(const tcar_ctx_t* c, void* stream) {
  if (!c || !c->Gx || !c->sqn_dense) return TCAR_E_ARG;
  TcarOpt o = opt_of(c);
  if (c->fold_scratch) { o.scratch = c->fold_scratch; o.scratch_words = c->fold_scratch_words; }
  return tcar_sqnorm_o(c->Gx, &c->segs_dense, c->sqn_dense, stream, &o);
}

// The whole catalog-sharded step of a rank that exchanges nothing (ONE rank without a live process group: every all-gather is a
// view, the reduce-scatter keeps the only slice), sequenced here instead of piece by piece from Python (round 6: nine ctypes calls,
// their argument marshalling and the tensor views between them were ~0.2 ms of host time per step).  Same pieces, same order, same
// streams as ShardExchange.step drives them: begin -> score -> backward -> finish -> session backward -> join -> scatter of the
// packed rows -> dense norms [-> update].  sc = the shard's context (tcar_shard_score / _backward / _finish / tcar_step_update);
// s->att_all / ld_att / head_K must already point at `head`.  lr_update < 0: no update inside the step (the caller defers it).
extern "C" int tcar_shard_step_local(const tcar_ctx_t* c, const tcar_ctx_t* sc, const tcar_shard_t* s, const tcar_batch_t* bt, int Kc,
                                     float* head, int64_t ld_head, int refresh_time, float lr_pending, float* rows, int64_t ldr,
                                     int64_t rows_total, const tcar_dims_t* d_cand, float lr_update, void* stream) {
  if (!c || !sc || !s || !bt || !head || !rows || !d_cand || s->world != 1 || s->att_all != head || s->ld_att != ld_head) return TCAR_E_ARG;
  const int K = (bt->K > 0 && bt->neg) ? bt->K : 0;
  if (s->head_K != K) return TCAR_E_ARG;
  RET(tcar_shard_begin(c, bt, s->cap, Kc, head, ld_head, refresh_time, s->n_loc, lr_pending, stream));
  RET(tcar_shard_score(sc, s, refresh_time, stream));
  RET(tcar_shard_backward(sc, s, s->stats, stream));
  RET(tcar_shard_finish(sc, s, K, K ? s->neg_all : nullptr, K ? s->coef_all : nullptr, stream));
  RET(tcar_step_session_backward(c, bt, s->dx, rows, ldr, rows_total, s->ce, stream));
  RET(tcar_shard_join(c, stream));
  RET(tcar_scatter_add_rows_packed(d_cand, rows, ldr, rows_total, s->n0, c->big, stream));
  RET(tcar_step_dense_norms(c, stream));
  if (lr_update >= 0.f) RET(tcar_step_update(sc, lr_update, stream));
  return TCAR_OK;
}

// Diagnostic (tools/graph_probe.py): capture ONE fused training step (all three streams) into a hipGraph, replay it `iters`
// times and time the replays with HIP events.  The captured step keeps the learning rate it was captured with — the probe
// answers "what would graph replay cost per step", it is not a training path.  Returns the mean ms per replay in *ms_out.
extern "C" int tcar_graph_probe(const tcar_ctx_t* c, const tcar_batch_t* bt, float lr_t, int iters, float* ms_out, void* stream) {
  RET(check_ctx(c, bt));
  if (!ms_out || iters <= 0) return TCAR_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  // a replayed graph would poll the epochs of the captured step (long satisfied): capture the event forks only
  tcar_ctx_t cc = *c;
  TcarTuning no_flags = tn(c);
  no_flags.flag_fork = 0;
  cc.tune = &no_flags;
  c = &cc;
  RET(tcar_train_step(c, bt, 1, lr_t, stream));             // eager once: first-use attribute calls happen outside the capture
  if (hipStreamSynchronize(st) != hipSuccess) return TCAR_E_LAUNCH;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) != hipSuccess) return TCAR_E_LAUNCH;
  const int rc = tcar_train_step(c, bt, 1, lr_t, stream);
  const hipError_t e = hipStreamEndCapture(st, &graph);
  if (rc != TCAR_OK || e != hipSuccess || !graph) return TCAR_E_LAUNCH;
  if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { (void)hipGraphDestroy(graph); return TCAR_E_LAUNCH; }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) (void)hipGraphLaunch(exec, st);
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) (void)hipGraphLaunch(exec, st);
  (void)hipEventRecord(e1, st);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *ms_out = ms / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph);
  return TCAR_OK;
}

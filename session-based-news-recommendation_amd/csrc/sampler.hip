// Device-side batch formation and negative sampling for gfx950.
//
// Reference: Sampler.next_batch (sampler.py:52-113) builds every feed array of a batch in a per-click Python loop and draws
// the negatives with `random.choice` / `np.random.randint` (sampler.py:95-99,118-140).  Here the tensorised session store
// (host/data.py: CSR over clicks + per-click uint8 features) and the negative sources (CSR of `neighbor_dict`: the +-100
// publish-time neighbours of generate_neighbor.py:7-21, or the impression lists of mind_preprocess.py:62-69,85 mapped through
// item_dict) live in HBM, and ONE launch writes the packed int32 feed of a batch
//     seq | month | day | week | hour+1 | minute+1 | dwell bucket | click week | click hour | label | negatives
// from the example indices of the batch — the host only keeps the bucketed shuffle (sampler.py:40-49).  The feed moves
// 4 B per session over PCIe instead of ~150 B.
//
// Negatives follow the reference's RULES with a counter-based generator (splitmix64 of (seed, counter, session, draw)):
//   uniform     K draws from [0, N), with replacement, label not excluded                     sampler.py:98-99
//   neighbour   K draws from the label's list, each redrawn until it differs from the label    sampler.py:133-140
//   impression  at most 21 tries from the session's list; a try counts when the article is a catalog item; the first K
//               hits in try order, then uniform draws fill up                                  sampler.py:118-131
// The stream of numbers is not Python's Mersenne twister: rule equivalence is tested, not draw-for-draw equality (the
// host sampler replays the reference bit for bit when that is wanted).
#include "tcar_common.h"

namespace {

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// draw #`j` of stream (seed, counter, session b): uniform integer in [0, n)
__device__ __forceinline__ unsigned draw(unsigned long long key, unsigned j, unsigned n) {
  const unsigned long long h = mix64(key ^ mix64(0xD1B54A32D192ED03ull * (j + 1)));
  return (unsigned)(((h >> 32) * (unsigned long long)n) >> 32);
}

struct FormArgs {
  tcar_store_t st;
  tcar_negsrc_t src;
  const int32_t* idx;
  int B, T, K, gap_mode, n_items;
  unsigned long long seed, counter;
  int32_t* feed;
};

// thread = one element of the per-click block (7 arrays x B x T), then the per-session block (cw, ch, label)
__global__ __launch_bounds__(256) void form_batch_kernel(const FormArgs a) {
  const long BT = (long)a.B * a.T;
  const long n_click = 7 * BT, n_sess = 3L * a.B;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_click + n_sess; i += (long)gridDim.x * 256) {
    if (i < n_click) {
      const int f = (int)(i / BT);                    // 0 seq, 1..5 publish fields, 6 dwell bucket
      const long r = i - (long)f * BT;
      const int b = (int)(r / a.T), t = (int)(r - (long)b * a.T);
      const long c = a.st.off[a.idx[b]] + t;          // click row of (session b, position t)
      int v;
      if (f == 0) v = a.st.items[c];
      else if (f <= 5) v = a.st.pub[c * 5 + (f - 1)];
      else v = a.gap_mode ? a.st.gap_delta[c] : a.st.gap_active[c];
      a.feed[i] = v;
    } else {
      const long r = i - n_click;
      const int f = (int)(r / a.B), b = (int)(r - (long)f * a.B);
      const long base = a.st.off[a.idx[b]];
      const long last = base + a.T - 1;               // the last INPUT click (sampler.py:86,105-109)
      int v;
      if (f == 0) v = a.st.clk[last * 5 + 2];         // isoweekday - 1
      else if (f == 1) v = a.st.clk[last * 5 + 3];    // hour
      else v = a.st.items[base + a.T] - 1;            // label = last click, 0-based (sampler.py:69)
      a.feed[i] = v;
    }
  }
}

// thread = one session: its K negatives (the impression rule is sequential in the tries; the others ride along)
__global__ __launch_bounds__(256) void sample_neg_kernel(const FormArgs a) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= a.B) return;
  const long BT = (long)a.B * a.T;
  int32_t* out = a.feed + 7 * BT + 3L * a.B + (long)b * a.K;
  const int label = a.feed[7 * BT + 2L * a.B + b];
  const unsigned long long key = mix64(a.seed ^ mix64(a.counter * 0x9E3779B97F4A7C15ull + (unsigned long long)a.idx[b]));
  const unsigned N = (unsigned)a.n_items;
  unsigned j = 0;                                       // draw counter of this session's stream
  if (a.src.mode == 1) {                                // neighbour negatives
    const long lo = a.src.off[label], cnt = a.src.off[label + 1] - lo;
    for (int k = 0; k < a.K; ++k) {
      int pick = label;
      // redraw until the pick differs from the label (a list that holds only the label would never end: bounded)
      for (int tries = 0; tries < 64 && pick == label && cnt > 0; ++tries) pick = a.src.flat[lo + draw(key, j++, (unsigned)cnt)];
      out[k] = (cnt > 0 && pick != label) ? pick : (int)draw(key, j++, N);
    }
  } else if (a.src.mode == 2) {                         // impression negatives
    const int slot = a.src.slot_of_example[a.idx[b]];
    const long lo = a.src.off[slot], cnt = a.src.off[slot + 1] - lo;
    int got = 0;
    for (int tries = 0; tries < 21 && got < a.K && cnt > 0; ++tries) {
      const int pick = a.src.flat[lo + draw(key, j++, (unsigned)cnt)];   // 0-based item id, -1: not a catalog item
      if (pick >= 0) out[got++] = pick;
    }
    for (; got < a.K; ++got) out[got] = (int)draw(key, j++, N);
  } else {
    for (int k = 0; k < a.K; ++k) out[k] = (int)draw(key, j++, N);
  }
}

}  // namespace

extern "C" int tcar_form_batch(const tcar_dims_t* d, const tcar_store_t* st, const tcar_negsrc_t* src, const int32_t* idx, int B,
                               int T, int K, int gap_mode, uint64_t seed, uint64_t counter, int32_t* feed, void* stream) {
  if (B <= 0) return TCAR_OK;
  if (!d || !st || !idx || !feed || T <= 0 || T > TCAR_POS_VOCAB || K < 0 || !st->off || !st->items || !st->pub || !st->clk ||
      !st->gap_active || !st->gap_delta)
    return TCAR_E_ARG;
  FormArgs a{};
  a.st = *st;
  if (src) a.src = *src;
  if (K > 0 && a.src.mode != 0 && (!a.src.off || !a.src.flat || (a.src.mode == 2 && !a.src.slot_of_example))) return TCAR_E_ARG;
  if (a.src.mode < 0 || a.src.mode > 2) return TCAR_E_ARG;
  a.idx = idx; a.B = B; a.T = T; a.K = K; a.gap_mode = gap_mode; a.n_items = d->n_items;
  a.seed = seed; a.counter = counter; a.feed = feed;
  const long n = 7L * B * T + 3L * B;
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  TCAR_LAUNCH(form_batch_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  TCAR_CHECK_LAUNCH();
  if (K > 0) {
    TCAR_LAUNCH(sample_neg_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    TCAR_CHECK_LAUNCH();
  }
  return TCAR_OK;
}

// Host side of the multi-stream beam search (include/caiman_beam.h, part 2).  No device code here:
// the file is a .hip only so that it is built into the same library by the same rule.
//
// Behaviour restated from the reference's Python (training/caiman_asr_train/rnnt/beam.py:285-516,
// hypothesis.py:38-189, serialise_responses.py:28-205, keywords/trie.py:117-203).  Where the reference
// relies on the iteration order of Python dicts (first maximum wins, stable sorts), the containers
// below are insertion-ordered vectors searched linearly -- beams hold a handful of entries.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <thread>
#include <memory>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/caiman_beam.h"
#include "common.h"

namespace caiman {
namespace beam {

constexpr uint64_t kMaxUnicode = 0x10FFFF;
constexpr uint64_t kHashSize = 1000000000039ull;
constexpr uint32_t kSpu = 0x2581;  // sentencepiece word-boundary mark
constexpr int32_t kSosPiece = -1;  // s_seq sentinel of a fresh hypothesis: the text "▁"
constexpr double kInf = std::numeric_limits<double>::infinity();

// ---- text ----------------------------------------------------------------------------------------------
static bool decode_utf8(const std::string& s, std::vector<uint32_t>* out) {
  out->clear();
  for (size_t i = 0; i < s.size();) {
    const unsigned char c = (unsigned char)s[i];
    int n = c < 0x80 ? 1 : (c >> 5) == 0x6 ? 2 : (c >> 4) == 0xE ? 3 : (c >> 3) == 0x1E ? 4 : 0;
    if (n == 0 || i + n > s.size()) return false;
    uint32_t cp = n == 1 ? c : c & (0xFF >> (n + 1));
    for (int j = 1; j < n; ++j) {
      const unsigned char d = (unsigned char)s[i + j];
      if ((d >> 6) != 0x2) return false;
      cp = (cp << 6) | (d & 0x3F);
    }
    out->push_back(cp);
    i += n;
  }
  return true;
}

struct Piece {
  std::string utf8;           // byte order of UTF-8 == code-point order, so string compare matches Python's
  std::vector<uint32_t> cps;  // code points
};

// ---- keyword automaton (keywords/trie.py) ------------------------------------------------------------
struct Keywords {
  std::vector<std::unordered_map<uint32_t, int>> children{1};
  std::vector<double> edge_weight{0.0};
  std::vector<double> committed{0.0};
  std::vector<char> has_committed{0};
  using State = std::vector<std::pair<int, double>>;  // (node, uncommitted score), insertion-ordered

  bool add(const std::vector<uint32_t>& word, double w) {
    if (word.empty()) return false;
    int node = 0;
    for (uint32_t sym : word) {
      auto it = children[node].find(sym);
      int nxt;
      if (it == children[node].end()) {
        nxt = (int)children.size();
        children[node][sym] = nxt;
        children.emplace_back();
        edge_weight.push_back(0.0);
        committed.push_back(0.0);
        has_committed.push_back(0);
      } else {
        nxt = it->second;
      }
      edge_weight[nxt] += w;
      node = nxt;
    }
    if (has_committed[node]) return false;  // duplicate keyword
    has_committed[node] = 1;
    committed[node] = w * (double)word.size();
    return true;
  }

  static State init() { return State{{0, 0.0}}; }

  double step(uint32_t tok, State* st) const {
    State nxt = init();
    double delta = 0.0;
    for (const auto& th : *st) {
      double acc = th.second;
      if (has_committed[th.first]) acc -= committed[th.first];
      auto it = children[th.first].find(tok);
      if (it == children[th.first].end()) {
        delta -= acc;
      } else {
        nxt.emplace_back(it->second, acc + edge_weight[it->second]);
        delta += edge_weight[it->second];
      }
    }
    st->swap(nxt);
    return delta;
  }

  double steps(const std::vector<uint32_t>& toks, State* st) const {
    double total = 0.0;
    for (uint32_t t : toks) total += step(t, st);
    return total;
  }
};

// ---- prediction-state slots --------------------------------------------------------------------------
// Slots are only ever shared between hypotheses of ONE stream, so reference counts need no locking when streams
// are spread over threads; frees are parked in a per-thread list and merged after the parallel section.
static thread_local std::vector<int32_t>* tl_freed = nullptr;

struct SlotPool {
  std::vector<int32_t> refs;
  std::vector<int32_t> free_list;
  int32_t acquire() {
    int32_t s;
    if (!free_list.empty()) {
      s = free_list.back();
      free_list.pop_back();
    } else {
      s = (int32_t)refs.size();
      refs.push_back(0);
    }
    refs[s] = 1;
    return s;
  }
  void retain(int32_t s) {
    if (s >= 0) ++refs[s];
  }
  void release(int32_t s) {
    if (s >= 0 && --refs[s] == 0) (tl_freed ? *tl_freed : free_list).push_back(s);
  }
};

// ---- persistent worker threads: parallel_for over streams ---------------------------------------------------
class Workers {
 public:
  explicit Workers(int n) : n_(n) {
    for (int i = 1; i < n_; ++i) threads_.emplace_back([this, i] { loop(i); });
  }
  ~Workers() {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_ = true;
      ++generation_;
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  int size() const { return n_; }
  // fn(begin, end, worker) over [0, n) in chunks handed out dynamically; the caller is worker 0
  void run(int64_t n, int64_t chunk, const std::function<void(int64_t, int64_t, int)>& fn) {
    if (n_ == 1 || n <= chunk) return fn(0, n, 0);
    {
      std::lock_guard<std::mutex> g(mu_);
      fn_ = &fn;
      total_ = n;
      chunk_ = chunk;
      next_.store(0);
      pending_ = n_ - 1;
      ++generation_;
    }
    cv_.notify_all();
    work(0);
    std::unique_lock<std::mutex> g(mu_);
    done_cv_.wait(g, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  void work(int id) {
    for (;;) {
      const int64_t b = next_.fetch_add(chunk_);
      if (b >= total_) return;
      (*fn_)(b, std::min(total_, b + chunk_), id);
    }
  }
  void loop(int id) {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return generation_ != seen; });
        seen = generation_;
        if (stop_) return;
      }
      work(id);
      std::lock_guard<std::mutex> g(mu_);
      if (--pending_ == 0) done_cv_.notify_one();
    }
  }
  int n_;
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(int64_t, int64_t, int)>* fn_ = nullptr;
  int64_t total_ = 0, chunk_ = 1;
  std::atomic<int64_t> next_{0};
  int pending_ = 0;
  uint64_t generation_ = 0;
  bool stop_ = false;
};

// ---- hypothesis (hypothesis.py:38-162) -----------------------------------------------------------------
// One entry per non-blank token that has not been shipped yet: id, frame, confidence.  Entry 0 is a sentinel (SOS = -1,
// or the last shipped token).  The reference keeps a fourth list with the piece strings; a piece is a function of
// the id (the sentinel -1 stands for the text "▁"), so it is not stored.
struct Tok {
  int32_t y, ts;
  float p;
};

// Short sequences live inside the hypothesis (no heap traffic when a hypothesis is cloned, which happens about five
// times per expansion); finals keep them short.
class Seq {
 public:
  static constexpr uint32_t kInline = 12;
  Seq() = default;
  Seq(const Seq& o) : n_(o.n_) {
    if (o.n_ <= kInline) {
      std::memcpy(inl_, o.data(), o.n_ * sizeof(Tok));
    } else {
      cap_ = o.n_ + kInline;
      heap_ = static_cast<Tok*>(std::malloc(cap_ * sizeof(Tok)));
      std::memcpy(heap_, o.data(), o.n_ * sizeof(Tok));
    }
  }
  Seq& operator=(const Seq&) = delete;
  ~Seq() { std::free(heap_); }
  uint32_t size() const { return n_; }
  const Tok* data() const { return heap_ ? heap_ : inl_; }
  const Tok& operator[](uint32_t i) const { return data()[i]; }
  const Tok& back() const { return data()[n_ - 1]; }
  void push_back(Tok t) {
    if (n_ == cap_) {
      cap_ *= 2;
      Tok* nh = static_cast<Tok*>(std::malloc(cap_ * sizeof(Tok)));
      std::memcpy(nh, data(), n_ * sizeof(Tok));
      std::free(heap_);
      heap_ = nh;
    }
    mut()[n_++] = t;
  }
  void drop_front(uint32_t k) {
    std::memmove(mut(), data() + k, (n_ - k) * sizeof(Tok));
    n_ -= k;
  }

 private:
  Tok* mut() { return heap_ ? heap_ : inl_; }
  Tok inl_[kInline];
  Tok* heap_ = nullptr;
  uint32_t n_ = 0, cap_ = kInline;
};

struct Hyp {
  double score = 0.0;
  Seq seq;
  int32_t y_len_t = 0;
  uint64_t hash = 0;
  int32_t slot = -1;
  bool terminal = false;
  int64_t prev_len = 0;
  Keywords::State kws;  // empty = the start state {0: 0.0} (only materialised when keywords are configured)
  SlotPool* pool;

  explicit Hyp(SlotPool* pl) : pool(pl) {}
  Hyp(const Hyp& o)
      : score(o.score), seq(o.seq), y_len_t(o.y_len_t), hash(o.hash), slot(o.slot), terminal(o.terminal),
        prev_len(o.prev_len), kws(o.kws), pool(o.pool) {
    pool->retain(slot);
  }
  Hyp& operator=(const Hyp&) = delete;
  ~Hyp() { pool->release(slot); }

  // hypotheses are created and destroyed by the million: recycle their blocks per thread
  static std::vector<void*>& block_cache() {
    struct Holder {
      std::vector<void*> v;
      ~Holder() { for (void* p : v) ::operator delete(p); }
    };
    static thread_local Holder h;
    return h.v;
  }
  static void* operator new(size_t sz) {
    auto& c = block_cache();
    if (!c.empty()) {
      void* p = c.back();
      c.pop_back();
      return p;
    }
    return ::operator new(sz);
  }
  static void operator delete(void* p) {
    auto& c = block_cache();
    if (c.size() < 8192) c.push_back(p);
    else ::operator delete(p);
  }

  void set_slot(int32_t ns) {
    pool->retain(ns);
    pool->release(slot);
    slot = ns;
  }
  int64_t len_tot() const { return (int64_t)seq.size() + prev_len; }
  double norm_score() const { return score / (double)len_tot(); }
  void truncate(size_t tkn_idx) {
    const size_t cut = tkn_idx - 1;
    prev_len += (int64_t)cut;
    seq.drop_front((uint32_t)cut);
  }
  void update_hash(const uint32_t* cps, size_t n) {
    for (size_t i = 0; i < n; ++i) hash = (hash * kMaxUnicode + cps[i]) % kHashSize;
  }
};
using HypPtr = std::unique_ptr<Hyp>;

// insertion-ordered map hash -> hypothesis
struct HypSet {
  std::vector<HypPtr> v;
  int find(uint64_t h) const {
    for (size_t i = 0; i < v.size(); ++i)
      if (v[i]->hash == h) return (int)i;
    return -1;
  }
  bool empty() const { return v.empty(); }
  size_t size() const { return v.size(); }
  void clear() { v.clear(); }
  int argmax_score() const {  // first maximum, like max(dict.values(), key=score)
    int best = 0;
    for (size_t i = 1; i < v.size(); ++i)
      if (v[i]->score > v[best]->score) best = (int)i;
    return best;
  }
  HypPtr pop(int i) {
    HypPtr h = std::move(v[i]);
    v.erase(v.begin() + i);
    return h;
  }
};

static double logaddexp(double a, double b) {
  if (a == b) return a + 0.6931471805599453;
  const double d = a - b;
  if (d > 0) return a + std::log1p(std::exp(-d));
  if (d <= 0) return b + std::log1p(std::exp(d));
  return a + b;  // NaN
}

struct Request {
  int32_t stream, frame, y_last, state_in, state_out;
};

}  // namespace beam
}  // namespace caiman

using namespace caiman::beam;

struct caiman_beam {
  caiman_beam_config_t cfg;
  double topk_thresh, score_thresh, final_thresh;
  std::vector<Piece> pieces;
  Piece sos_piece;
  Keywords keywords;
  SlotPool pool;

  struct Stream {
    HypSet kept, open, closed;
    HypPtr cur;
    int32_t cur_out_slot = -1;
    int64_t t = 0;      // frames finished
    int32_t expansions = 0;  // of the open frame
    int64_t avail = 0;  // frames pushed
    int64_t last_final_idx = 0;
    bool frame_open = false;
    bool done = false;
  };
  std::vector<Stream> streams;
  std::vector<int32_t> live;  // streams with an open frame
  std::vector<Request> pending;
  // response records: written per worker thread, gathered into out_i / out_f when the host asks for them
  struct Out {
    std::vector<int32_t> i;
    std::vector<float> f;
    std::vector<int32_t> freed;
  };
  std::vector<Out> outs{1};
  static thread_local Out* tl_out;
  std::vector<int32_t> out_i;
  std::vector<float> out_f;
  std::atomic<bool> saw_unk{false};
  std::atomic<int64_t> capped_frames{0};
  std::unique_ptr<Workers> workers;
  Out& out() { return tl_out ? *tl_out : outs[0]; }

  // ---- text helpers --------------------------------------------------------------------------------
  const Piece& piece(int32_t id) const { return id == kSosPiece ? sos_piece : pieces[id]; }
  int cmp_sseq(const Hyp& a, const Hyp& b) const {  // Python list-of-str comparison
    const size_t n = std::min(a.seq.size(), b.seq.size());
    for (size_t i = 0; i < n; ++i) {
      if (a.seq[i].y == b.seq[i].y) continue;
      const int c = piece(a.seq[i].y).utf8.compare(piece(b.seq[i].y).utf8);
      if (c != 0) return c;
    }
    return a.seq.size() < b.seq.size() ? -1 : a.seq.size() > b.seq.size() ? 1 : 0;
  }
  bool same_piece(int32_t a, int32_t b) const { return a == b || piece(a).utf8 == piece(b).utf8; }

  // ---- set helpers ------------------------------------------------------------------------------------
  HypPtr sos_hyp() {
    HypPtr h(new Hyp(&pool));
    h->seq.push_back({kSosPiece, -1, 1.0f});
    h->y_len_t = 1;
    return h;
  }
  void best_beam_width(HypSet* set) {  // beam.py:661-672
    if ((int)set->size() <= cfg.beam_width) return;
    std::stable_sort(set->v.begin(), set->v.end(), [](const HypPtr& a, const HypPtr& b) { return a->score > b->score; });
    set->v.resize(cfg.beam_width);
  }
  void prune_beam(HypSet* set) {  // beam.py:674-683
    double best = -kInf;
    for (auto& h : set->v) best = std::max(best, h->norm_score());
    const double floor = best - score_thresh;
    auto& v = set->v;
    v.erase(std::remove_if(v.begin(), v.end(), [&](const HypPtr& h) { return !(h->norm_score() >= floor); }), v.end());
  }
  std::vector<const Hyp*> nbest(const HypSet& set) const {  // stable, best normalised score first
    std::vector<const Hyp*> r;
    for (auto& h : set.v) r.push_back(h.get());
    std::stable_sort(r.begin(), r.end(), [](const Hyp* a, const Hyp* b) { return a->norm_score() > b->norm_score(); });
    return r;
  }

  // ---- response records ---------------------------------------------------------------------------------
  void emit_header(int32_t stream, int64_t key, int kind, int64_t start, int64_t dur, int n_alt) {
    auto& o = out().i;
    o.insert(o.end(), {stream, (int32_t)key, kind, (int32_t)start, (int32_t)dur, n_alt});
  }
  // tokens [1, 1 + n) of a hypothesis; `frames` overrides its own frame indices when given
  void emit_alt(const Hyp& h, size_t n, const int32_t* frames = nullptr) {
    Out& o = out();
    o.i.push_back((int32_t)n);
    for (size_t i = 0; i < n; ++i) o.i.push_back(h.seq[1 + i].y);
    for (size_t i = 0; i < n; ++i) o.i.push_back(frames ? frames[i] : h.seq[1 + i].ts);
    for (size_t i = 0; i < n; ++i) o.f.push_back(h.seq[1 + i].p);
  }
  // run fn(stream index in `list`) over a list of streams, on the worker threads when the list is long
  template <typename F>
  void for_streams(int64_t n, F&& fn) {
    if (!workers || n < 128) {
      for (int64_t i = 0; i < n; ++i) fn(i);
      return;
    }
    workers->run(n, 32, [&](int64_t b, int64_t e, int w) {
      tl_out = &outs[w];
      tl_freed = &outs[w].freed;
      for (int64_t i = b; i < e; ++i) fn(i);
      tl_out = nullptr;
      tl_freed = nullptr;
    });
    for (auto& o : outs) {
      pool.free_list.insert(pool.free_list.end(), o.freed.begin(), o.freed.end());
      o.freed.clear();
    }
  }
  // final built from hyps that share s_seq[1:tkn_idx] (serialise_responses.py:150-205); hyps[0] gives the
  // ids and confidences, the frame of each token is the earliest any hypothesis saw it
  void emit_final(int32_t stream, int64_t key, const std::vector<const Hyp*>& hyps, size_t tkn_idx) {
    const Hyp& head = *hyps[0];
    const size_t n = tkn_idx - 1;
    std::vector<int32_t> frames(n);
    for (size_t i = 0; i < n; ++i) {
      int32_t m = head.seq[1 + i].ts;
      for (const Hyp* h : hyps) m = std::min(m, h->seq[1 + i].ts);
      frames[i] = m;
    }
    const int32_t lo = *std::min_element(frames.begin(), frames.end());
    const int32_t hi = *std::max_element(frames.begin(), frames.end());
    emit_header(stream, key, 0, lo, hi - lo + 1, 1);
    emit_alt(head, n, frames.data());
  }
  // -> true if a final was shipped (and the shared prefix cut off every hypothesis)
  bool get_final(int32_t stream, int64_t key, HypSet* kept) {
    std::vector<const Hyp*> sorted;
    for (auto& h : kept->v) sorted.push_back(h.get());
    std::stable_sort(sorted.begin(), sorted.end(), [&](const Hyp* a, const Hyp* b) { return cmp_sseq(*a, *b) < 0; });
    const Hyp &first = *sorted.front(), &last = *sorted.back();
    const size_t lim = std::min(first.seq.size(), last.seq.size());
    size_t k = 1;
    while (k < lim && same_piece(first.seq[k].y, last.seq[k].y)) ++k;
    if (k == 1) return false;
    emit_final(stream, key, sorted, k);
    for (auto& h : kept->v) h->truncate(k);
    return true;
  }
  void emit_partials(int32_t stream, int64_t key, const HypSet& kept) {
    auto order = nbest(kept);
    int64_t start = key;
    int n_alt = 0;
    for (const Hyp* h : order)
      if (h->seq.size() > 1) {
        ++n_alt;
        for (uint32_t i = 1; i < h->seq.size(); ++i) start = std::min<int64_t>(start, h->seq[i].ts);
      }
    emit_header(stream, key, 1, start, key - start + 1, n_alt);
    for (const Hyp* h : order)
      if (h->seq.size() > 1) emit_alt(*h, h->seq.size() - 1);
  }
  void last_frame_response(int32_t stream, int64_t key, const HypSet& kept) {  // serialise_responses.py:58-75
    const Hyp* best = nbest(kept)[0];
    if (best->seq.size() > 1)
      emit_final(stream, key, {best}, best->seq.size());
    else
      emit_header(stream, key, 2, key, 0, 0);
  }

  // ---- per-stream state machine (beam.py:285-415) ---------------------------------------------------------
  void finish(int32_t si, int64_t key) {
    Stream& s = streams[si];
    last_frame_response(si, key, s.kept);
    s.done = true;
    s.frame_open = false;
    s.kept.clear();
    s.open.clear();
    s.closed.clear();
  }
  // starts the expansion of frame s.t (does not touch the live list: safe on a worker thread)
  void begin_frame(int32_t si) {
    Stream& s = streams[si];
    if (cfg.max_symbol_per_sample >= 0) {
      const Hyp& best = *s.kept.v[s.kept.argmax_score()];
      if (best.len_tot() > cfg.max_symbol_per_sample) return finish(si, s.t + 1);
    }
    for (auto& h : s.kept.v) h->y_len_t = 0;
    s.open.v = std::move(s.kept.v);
    s.kept.v.clear();
    s.closed.clear();
    s.frame_open = true;
    s.expansions = 0;
  }
  bool silence_terminate(const HypSet& kept, int64_t idx) const {  // beam.py:266-283
    if (cfg.eos_vad_threshold == kInf) return false;
    int32_t last = std::numeric_limits<int32_t>::min();
    for (auto& h : kept.v) last = std::max(last, h->seq.back().ts);
    if (last < 0) return false;
    return (double)(idx - last) * cfg.frame_width >= cfg.eos_vad_threshold;
  }
  void close_frame(int32_t si) {
    Stream& s = streams[si];
    const int64_t t = s.t;
    s.frame_open = false;
    s.kept.v = std::move(s.closed.v);
    s.closed.v.clear();
    prune_beam(&s.kept);
    if (s.kept.v[s.kept.argmax_score()]->terminal) {
      last_frame_response(si, t, s.kept);
      s.done = true;
      s.kept.clear();
      return;
    }
    const double since_final = (double)(t - s.last_final_idx) * cfg.frame_width;
    // a frame may retry the final after dropping hypotheses; only the last attempt's record is kept
    while (true) {
      Out& o = out();
      const size_t mark_i = o.i.size(), mark_f = o.f.size();
      const bool shipped = get_final(si, t, &s.kept);
      if (cfg.return_partials) emit_partials(si, t, s.kept);
      if (!shipped && !cfg.return_partials) emit_header(si, t, 2, t, 0, 0);
      if (s.kept.size() <= 1) {
        s.last_final_idx = t;
        break;
      }
      if (shipped) {
        int32_t m = std::numeric_limits<int32_t>::max();
        for (auto& h : s.kept.v) m = std::min(m, h->seq[0].ts);
        s.last_final_idx = m;
        break;
      }
      if (since_final <= final_thresh) break;
      // overdue: drop the weakest hypothesis and try again (beam.py:345-348)
      int worst = 0;
      for (size_t i = 1; i < s.kept.size(); ++i)
        if (s.kept.v[i]->norm_score() < s.kept.v[worst]->norm_score()) worst = (int)i;
      s.kept.pop(worst);
      o.i.resize(mark_i);
      o.f.resize(mark_f);
    }
    if (silence_terminate(s.kept, t)) return finish(si, t + 1);
    s.t += 1;
    if (s.t < s.avail) begin_frame(si);  // a queued frame: the stream stays in the live list
  }

  void update_hyps(Stream& s, float logp_f, int32_t tok, int64_t time_idx, int32_t out_slot) {  // beam.py:449-516
    const Hyp& parent = *s.cur;
    const double logp = (double)logp_f;
    if (tok == cfg.blank_idx) {
      const int i = s.closed.find(parent.hash);
      if (i >= 0) {
        s.closed.v[i]->score = logaddexp(s.closed.v[i]->score, parent.score + logp);
      } else {
        HypPtr h(new Hyp(parent));
        h->score += logp;
        s.closed.v.push_back(std::move(h));
      }
      return;
    }
    if (tok == 0) {  // id 0 is <unk>: it has no text to hash or score (beam.py:621,635)
      saw_unk.store(true);
      return;
    }
    HypPtr h(new Hyp(parent));
    h->score += logp;
    const int32_t prev_tok = h->seq.back().y;
    h->seq.push_back({tok, (int32_t)time_idx, std::exp(logp_f)});
    h->set_slot(out_slot);
    h->y_len_t += 1;
    if (cfg.eos_terminal_idx >= 0 && tok == cfg.eos_terminal_idx) h->terminal = true;
    const Piece& pc = pieces[tok];
    if (keywords.children.size() > 1) {
      if (h->kws.empty()) h->kws = Keywords::init();
      h->score += keywords.steps(pc.cps, &h->kws);
    }
    // a word-boundary mark right after a word-boundary mark adds nothing to the text (beam.py:644-659)
    const Piece& prev = piece(prev_tok);
    const bool doubled = !prev.cps.empty() && prev.cps.back() == kSpu && pc.cps[0] == kSpu;
    h->update_hash(pc.cps.data() + (doubled ? 1 : 0), pc.cps.size() - (doubled ? 1 : 0));
    const int i = s.open.find(h->hash);
    if (i < 0) {
      s.open.v.push_back(std::move(h));
    } else {
      const double merged = logaddexp(s.open.v[i]->score, h->score);
      if (h->score > s.open.v[i]->score) s.open.v[i] = std::move(h);  // keep the likelier tokenisation
      s.open.v[i]->score = merged;
    }
  }

  void feed_one(int32_t si, int32_t k, const float* sc, const int32_t* tk, float blank_logp) {
    Stream& s = streams[si];
    const Hyp& cur = *s.cur;
    const bool add_ys = cfg.max_symbols_per_step <= 0 || cur.y_len_t < cfg.max_symbols_per_step;
    if (add_ys) {
      float smax = sc[0];
      for (int j = 1; j < k; ++j) smax = std::max(smax, sc[j]);
      const float floor = topk_thresh == kInf ? -std::numeric_limits<float>::infinity() : smax - (float)topk_thresh;
      bool saw_blank = false;
      for (int j = 0; j < k; ++j) {
        if (!(sc[j] >= floor)) continue;
        saw_blank |= tk[j] == cfg.blank_idx;
        update_hyps(s, sc[j], tk[j], s.t, s.cur_out_slot);
      }
      if (!saw_blank) update_hyps(s, blank_logp, cfg.blank_idx, s.t, s.cur_out_slot);
    } else {
      update_hyps(s, blank_logp, cfg.blank_idx, s.t, s.cur_out_slot);
    }
    s.cur.reset();
    pool.release(s.cur_out_slot);  // the children hold their own references now
    s.cur_out_slot = -1;
    s.expansions += 1;
    const bool capped = cfg.max_expansions_per_frame > 0 && s.expansions >= cfg.max_expansions_per_frame && !s.open.empty();
    if (capped) capped_frames.fetch_add(1);  // serving safeguard: settle the frame with what has been closed so far
    if (!s.open.empty() && !capped) {
      const double bar = s.open.v[s.open.argmax_score()]->score;
      size_t ahead = 0;
      for (auto& h : s.closed.v) ahead += h->score > bar;
      if ((int)ahead < cfg.beam_width) return;  // keep expanding this frame
      auto& v = s.closed.v;
      v.erase(std::remove_if(v.begin(), v.end(), [&](const HypPtr& h) { return !(h->score > bar); }), v.end());
    }
    best_beam_width(&s.closed);
    s.open.clear();
    close_frame(si);
  }
};

thread_local caiman_beam::Out* caiman_beam::tl_out = nullptr;

// ---- C entry points --------------------------------------------------------------------------------------
#define BEAM_CHECK(cond, ...)             \
  do {                                    \
    if (!(cond)) {                        \
      ::caiman::set_error(__VA_ARGS__);   \
      return CAIMAN_ERR_INVALID;          \
    }                                     \
  } while (0)

extern "C" caiman_beam_t* caiman_beam_create(const caiman_beam_config_t* cfg, int32_t n_streams,
                                             const char* const* pieces, int32_t n_pieces,
                                             const char* const* keywords, const double* keyword_weights,
                                             int32_t n_keywords) {
  using caiman::set_error;
  if (!cfg || n_streams < 1 || !pieces || n_pieces < 1) {
    set_error("beam_create: null config / pieces or n_streams < 1");
    return nullptr;
  }
  if (cfg->beam_width < 1 || cfg->blank_idx < 0 || cfg->blank_idx > n_pieces) {
    set_error("beam_create: beam_width must be > 0 and blank_idx within [0, n_pieces]");
    return nullptr;
  }
  std::unique_ptr<caiman_beam> h(new caiman_beam);
  h->cfg = *cfg;
  h->topk_thresh = cfg->beam_prune_topk_thresh < 0 ? kInf : cfg->beam_prune_topk_thresh;
  h->score_thresh = cfg->beam_prune_score_thresh < 0 ? kInf : cfg->beam_prune_score_thresh;
  h->final_thresh = cfg->final_emission_thresh < 0 ? kInf : cfg->final_emission_thresh;
  if (h->cfg.eos_vad_threshold < 0) h->cfg.eos_vad_threshold = kInf;
  if (h->topk_thresh <= 1e-9 || h->score_thresh <= 1e-9) {
    set_error("beam_create: a prune threshold of 0 keeps only the most probable entry; use the greedy decoder");
    return nullptr;
  }
  if ((h->cfg.eos_vad_threshold != kInf || h->final_thresh != kInf) && !(cfg->frame_width > 0.0)) {
    set_error("beam_create: frame_width > 0 is required with eos_vad_threshold / final_emission_thresh");
    return nullptr;
  }
  h->pieces.resize(n_pieces);
  for (int32_t i = 0; i < n_pieces; ++i) {
    h->pieces[i].utf8 = pieces[i] ? pieces[i] : "";
    if (!decode_utf8(h->pieces[i].utf8, &h->pieces[i].cps) || h->pieces[i].cps.empty()) {
      if (i == cfg->blank_idx) continue;  // the blank has no text
      set_error("beam_create: piece %d is empty or not valid UTF-8", i);
      return nullptr;
    }
  }
  h->sos_piece.utf8 = "\xE2\x96\x81";
  h->sos_piece.cps = {kSpu};
  for (int32_t i = 0; i < n_keywords; ++i) {
    std::vector<uint32_t> cps;
    if (!keywords || !keyword_weights || !keywords[i] || !decode_utf8(keywords[i], &cps) || !h->keywords.add(cps, keyword_weights[i])) {
      set_error("beam_create: keyword %d is empty, duplicated or not valid UTF-8", i);
      return nullptr;
    }
  }
  h->streams.resize(n_streams);
  for (int32_t s = 0; s < n_streams; ++s) h->streams[s].kept.v.push_back(h->sos_hyp());
  // streams are independent: long request lists are spread over worker threads (CAIMAN_BEAM_THREADS, default
  // min(hardware threads, 16); 1 = everything on the calling thread)
  int nt = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char* e = std::getenv("CAIMAN_BEAM_THREADS")) nt = std::max(1, std::atoi(e));
  if (nt > 1 && n_streams >= 128) {
    h->workers.reset(new Workers(nt));
    h->outs.resize(nt);
  }
  return h.release();
}

extern "C" void caiman_beam_destroy(caiman_beam_t* h) { delete h; }

extern "C" int caiman_beam_reset_stream(caiman_beam_t* h, int32_t stream) {
  BEAM_CHECK(h && stream >= 0 && stream < (int32_t)h->streams.size(), "beam_reset_stream: bad handle / stream %d", stream);
  BEAM_CHECK(h->pending.empty(), "beam_reset_stream: requests are outstanding");
  auto& s = h->streams[stream];
  h->live.erase(std::remove(h->live.begin(), h->live.end(), stream), h->live.end());
  s = caiman_beam::Stream();
  s.kept.v.push_back(h->sos_hyp());
  return CAIMAN_OK;
}

extern "C" int caiman_beam_push_frame(caiman_beam_t* h, const int32_t* streams, int32_t n) {
  BEAM_CHECK(h && (streams || n == 0) && n >= 0, "beam_push_frame: null argument");
  BEAM_CHECK(h->pending.empty(), "beam_push_frame: requests are outstanding; feed them first");
  for (int32_t i = 0; i < n; ++i) {
    const int32_t si = streams[i];
    BEAM_CHECK(si >= 0 && si < (int32_t)h->streams.size(), "beam_push_frame: stream %d out of range", si);
  }
  auto& live = h->live;
  live.erase(std::remove_if(live.begin(), live.end(), [&](int32_t si) { return !h->streams[si].frame_open; }), live.end());
  for (int32_t i = 0; i < n; ++i) {
    auto& s = h->streams[streams[i]];
    if (s.done) continue;
    s.avail += 1;
    if (!s.frame_open) {  // idle: start on the new frame now; otherwise it waits its turn
      h->begin_frame(streams[i]);
      if (s.frame_open) live.push_back(streams[i]);
    }
  }
  return CAIMAN_OK;
}

extern "C" int64_t caiman_beam_requests(caiman_beam_t* h, int32_t* stream, int32_t* frame, int32_t* y_last,
                                        int32_t* state_in, int32_t* state_out, int64_t cap) {
  if (!h || !stream || !frame || !y_last || !state_in || !state_out) {
    caiman::set_error("beam_requests: null argument");
    return -1;
  }
  if (!h->pending.empty()) {
    caiman::set_error("beam_requests: the previous requests have not been fed");
    return -1;
  }
  // streams whose frame closed since the last round drop out of the live list
  auto& live = h->live;
  live.erase(std::remove_if(live.begin(), live.end(), [&](int32_t si) { return !h->streams[si].frame_open; }), live.end());
  if ((int64_t)live.size() > cap) {
    caiman::set_error("beam_requests: %lld requests pending but room for %lld", (long long)live.size(), (long long)cap);
    return -1;
  }
  h->for_streams((int64_t)live.size(), [&](int64_t i) {
    auto& s = h->streams[live[i]];
    s.cur = s.open.pop(s.open.argmax_score());
  });
  for (int32_t si : live) {
    auto& s = h->streams[si];
    s.cur_out_slot = h->pool.acquire();
    h->pending.push_back({si, (int32_t)s.t, s.cur->seq.back().y, s.cur->slot, s.cur_out_slot});
  }
  for (size_t i = 0; i < h->pending.size(); ++i) {
    const Request& r = h->pending[i];
    stream[i] = r.stream;
    frame[i] = r.frame;
    y_last[i] = r.y_last;
    state_in[i] = r.state_in;
    state_out[i] = r.state_out;
  }
  return (int64_t)h->pending.size();
}

extern "C" int caiman_beam_feed(caiman_beam_t* h, int64_t n, int32_t k, const float* top_scores,
                                const int32_t* top_tokens, const float* blank_logp) {
  BEAM_CHECK(h && (n == 0 || (top_scores && top_tokens && blank_logp)), "beam_feed: null argument");
  BEAM_CHECK(n == (int64_t)h->pending.size(), "beam_feed: %lld answers for %lld requests", (long long)n,
             (long long)h->pending.size());
  BEAM_CHECK(k >= 1, "beam_feed: k must be >= 1");
  const int32_t n_tok = (int32_t)h->pieces.size();
  for (int64_t i = 0; i < n; ++i)
    for (int32_t j = 0; j < k; ++j) {
      const int32_t t = top_tokens[i * k + j];
      BEAM_CHECK(t >= 0 && (t < n_tok || t == h->cfg.blank_idx), "beam_feed: token id %d out of range", t);
    }
  std::vector<Request> reqs;
  reqs.swap(h->pending);
  h->for_streams(n, [&](int64_t i) {
    h->feed_one(reqs[i].stream, k, top_scores + i * k, top_tokens + i * k, blank_logp[i]);
  });
  BEAM_CHECK(!h->saw_unk.load(), "Decoding error: '<unk>' token encountered");
  return CAIMAN_OK;
}

extern "C" int caiman_beam_close_stream(caiman_beam_t* h, int32_t stream) {
  BEAM_CHECK(h && stream >= 0 && stream < (int32_t)h->streams.size(), "beam_close_stream: bad handle / stream %d", stream);
  auto& s = h->streams[stream];
  BEAM_CHECK(!s.frame_open, "beam_close_stream: stream %d still has frames to expand", stream);
  if (!s.done) h->finish(stream, s.t);
  return CAIMAN_OK;
}

extern "C" int caiman_beam_stream_done(const caiman_beam_t* h, int32_t stream) {
  return h && stream >= 0 && stream < (int32_t)h->streams.size() && h->streams[stream].done ? 1 : 0;
}

extern "C" int64_t caiman_beam_backlog(const caiman_beam_t* h, int32_t stream) {
  if (!h) return 0;
  auto lag = [](const caiman_beam::Stream& s) { return s.done ? (int64_t)0 : s.avail - s.t; };
  if (stream >= 0) return stream < (int32_t)h->streams.size() ? lag(h->streams[stream]) : 0;
  int64_t m = 0;
  for (auto& s : h->streams) m = std::max(m, lag(s));
  return m;
}

extern "C" int64_t caiman_beam_capped_frames(const caiman_beam_t* h) { return h ? h->capped_frames.load() : 0; }

extern "C" int64_t caiman_beam_state_slots(const caiman_beam_t* h) { return h ? (int64_t)h->pool.refs.size() : 0; }

extern "C" int caiman_beam_responses(caiman_beam_t* h, const int32_t** ints, int64_t* n_ints, const float** floats,
                                     int64_t* n_floats) {
  BEAM_CHECK(h && ints && n_ints && floats && n_floats, "beam_responses: null argument");
  for (auto& o : h->outs) {
    h->out_i.insert(h->out_i.end(), o.i.begin(), o.i.end());
    h->out_f.insert(h->out_f.end(), o.f.begin(), o.f.end());
    o.i.clear();
    o.f.clear();
  }
  *ints = h->out_i.data();
  *n_ints = (int64_t)h->out_i.size();
  *floats = h->out_f.data();
  *n_floats = (int64_t)h->out_f.size();
  return CAIMAN_OK;
}

extern "C" void caiman_beam_clear_responses(caiman_beam_t* h) {
  if (h) {
    h->out_i.clear();
    h->out_f.clear();
  }
}

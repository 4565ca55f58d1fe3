// numpy's LEGACY standard-normal stream, bit for bit, several times faster: the reference draws u ~ N(0,1)[B,S,P] with
// np.random.randn on the host every step (vihds/vae.py:22-24: 252 000 normals at B=36, S=200, P=35 = 2.3 ms of numpy on the
// GPU box, 4-6 ms in the build container -- 25x the GPU's work for the step).  A spec that keeps the reference's RNG stream
// (u_rng: numpy, the default) gets the SAME numbers from here:
//
//   * MT19937 exactly as numpy's RandomState holds it (key[624], pos): the 624-word regeneration is three dependency-free
//     runs (0..226, 227..453, 454..622, then the wrap-around word) that the compiler vectorises; the raw state of every
//     block is kept, so the generator's state after any number of consumed words is at hand;
//   * legacy_double = (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53, legacy_gauss = Marsaglia's polar method with its rejection
//     loop and the cached second deviate (numpy/random/src/legacy/legacy-distributions.c): an ATTEMPT consumes four words
//     whatever its outcome, so attempt a reads words 4a .. 4a+3 and all attempts are independent -- evaluated by a small
//     persistent pool of threads, accepted ones compacted in order (prefix sum over per-chunk counts); accepted attempt m
//     yields outputs 2m (= f x2) and 2m + 1 (= f x1), f = sqrt(-2 log(r2) / r2) with libm's log, as numpy calls it;
//   * the state handed back is numpy's state after the same call (key, pos, has_gauss, cached deviate).
//
// C ABI (ctypes: vi-hds_amd/vihds/nprand.py).  Host code only; not part of the HIP library.
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace {

constexpr int N = 624, M = 397;
constexpr uint32_t MATRIX_A = 0x9908b0dfU, UPPER = 0x80000000U, LOWER = 0x7fffffffU;

// next block of raw state words from the previous one (numpy mt19937_gen)
inline void regenerate(const uint32_t* in, uint32_t* out) {
  for (int i = 0; i < N - M; ++i) {
    const uint32_t y = (in[i] & UPPER) | (in[i + 1] & LOWER);
    out[i] = in[i + M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
  }
  for (int i = N - M; i < 2 * (N - M); ++i) {
    const uint32_t y = (in[i] & UPPER) | (in[i + 1] & LOWER);
    out[i] = out[i - (N - M)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
  }
  for (int i = 2 * (N - M); i < N - 1; ++i) {
    const uint32_t y = (in[i] & UPPER) | (in[i + 1] & LOWER);
    out[i] = out[i - (N - M)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
  }
  const uint32_t y = (in[N - 1] & UPPER) | (out[0] & LOWER);
  out[N - 1] = out[M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
}
inline uint32_t temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680U;
  y ^= (y << 15) & 0xefc60000U;
  y ^= (y >> 18);
  return y;
}

// a small persistent pool: run(f, k) calls f(0) .. f(k-1), f(0) on the calling thread
class Pool {
 public:
  explicit Pool(int workers) {
    for (int t = 0; t < workers; ++t) threads_.emplace_back([this, t] { loop(t + 1); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
      ++gen_;
    }
    cv_.notify_all();
    for (auto& th : threads_) th.join();
  }
  int size() const { return (int)threads_.size() + 1; }
  void run(const std::function<void(int)>& f, int k) {
    if (k <= 1) { f(0); return; }
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &f;
      parts_ = k;
      pending_.store(k - 1);
      ++gen_;
    }
    cv_.notify_all();
    f(0);
    while (pending_.load(std::memory_order_acquire) > 0) std::this_thread::yield();
  }

 private:
  void loop(int id) {
    unsigned long seen = 0;
    for (;;) {
      const std::function<void(int)>* f;
      int parts;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        f = fn_;
        parts = parts_;
      }
      if (id < parts) {
        (*f)(id);
        pending_.fetch_sub(1, std::memory_order_release);
      }
    }
  }
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_;
  const std::function<void(int)>* fn_ = nullptr;
  int parts_ = 0;
  std::atomic<int> pending_{0};
  unsigned long gen_ = 0;
  bool stop_ = false;
};

constexpr long long CH = 2048;  // attempts per chunk

struct Buffers {  // kept between calls: fresh allocations of this size cost more in page faults than the arithmetic
  std::vector<uint32_t> raw;       // raw generator states, block after block; block 0 = the caller's key
  std::vector<long long> count;    // accepted attempts per chunk
  std::vector<float> vals;         // per chunk: its accepted attempts' outputs in order (f x2, f x1), 2 CH floats
};

std::mutex g_lock;           // one call at a time (the pool and the buffers are shared)
Pool* g_pool = nullptr;
Buffers g_buf;

inline void cpu_relax() {
#if defined(__x86_64__)
  __builtin_ia32_pause();
#endif
}

// the attempts a0 .. a0 + CH of the stream whose word w is temper(raw[pos0 + w]): r2, x1, x2 (no branch, no call: vectorised)
inline void chunk_attempts(const uint32_t* raw, long long pos0, long long a0, double* X1, double* X2, double* R2) {
  const uint32_t* w = raw + pos0 + 4 * a0;
  for (long long a = 0; a < CH; ++a) {
    const uint32_t w0 = temper(w[4 * a]), w1 = temper(w[4 * a + 1]), w2 = temper(w[4 * a + 2]), w3 = temper(w[4 * a + 3]);
    const double d1 = ((double)(w0 >> 5) * 67108864.0 + (double)(w1 >> 6)) / 9007199254740992.0;
    const double d2 = ((double)(w2 >> 5) * 67108864.0 + (double)(w3 >> 6)) / 9007199254740992.0;
    const double x1 = 2.0 * d1 - 1.0, x2 = 2.0 * d2 - 1.0;
    X1[a] = x1;
    X2[a] = x2;
    R2[a] = x1 * x1 + x2 * x2;
  }
}

}  // namespace

extern "C" {

// out[0..n) <- what np.random.standard_normal(n).astype(np.float32) would return from the RandomState (key, pos, has_gauss,
// gauss); the four are updated to the state numpy would be left in.  Returns 0, or -1 on bad arguments.
//
// One thread regenerates the generator's blocks in order and publishes how far it is; the others (and then that one too)
// take chunks of 2048 attempts as soon as the blocks under them exist, evaluate them (a vectorised pass for x1, x2, r2, then
// the accepted ones' sqrt(-2 log r2 / r2)) into the chunk's own buffer; a last pass copies the chunks' outputs to their
// places (prefix sum over the chunks' counts).
int vihds_np_randn_f32(uint32_t* key, int* pos, int* has_gauss, double* gauss, float* out, long long n, int n_threads) {
  if (!key || !pos || !has_gauss || !gauss || !out || n < 0 || *pos < 0 || *pos > N) return -1;
  std::lock_guard<std::mutex> guard(g_lock);
  long long done = 0;
  if (n > 0 && *has_gauss) {
    out[done++] = (float)*gauss;
    *has_gauss = 0;
    *gauss = 0.0;
  }
  const long long need_pairs = (n - done + 1) / 2;  // accepted attempts still to find
  if (need_pairs == 0) return 0;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 64) n_threads = 64;
  if (!g_pool || g_pool->size() < n_threads) {
    delete g_pool;
    g_pool = new Pool(n_threads - 1);
  }
  const int K = need_pairs < 4096 ? 1 : n_threads;
  std::vector<uint32_t>& raw = g_buf.raw;
  std::vector<long long>& count = g_buf.count;
  std::vector<float>& vals = g_buf.vals;
  const long long pos0 = *pos;
  raw.resize(N);
  std::memcpy(raw.data(), key, N * sizeof(uint32_t));
  count.clear();
  long long pairs_found = 0, chunks_done = 0, blocks_done = 1;
  while (pairs_found < need_pairs) {
    const long long want_attempts = chunks_done * CH + (long long)((need_pairs - pairs_found) * 1.2732395447 * 1.02) + 64;
    const long long want_chunks = (want_attempts + CH - 1) / CH;
    const long long need_blocks = ((long long)pos0 + 4 * want_chunks * CH + N - 1) / N;
    raw.resize((size_t)std::max<long long>(need_blocks, blocks_done) * N);
    count.resize((size_t)want_chunks);
    vals.resize((size_t)want_chunks * 2 * CH);
    std::atomic<long long> blocks_ready(blocks_done), next_chunk(chunks_done);
    uint32_t* R = raw.data();
    long long* C = count.data();
    float* V = vals.data();
    g_pool->run([&](int t) {
      if (t == 0) {
        for (long long b = blocks_done; b < need_blocks; ++b) {
          regenerate(R + (size_t)(b - 1) * N, R + (size_t)b * N);
          blocks_ready.store(b + 1, std::memory_order_release);
        }
      }
      double X1[CH], X2[CH], R2[CH];
      for (;;) {
        const long long c = next_chunk.fetch_add(1, std::memory_order_relaxed);
        if (c >= want_chunks) break;
        const long long blk = (pos0 + 4 * (c + 1) * CH + N - 1) / N;
        for (int spins = 0; blocks_ready.load(std::memory_order_acquire) < blk; ++spins) {
          if (spins < 256) cpu_relax();
          else std::this_thread::yield();  // (an oversubscribed host: let the regenerating thread run)
        }
        chunk_attempts(R, pos0, c * CH, X1, X2, R2);
        float* v = V + (size_t)c * 2 * CH;
        long long m = 0;
        for (long long a = 0; a < CH; ++a) {
          const double r2 = R2[a];
          if (r2 >= 1.0 || r2 == 0.0) continue;
          const double f = std::sqrt(-2.0 * std::log(r2) / r2);
          v[2 * m] = (float)(f * X2[a]);
          v[2 * m + 1] = (float)(f * X1[a]);
          ++m;
        }
        C[c] = m;
      }
    }, K);
    for (long long c = chunks_done; c < want_chunks; ++c) pairs_found += count[(size_t)c];
    chunks_done = want_chunks;
    blocks_done = std::max<long long>(need_blocks, blocks_done);
  }
  // where every chunk's outputs go; the chunk in which the last needed pair falls
  std::vector<long long> start((size_t)chunks_done + 1, 0);
  long long last_chunk = 0;
  for (long long c = 0; c < chunks_done; ++c) {
    start[(size_t)c + 1] = start[(size_t)c] + count[(size_t)c];
    if (start[(size_t)c] < need_pairs) last_chunk = c;
  }
  {
    const float* V = vals.data();
    g_pool->run([&](int t) {
      for (long long c = t; c <= last_chunk; c += K) {
        const long long m0 = start[(size_t)c], m1 = std::min<long long>(start[(size_t)c + 1], need_pairs);
        const long long o = done + 2 * m0;
        long long cnt = 2 * (m1 - m0);
        if (o + cnt > n) cnt = n - o;  // (an odd request: the last pair's second deviate stays cached)
        if (cnt > 0) std::memcpy(out + o, V + (size_t)c * 2 * CH, (size_t)cnt * sizeof(float));
      }
    }, (int)std::min<long long>(K, last_chunk + 1));
  }
  // the attempt that yielded the last needed pair: the (need_pairs - start[last_chunk])-th accepted one of its chunk
  long long last_attempt = -1;
  double last_gauss = 0.0;
  {
    double X1[CH], X2[CH], R2[CH];
    chunk_attempts(raw.data(), pos0, last_chunk * CH, X1, X2, R2);
    long long m = start[(size_t)last_chunk];
    for (long long a = 0; a < CH; ++a) {
      const double r2 = R2[a];
      if (r2 >= 1.0 || r2 == 0.0) continue;
      if (++m == need_pairs) {
        last_attempt = last_chunk * CH + a;
        last_gauss = std::sqrt(-2.0 * std::log(r2) / r2) * X1[a];
        break;
      }
    }
  }
  // the state numpy is left in: words consumed = 4 (last_attempt + 1); an odd request caches the last pair's first deviate
  if ((n - done) & 1) {
    *gauss = last_gauss;
    *has_gauss = 1;
  }
  const long long abs_pos = (long long)pos0 + 4 * (last_attempt + 1);
  long long blk = abs_pos / N, p = abs_pos % N;
  if (p == 0 && blk > 0) { blk -= 1; p = N; }  // numpy leaves pos = 624 at a block's end (regeneration on the next read)
  // (key may BE numpy's own state array: block 0 of `raw` is a copy, so the source and the target never overlap)
  std::memcpy(key, &raw[(size_t)blk * N], N * sizeof(uint32_t));
  *pos = (int)p;
  return 0;
}

int vihds_host_abi_version(void) { return 1; }

}  // extern "C"

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
//   * the state handed back is numpy's state after the same call (key, pos, has_gauss, cached deviate);
//   * the OUTPUT is float32: the double f x is formed with a four-wide table-driven log (|error| of f x below 1e-15
//     relative) and rounded to float32; wherever that double lies within 4096 ulps of a float32 rounding boundary -- where a
//     last-bit difference from libm's log could change the rounded result, about 1.5e-5 of the values -- the value is
//     recomputed with libm's log as numpy calls it.  The float32 numbers are therefore the ones numpy's float64 stream
//     rounds to, bit for bit; the cached deviate (a double) always comes from libm.
//
// C ABI (ctypes: vi-hds_amd/vihds/nprand.py).  Host code only; not part of the HIP library.
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <immintrin.h>
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
  std::vector<long long> count;    // accepted attempts per chunk
  std::vector<long long> start;    // prefix sums of `count`
  std::vector<float> vals;         // per chunk: its accepted attempts' outputs in order (f x2, f x1), 2 CH floats
};

std::mutex g_lock;           // one call at a time (the pool and the buffers are shared)
Pool* g_pool = nullptr;
Buffers& g_buf = *new Buffers;  // (never destroyed: the helper thread may still be drawing when the process's statics are torn down)

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

// the accepted attempts moved to the front, in order (no branch); returns how many
inline long long compact_accepted(double* X1, double* X2, double* R2) {
  long long m = 0;
  for (long long a = 0; a < CH; ++a) {
    const double r2 = R2[a], x1 = X1[a], x2 = X2[a];
    X1[m] = x1;
    X2[m] = x2;
    R2[m] = r2;
    m += (r2 < 1.0) & (r2 != 0.0);
  }
  return m;
}

// log(z) = k ln2 + log(c_i) + log1p(z'/c_i - 1) over 128 sub-intervals of z' in [0.6875, 1.375) (the interval below 1 is
// centred ON 1, so that arguments near 1 keep their relative accuracy); a few 1e-16 relative -- only the float32 rounding
// of what is built on it has to agree with libm's log (see `finish_chunk`)
struct LogTable {
  alignas(16) double t[128][2];  // (1 / c_i, log c_i)
  LogTable() {
    for (int i = 0; i < 128; ++i) {
      const uint64_t lo = 0x3fe6000000000000ULL + ((uint64_t)i << 45), hi = lo + (1ULL << 45);
      double a, b;
      std::memcpy(&a, &lo, 8);
      std::memcpy(&b, &hi, 8);
      const double c = i == 79 ? 1.0 : 0.5 * (a + b);
      t[i][0] = i == 79 ? 1.0 : 1.0 / c;
      t[i][1] = i == 79 ? 0.0 : (double)-std::log((long double)t[i][0]);  // log of the c that 1/c stands for exactly
    }
  }
};
const LogTable g_log;

__attribute__((target("avx2,fma"))) inline __m256d log4(__m256d x) {
  const __m256i ix = _mm256_castpd_si256(x);
  const __m256i tmp = _mm256_sub_epi64(ix, _mm256_set1_epi64x(0x3fe6000000000000LL));
  const __m256i idx = _mm256_and_si256(_mm256_srli_epi64(tmp, 45), _mm256_set1_epi64x(127));
  const __m256i kbits = _mm256_and_si256(tmp, _mm256_set1_epi64x((long long)(0xfffULL << 52)));
  const __m256d z = _mm256_castsi256_pd(_mm256_sub_epi64(ix, kbits));
  // k = (int64) tmp >> 52: twelve bits, sign-extended, as doubles
  __m256i k12 = _mm256_srli_epi64(tmp, 52);
  k12 = _mm256_sub_epi64(_mm256_xor_si256(k12, _mm256_set1_epi64x(0x800)), _mm256_set1_epi64x(0x800));
  const __m128i k32 = _mm256_castsi256_si128(_mm256_permutevar8x32_epi32(k12, _mm256_setr_epi32(0, 2, 4, 6, 0, 0, 0, 0)));
  const __m256d k = _mm256_cvtepi32_pd(k32);
  // the four table rows by index: plain 16-byte loads and two unpacks (a vgatherqpd here was slower than libm's scalar log
  // on an EPYC 9575F, and on Intel it chained the loop's iterations through its merge destination)
  alignas(32) long long ii[4];
  _mm256_store_si256((__m256i*)ii, idx);
  const __m256d t02 = _mm256_set_m128d(_mm_load_pd(g_log.t[ii[2]]), _mm_load_pd(g_log.t[ii[0]]));
  const __m256d t13 = _mm256_set_m128d(_mm_load_pd(g_log.t[ii[3]]), _mm_load_pd(g_log.t[ii[1]]));
  const __m256d invc = _mm256_unpacklo_pd(t02, t13), logc = _mm256_unpackhi_pd(t02, t13);
  const __m256d r = _mm256_fmadd_pd(z, invc, _mm256_set1_pd(-1.0));
  __m256d p = _mm256_set1_pd(1.0 / 7.0);
  p = _mm256_fmadd_pd(p, r, _mm256_set1_pd(-1.0 / 6.0));
  p = _mm256_fmadd_pd(p, r, _mm256_set1_pd(1.0 / 5.0));
  p = _mm256_fmadd_pd(p, r, _mm256_set1_pd(-1.0 / 4.0));
  p = _mm256_fmadd_pd(p, r, _mm256_set1_pd(1.0 / 3.0));
  p = _mm256_fmadd_pd(p, r, _mm256_set1_pd(-0.5));
  const __m256d hi = _mm256_fmadd_pd(k, _mm256_set1_pd(0.693147180559945309417232121458), logc);
  const __m256d lo = _mm256_fmadd_pd(_mm256_mul_pd(r, r), p, r);
  return _mm256_add_pd(hi, lo);
}

// does any of the eight doubles have a doubtful float32 rounding -- the 29 bits below float32's mantissa within 4096 of the
// half-way pattern?  (Their magnitudes are always inside float32's normal range: |f x| <= sqrt(-2 log r2) <= 12 since
// |x| <= sqrt(r2) and r2 >= 2^-104, and |f x| >= sqrt(2 * 2^-53) * 2^-52 since r2 <= 1 - 2^-53 and |x| >= 2^-52.)
__attribute__((target("avx2,fma"))) inline bool doubtful8(__m256d a, __m256d b) {
  const __m256i lo = _mm256_castps_si256(_mm256_shuffle_ps(_mm256_castpd_ps(a), _mm256_castpd_ps(b), 0x88));  // low words
  const __m256i near = _mm256_add_epi32(_mm256_and_si256(lo, _mm256_set1_epi32(0x1fffffff)),
                                        _mm256_set1_epi32(4096 - 0x10000000));  // in [0, 8192) when doubtful
  return _mm256_movemask_epi8(_mm256_cmpeq_epi32(_mm256_srli_epi32(near, 13), _mm256_setzero_si256())) != 0;
}

// v[2j], v[2j+1] <- float32 of f x2, f x1 for the m accepted attempts (X1, X2, R2 compacted), f = sqrt(-2 log(r2) / r2)
__attribute__((target("avx2,fma"))) inline void finish_chunk(const double* X1, const double* X2, const double* R2,
                                                             long long m, float* v) {
  long long j = 0;
  for (; j + 4 <= m; j += 4) {
    const __m256d r2 = _mm256_loadu_pd(R2 + j);
    const __m256d f = _mm256_sqrt_pd(_mm256_div_pd(_mm256_mul_pd(_mm256_set1_pd(-2.0), log4(r2)), r2));
    const __m256d d2 = _mm256_mul_pd(f, _mm256_loadu_pd(X2 + j)), d1 = _mm256_mul_pd(f, _mm256_loadu_pd(X1 + j));
    // interleave (d2[0], d1[0], d2[1], d1[1]) | (d2[2], d1[2], d2[3], d1[3])
    const __m256d ul = _mm256_unpacklo_pd(d2, d1), uh = _mm256_unpackhi_pd(d2, d1);  // (a0 b0 a2 b2), (a1 b1 a3 b3)
    const __m256d first = _mm256_permute2f128_pd(ul, uh, 0x20), second = _mm256_permute2f128_pd(ul, uh, 0x31);
    _mm_storeu_ps(v + 2 * j, _mm256_cvtpd_ps(first));
    _mm_storeu_ps(v + 2 * j + 4, _mm256_cvtpd_ps(second));
    if (doubtful8(d2, d1)) {
      for (long long q = j; q < j + 4; ++q) {
        const double fe = std::sqrt(-2.0 * std::log(R2[q]) / R2[q]);
        v[2 * q] = (float)(fe * X2[q]);
        v[2 * q + 1] = (float)(fe * X1[q]);
      }
    }
  }
  for (; j < m; ++j) {
    const double fe = std::sqrt(-2.0 * std::log(R2[j]) / R2[j]);
    v[2 * j] = (float)(fe * X2[j]);
    v[2 * j + 1] = (float)(fe * X1[j]);
  }
}

// bound[0..K]: contiguous chunk ranges of equal COST for K threads, where a thread pays `RHO` per chunk before its range's end
// (regenerating the generator up to there) on top of 1 per chunk of its own (RHO: 73 ns per block x 13.1 blocks per chunk
// against 6.4 us per evaluated chunk on an EPYC 9575F; 0.15 on a Xeon as well)
inline void split_chunks(long long chunks, int K, long long* bound) {
  constexpr double RHO = 0.15;
  double lo = 0.0, hi = (double)chunks * (1.0 + RHO);
  for (int it = 0; it < 60; ++it) {
    const double T = 0.5 * (lo + hi);
    double e = 0.0;
    for (int t = 0; t < K; ++t) e = (T + e) / (1.0 + RHO);
    if (e < (double)chunks) lo = T;
    else hi = T;
  }
  double e = 0.0;
  bound[0] = 0;
  for (int t = 0; t < K; ++t) {
    e = (hi + e) / (1.0 + RHO);
    bound[t + 1] = std::min<long long>(chunks, std::max<long long>(bound[t], (long long)(e + 0.5)));
  }
  bound[K] = chunks;
}

}  // namespace

extern "C" {

// out[0..n) <- what np.random.standard_normal(n).astype(np.float32) would return from the RandomState (key, pos, has_gauss,
// gauss); the four are updated to the state numpy would be left in.  Returns 0, or -1 on bad arguments.
//
// The attempts are cut into chunks of 2048 and the chunks into one contiguous range per thread.  Nothing is shared while the
// threads work: each regenerates the generator's blocks from the caller's key up to the end of its own range (that costs
// 1/7 of evaluating the same stretch, so later threads get shorter ranges: `split_chunks`) and keeps only the blocks under its
// range, evaluates its chunks (a vectorised pass for x1, x2, r2, then the accepted ones' sqrt(-2 log r2 / r2)) into the
// chunks' own buffers; then one thread places the chunks (prefix sum over their counts), every thread copies its chunks'
// outputs to their places, and the thread that owns the last needed attempt leaves the generator state behind it.  (A first
// version had one thread regenerate for all: on a 64-core EPYC the others then pulled 2.5 MB of state across the chip.)
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
  const long long pos0 = *pos;
  std::vector<long long>& count = g_buf.count;
  std::vector<long long>& start = g_buf.start;
  std::vector<float>& vals = g_buf.vals;
  const bool odd = (n - done) & 1;
  for (double margin = 1.02;; margin = 1.0 + 2.0 * (margin - 1.0)) {  // (a second round: never seen; the margin doubles)
    const long long want_attempts = (long long)(need_pairs * 1.2732395447 * margin) + 64;
    const long long want_chunks = (want_attempts + CH - 1) / CH;
    const int K = (int)std::min<long long>(need_pairs < 4096 ? 1 : n_threads, want_chunks);
    count.resize((size_t)want_chunks);
    start.resize((size_t)want_chunks + 1);
    if (vals.size() < (size_t)want_chunks * 2 * CH) vals.resize((size_t)want_chunks * 2 * CH);
    long long bound[65];
    split_chunks(want_chunks, K, bound);
    std::atomic<int> finished(0), verdict(0);  // verdict: 1 = enough pairs, copy out; 2 = another round is needed
    long long last_chunk = 0;
    long long* C = count.data();
    long long* S = start.data();
    float* V = vals.data();
    g_pool->run([&](int t) {
      static thread_local std::vector<uint32_t> local;  // the blocks first .. last of this thread's range
      const long long c0 = bound[t], c1 = bound[t + 1];
      // words of the range: [pos0 + 4 c0 CH, pos0 + 4 c1 CH); the block before is kept as well (the state handed back
      // when the last attempt ends exactly at a block's end)
      const long long first = std::max<long long>((pos0 + 4 * c0 * CH) / N - 1, 0), last = (pos0 + 4 * c1 * CH + N - 1) / N;
      if (local.size() < (size_t)(last - first + 1) * N) local.resize((size_t)(last - first + 1) * N);
      uint32_t* L = local.data();
      {
        uint32_t roll[2][N];
        const uint32_t* prev = key;
        for (long long b = 1; b <= first; ++b) {  // up to the range: nothing kept
          regenerate(prev, roll[b & 1]);
          prev = roll[b & 1];
        }
        std::memcpy(L, prev, N * sizeof(uint32_t));
        for (long long b = first + 1; b <= last; ++b) regenerate(L + (size_t)(b - first - 1) * N, L + (size_t)(b - first) * N);
      }
      const uint32_t* R = L - (size_t)first * N;  // R[w] = raw word w of the stream, for the words of this range
      double X1[CH], X2[CH], R2[CH];
      for (long long c = c0; c < c1; ++c) {
        chunk_attempts(R, pos0, c * CH, X1, X2, R2);
        const long long m = compact_accepted(X1, X2, R2);
        finish_chunk(X1, X2, R2, m, V + (size_t)c * 2 * CH);
        C[c] = m;
      }
      finished.fetch_add(1, std::memory_order_acq_rel);
      if (t == 0) {
        for (int spins = 0; finished.load(std::memory_order_acquire) < K; ++spins) {
          if (spins < 4096) cpu_relax();
          else std::this_thread::yield();
        }
        S[0] = 0;
        for (long long c = 0; c < want_chunks; ++c) {
          S[c + 1] = S[c] + C[c];
          if (S[c] < need_pairs) last_chunk = c;
        }
        verdict.store(S[want_chunks] >= need_pairs ? 1 : 2, std::memory_order_release);
      } else {
        for (int spins = 0; verdict.load(std::memory_order_acquire) == 0; ++spins) {
          if (spins < 4096) cpu_relax();
          else std::this_thread::yield();
        }
      }
      if (verdict.load(std::memory_order_acquire) != 1) return;
      for (long long c = c0; c < c1 && c <= last_chunk; ++c) {
        const long long m0 = S[c], m1 = std::min<long long>(S[c + 1], need_pairs);
        const long long o = done + 2 * m0;
        long long cnt = 2 * (m1 - m0);
        if (o + cnt > n) cnt = n - o;  // (an odd request: the last pair's second deviate stays cached)
        if (cnt > 0) std::memcpy(out + o, V + (size_t)c * 2 * CH, (size_t)cnt * sizeof(float));
      }
      if (c0 <= last_chunk && last_chunk < c1) {
        // the attempt that yielded the last needed pair: the (need_pairs - S[last_chunk])-th accepted one of its chunk
        long long last_attempt = -1;
        double last_gauss = 0.0;
        chunk_attempts(R, pos0, last_chunk * CH, X1, X2, R2);
        long long m = S[last_chunk];
        for (long long a = 0; a < CH; ++a) {
          const double r2 = R2[a];
          if (r2 >= 1.0 || r2 == 0.0) continue;
          if (++m == need_pairs) {
            last_attempt = last_chunk * CH + a;
            last_gauss = std::sqrt(-2.0 * std::log(r2) / r2) * X1[a];
            break;
          }
        }
        // the state numpy is left in: words consumed = 4 (last_attempt + 1); an odd request caches the last pair's first
        // deviate
        if (odd) {
          *gauss = last_gauss;
          *has_gauss = 1;
        }
        const long long abs_pos = pos0 + 4 * (last_attempt + 1);
        long long blk = abs_pos / N, p = abs_pos % N;
        if (p == 0 && blk > 0) { blk -= 1; p = N; }  // numpy leaves pos = 624 at a block's end (regeneration on the next read)
        if (blk > 0) std::memcpy(key, R + (size_t)blk * N, N * sizeof(uint32_t));  // (block 0 IS the caller's key)
        *pos = (int)p;
      }
    }, K);
    if (verdict.load() == 1) return 0;
  }
}

// The same draw on a native helper thread: `start` returns at once, `wait` blocks until the numbers are in `out` and numpy's
// state (key, pos: read and advanced IN PLACE) is the state after the draw.  For an even n on a state without a cached
// deviate only (the training loop's draws).  A Python helper thread cannot do this job: it needs the interpreter lock to
// start, which the main thread holds while it queues the step -- measured, the "overlapped" draw ran after the main thread's
// work, not beside it.  Up to two draws may be queued: the helper then goes from one straight into the next instead of
// waiting for the main thread to hand it the next job (its wake-ups were a third of the step).  The caller must leave numpy's
// generator alone until every started draw has been waited for.
struct AsyncDraw {
  struct Job {
    uint32_t* key;
    int* pos;
    float* out;
    long long n;
    int threads, rc;
  };
  static constexpr int DEPTH = 2;  // draws that may be queued or running at once (the helper works through them in order)
  std::thread worker;
  std::mutex m;
  std::condition_variable cv;
  Job ring[DEPTH];
  unsigned long submitted = 0, completed = 0, collected = 0;  // job j lives in ring[j % DEPTH]
  void loop() {
    for (;;) {
      Job* j;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return completed < submitted; });
        j = &ring[completed % DEPTH];
      }
      int has_gauss = 0;
      double gauss = 0.0;
      const int r = vihds_np_randn_f32(j->key, j->pos, &has_gauss, &gauss, j->out, j->n, j->threads);
      {
        std::lock_guard<std::mutex> lk(m);
        j->rc = r;
        ++completed;
      }
      cv.notify_all();
    }
  }
};
static AsyncDraw* g_async = nullptr;
static std::mutex g_async_create;

int vihds_np_randn_f32_start(uint32_t* key, int* pos, float* out, long long n, int n_threads) {
  if (!key || !pos || !out || n <= 0 || (n & 1)) return -1;
  {
    std::lock_guard<std::mutex> lk(g_async_create);
    if (!g_async) {
      g_async = new AsyncDraw();
      g_async->worker = std::thread([] { g_async->loop(); });
      g_async->worker.detach();
    }
  }
  AsyncDraw& a = *g_async;
  {
    std::lock_guard<std::mutex> lk(a.m);
    if (a.submitted - a.collected >= (unsigned long)AsyncDraw::DEPTH) return -2;  // the queue is full
    a.ring[a.submitted % AsyncDraw::DEPTH] = AsyncDraw::Job{key, pos, out, n, n_threads, 0};
    ++a.submitted;
  }
  a.cv.notify_all();
  return 0;
}

// result of the OLDEST draw started and not collected yet (0 = done); -3: there is none
int vihds_np_randn_f32_wait(void) {
  if (!g_async) return -3;
  AsyncDraw& a = *g_async;
  std::unique_lock<std::mutex> lk(a.m);
  if (a.collected == a.submitted) return -3;
  a.cv.wait(lk, [&] { return a.completed > a.collected; });
  const int rc = a.ring[a.collected % AsyncDraw::DEPTH].rc;
  ++a.collected;
  return rc;
}

int vihds_host_abi_version(void) { return 4; }
// 1 when this CPU runs the AVX2 code the library was compiled to (the Python binding asks once and falls back to numpy
// itself otherwise: on a host without AVX2 the first vector instruction would be a SIGILL)
int vihds_host_cpu_ok(void) {
#if defined(__x86_64__) || defined(__i386__)
  return __builtin_cpu_supports("avx2") ? 1 : 0;
#else
  return 0;
#endif
}

}  // extern "C"

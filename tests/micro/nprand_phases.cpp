// Where a single thread's time goes in csrc/host/vihds_nprand.cpp on the box at hand (build: see the first line printed).
//   g++ -O3 -mavx2 -pthread -std=c++17 tests/micro/nprand_phases.cpp -o /tmp/nprand_phases && /tmp/nprand_phases
#include <chrono>
#include <cstdio>
#include "../../vi-hds_amd/csrc/host/vihds_nprand.cpp"
#include <algorithm>
template <class F> double tm(F f, int reps = 41) {  // median
  f();
  std::vector<double> ts;
  for (int r = 0; r < reps; ++r) {
    auto t0 = std::chrono::steady_clock::now();
    f();
    ts.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}
int main() {
  std::vector<uint32_t> raw(1100 * 624, 12345u);
  for (int i = 0; i < 624; ++i) raw[i] = i * 2654435761u;
  printf("252 000 normals = 78 chunks of 2048 attempts = 1030 generator blocks; one thread\n");
  printf("regenerate 1030 blocks        %.3f ms\n", tm([&] { for (int b = 1; b < 1030; ++b) regenerate(&raw[(b - 1) * 624], &raw[b * 624]); }));
  static double X1[CH], X2[CH], R2[CH];
  static float v[2 * CH];
  double s = 0;
  printf("attempts                      %.3f ms\n", tm([&] { for (int c = 0; c < 78; ++c) { chunk_attempts(raw.data(), 0, c * CH, X1, X2, R2); s += R2[5]; } }));
  printf("attempts + compact            %.3f ms\n", tm([&] { for (int c = 0; c < 78; ++c) { chunk_attempts(raw.data(), 0, c * CH, X1, X2, R2); s += compact_accepted(X1, X2, R2); } }));
  printf("attempts + compact + finish   %.3f ms\n", tm([&] { for (int c = 0; c < 78; ++c) { chunk_attempts(raw.data(), 0, c * CH, X1, X2, R2); long long m = compact_accepted(X1, X2, R2); finish_chunk(X1, X2, R2, m, v); s += v[3]; } }));
  printf("attempts + compact + libm     %.3f ms\n", tm([&] { for (int c = 0; c < 78; ++c) { chunk_attempts(raw.data(), 0, c * CH, X1, X2, R2); long long m = compact_accepted(X1, X2, R2);
      for (long long j = 0; j < m; ++j) { const double f = std::sqrt(-2.0 * std::log(R2[j]) / R2[j]); v[2 * j] = (float)(f * X2[j]); v[2 * j + 1] = (float)(f * X1[j]); } s += v[3]; } }));
  std::vector<float> out(252000);
  std::vector<uint32_t> key(raw.begin(), raw.begin() + 624);
  int pos = 0, hg = 0; double g = 0;
  for (int th : {1, 2, 4, 8, 12, 16})
    printf("whole call, %d thread(s)       %.3f ms\n", th, tm([&] { vihds_np_randn_f32(key.data(), &pos, &hg, &g, out.data(), 252000, th); }));
  printf("(%g)\n", s);
  return 0;
}

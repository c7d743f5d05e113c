// BatchSort::predict through the C ABI from a C++ host (no Python between the calls): the synchronous call against the result handle
// (sa_tracker_predict_batch_begin + one sa_batch_result_get / _take per scene).  S scenes x n objects of the world scripts/bench_batch_tracker.py
// uses (dense random boxes, jittered every frame; [churn]: that fraction of the objects replaced every frame), device upkeep.
//   g++ -O2 -std=c++17 -I include scripts/micro/batch_handle_bench.cpp -L similari_amd/lib -lsimilari_assoc -Wl,-rpath,$PWD/similari_amd/lib -o /tmp/batch_handle_bench
//   /tmp/batch_handle_bench [scenes] [objects] [frames] [churn] [devices]
// devices: a comma-separated list of HIP ordinals — the tracker becomes a device GROUP (sa_tracker_options.n_devices / devices: one engine per
// entry, scene_id % n; "0,0" = two engines on one GPU), everything else unchanged: ONE tracker object, ONE handle.
#include "similari_tracker.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
int main(int argc, char** argv) {
  const uint32_t S = argc > 1 ? atoi(argv[1]) : 8, n = argc > 2 ? atoi(argv[2]) : 500, frames = argc > 3 ? atoi(argv[3]) : 60;
  const float churn = argc > 4 ? (float)atof(argv[4]) : 0.0f;
  std::vector<int32_t> devices;
  if (argc > 5)
    for (const char* c = argv[5]; *c;) { devices.push_back(atoi(c)); while (*c && *c != ',') ++c; if (*c == ',') ++c; }
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> u(0.f, 1.f);
  std::normal_distribution<float> g(0.f, 2.f);
  // the world of scripts/bench_batch_tracker.py (similari_amd/synth.py: dense_boxes / jitter_boxes): centres uniform over 1920 x 1080,
  // height U(40, 200), aspect U(0.3, 0.6), confidence U(0.3, 1); every frame centre + N(0, 2 px), size x (1 +- 0.001)
  auto fresh = [&](sa_box& b, uint32_t) {
    b = sa_box{};
    b.xc = 1920.f * u(rng); b.yc = 1080.f * u(rng);
    b.aspect = 0.3f + 0.3f * u(rng); b.height = 40.f + 160.f * u(rng); b.confidence = 0.3f + 0.7f * u(rng);
  };
  std::vector<std::vector<sa_box>> world(S, std::vector<sa_box>(n));
  for (auto& w : world) for (uint32_t k = 0; k < n; ++k) fresh(w[k], k);
  double med[3] = {0, 0, 0}, first[1] = {0}, ret[1] = {0};
  for (int mode = 0; mode < 3; ++mode) {   // 0 = synchronous, 1 = handle + get (a copy per scene), 2 = handle + take (in place)
    sa_tracker_options o;
    sa_tracker_options_default(&o, 0);
    o.history_length = 3; o.max_idle_epochs = 3; o.batch_ids = 1; o.device_upkeep = 1;
    if (devices.size() > 1) { o.n_devices = (uint32_t)devices.size(); o.devices = devices.data(); }
    sa_tracker* t = nullptr;
    if (sa_tracker_create(&o, &t) != 0) { printf("create failed: %s\n", sa_tracker_last_error(nullptr)); return 1; }
    std::vector<std::vector<sa_observation>> obs(S, std::vector<sa_observation>(n));
    std::vector<std::vector<sa_sort_track>> out(S, std::vector<sa_sort_track>(n));
    std::vector<uint64_t> ids(S);
    std::vector<uint32_t> counts(S, n);
    std::vector<const sa_observation*> po(S);
    std::vector<sa_sort_track*> pt(S);
    for (uint32_t s = 0; s < S; ++s) { ids[s] = s; po[s] = obs[s].data(); pt[s] = out[s].data(); }
    std::vector<double> tt, tf, tr;
    for (uint32_t f = 0; f < frames; ++f) {
      for (uint32_t s = 0; s < S; ++s)
        for (uint32_t k = 0; k < n; ++k) {
          sa_box& b = world[s][k];
          if (churn > 0.f && u(rng) < churn) fresh(b, k);
          else {
            b.xc += g(rng); b.yc += g(rng);
            b.height *= 1.0f + 0.002f * (u(rng) - 0.5f); b.aspect *= 1.0f + 0.002f * (u(rng) - 0.5f); b.confidence = 0.3f + 0.7f * u(rng);
          }
          sa_observation& ob = obs[s][k];
          ob = sa_observation{};
          ob.bbox = b; ob.feature_quality = std::nanf(""); ob.own_area = std::nanf("");
        }
      const auto t0 = clk::now();
      if (mode == 0) {
        if (sa_tracker_predict_batch(t, S, ids.data(), counts.data(), po.data(), pt.data()) != 0) { printf("predict failed: %s\n", sa_tracker_last_error(t)); return 1; }
        tt.push_back(us(t0, clk::now()));
      } else {
        sa_batch_result* r = nullptr;
        if (sa_tracker_predict_batch_begin(t, S, ids.data(), counts.data(), po.data(), &r) != 0) { printf("begin failed: %s\n", sa_tracker_last_error(t)); return 1; }
        const auto t1 = clk::now();
        for (uint32_t k = 0; k < S; ++k) {
          uint64_t sid; uint32_t cnt;
          const sa_sort_track* view = nullptr;
          if ((mode == 1 ? sa_batch_result_get(r, &sid, out[0].data(), n, &cnt) : sa_batch_result_take(r, &sid, &view, &cnt)) != 0) { printf("get failed\n"); return 1; }
          if (k == 0) tf.push_back(us(t0, clk::now()));
        }
        tt.push_back(us(t0, clk::now()));
        tr.push_back(us(t0, t1));
        sa_batch_result_free(r);
      }
    }
    auto median = [](std::vector<double> v) { v.erase(v.begin(), v.begin() + std::min<size_t>(5, v.size() - 1)); std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    med[mode] = median(tt);
    if (mode == 2) { first[0] = median(tf); ret[0] = median(tr); }
    sa_tracker_destroy(t);
  }
  printf("{\"host\": \"C++\", \"tracker\": \"BatchSort\", \"engines\": %u, \"scenes\": %u, \"objects_per_scene\": %u, \"churn_per_frame\": %.2f, \"us_per_predict_sync\": %.1f, \"us_per_predict_through_the_handle_get\": %.1f, "
         "\"us_per_predict_through_the_handle_take\": %.1f, \"us_until_begin_returns\": %.1f, \"us_until_first_scene\": %.1f}\n", (uint32_t)(devices.size() > 1 ? devices.size() : 1), S, n, (double)churn, med[0], med[1], med[2], ret[0], first[0]);
  return 0;
}

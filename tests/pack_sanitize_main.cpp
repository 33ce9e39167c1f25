// Host-only driver of the weight packer (audiosourcesep_amd/csrc/glowk_pack.h) for the sanitizer builds of
// tests/test_pack_sanitizers.py:  g++ -fsanitize=address,undefined  /  g++ -fsanitize=thread.
// Every step packs into a buffer of EXACTLY step_layout(c, F).total floats (its own heap block, so an index one past the end is an
// ASan report, not a write into the neighbouring step), several steps at once on the packer's own thread pool, and the images are
// checked against what a scalar reading of the layouts demands: with index-coded kernels every element of a convolution kernel
// must land in the exact-fp32 ring image the number of times the kernels read it, and nothing else may be there.
#include "../audiosourcesep_amd/csrc/glowk_pack.h"

#include <cstdio>
#include <map>
#include <random>

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); ++fails; } } while (0)

static Level make_level(const glowk_config& cfg, int h, int w, int c, int K, std::mt19937& rng, bool coded) {
  Level lv;
  lv.h = h; lv.w = w; lv.c = c; lv.z_off = 0; lv.z_width = 0; lv.Cz = 0;
  std::normal_distribution<float> nd(0.0f, 0.05f);
  for (int id = 0; id < GLOWK_NUM_STEP_TENSORS; ++id) {
    lv.host[id].resize(K);
    for (int k = 0; k < K; ++k) {
      std::vector<float>& t = lv.host[id][k];
      t.assign(step_tensor_size(cfg, lv, id), 0.0f);
      for (float& v : t) v = nd(rng);
    }
  }
  for (int k = 0; k < K; ++k) {
    // a well-conditioned 1x1: P = a cyclic shift, unit lower L, U with a +-e^{log_S} diagonal
    std::vector<float>& P = lv.host[GLOWK_INV1X1_P][k];
    std::fill(P.begin(), P.end(), 0.0f);
    for (int i = 0; i < c; ++i) P[(size_t)i * c + (i + 1 + k) % c] = 1.0f;
    for (int i = 0; i < c; ++i) lv.host[GLOWK_INV1X1_SIGN_S][k][i] = (i & 1) ? -1.0f : 1.0f;
    std::fill(lv.host[GLOWK_INV1X1_P_INV][k].begin(), lv.host[GLOWK_INV1X1_P_INV][k].end(), 0.0f);
    for (float& v : lv.host[GLOWK_BN1_VAR][k]) v = 1.0f + std::fabs(v);
    for (float& v : lv.host[GLOWK_BN2_VAR][k]) v = 1.0f + std::fabs(v);
    for (float& v : lv.host[GLOWK_BN1_GAMMA][k]) v += 1.0f;
    for (float& v : lv.host[GLOWK_BN2_GAMMA][k]) v += 1.0f;
    if (coded) {
      // K2 carries exact integer codes; the other two kernels are zero, so that whatever else sits in the K2 chunks shows up
      std::vector<float>& K2 = lv.host[GLOWK_CONV2_KERNEL][k];
      for (size_t i = 0; i < K2.size(); ++i) K2[i] = (float)(i + 1);
      std::fill(lv.host[GLOWK_CONV1_KERNEL][k].begin(), lv.host[GLOWK_CONV1_KERNEL][k].end(), 0.0f);
      std::fill(lv.host[GLOWK_CONV3_KERNEL][k].begin(), lv.host[GLOWK_CONV3_KERNEL][k].end(), 0.0f);
    }
  }
  return lv;
}

int main(int argc, char** argv) {
  const unsigned threads = argc > 1 ? (unsigned)std::atoi(argv[1]) : 4u;
  std::mt19937 rng(1234);
  const int shapes[][2] = {{4, 128}, {8, 128}, {16, 256}, {32, 128}, {4, 512}, {16, 384}, {32, 512}};
  for (const auto& sh : shapes) {
    const int c = sh[0], F = sh[1], K = 3;
    glowk_config cfg{};
    cfg.H = 16; cfg.W = 16; cfg.C = 1; cfg.L = 2; cfg.K = K; cfg.F = F; cfg.learntop = 1; cfg.use_logit = 0;
    cfg.minval = -100.f; cfg.maxval = 20.f; cfg.alpha = 1e-10f; cfg.bn_eps = 1e-3f;
    const StepLayout SL = step_layout(c, F);
    for (int coded = 0; coded < 2; ++coded) {
      std::vector<Level> levels;
      levels.push_back(make_level(cfg, 8, 8, c, K, rng, coded != 0));
      // one heap block per step, exactly as large as the layout says: a packer index past the end is a sanitizer report
      std::vector<std::vector<float>> bufs(K, std::vector<float>(SL.total, 0.0f));
      std::vector<PackJob> jobs;
      // pack_all_steps addresses `stage + off`: give every job the offset of its own block relative to block 0
      // (pointer differences between separate heap blocks are not arithmetic the standard blesses, so pack them one pool at a time)
      for (int k = 0; k < K; ++k) {
        std::vector<PackJob> one{PackJob{0, k, 0, 0.0, {1, 1, 1, 1, 1, 1, 0, 0}, std::string(), false}};
        pack_all_steps(cfg, levels, bufs[k].data(), one, 1);
        jobs.push_back(one[0]);
      }
      // the thread pool itself (TSan): all steps into one staging arena, as glowk_finalize_weights does
      std::vector<float> stage((size_t)K * SL.total, 0.0f);
      std::vector<PackJob> pool;
      for (int k = 0; k < K; ++k) pool.push_back(PackJob{0, k, (size_t)k * SL.total, 0.0, {1, 1, 1, 1, 1, 1, 0, 0}, std::string(), false});
      pack_all_steps(cfg, levels, stage.data(), pool, threads);
      for (int k = 0; k < K; ++k) {
        CHECK(jobs[k].ok && pool[k].ok, "c=%d F=%d step %d: %s", c, F, k, jobs[k].err.c_str());
        CHECK(std::memcmp(bufs[k].data(), stage.data() + (size_t)k * SL.total, SL.total * sizeof(float)) == 0,
              "c=%d F=%d step %d: threaded packing differs from the single-threaded one", c, F, k);
        CHECK(std::isfinite(jobs[k].ldc), "log-det constant");
        for (int i = 0; i < 8; ++i) CHECK(std::isfinite(jobs[k].sc[i]) && jobs[k].sc[i] >= 0.0f, "scale %d", i);
      }
      if (coded) {
        // exact-fp32 forward ring image: K2 chunk fi = slot fi, main part (NF * 1024 floats); every K2[f_in][f_out] exactly once over
        // the NF chunks, chunk fi holding exactly the rows f_in in [32 fi, 32 fi + 32); same for the backward image (K2^T)
        const int NF = F / 32;
        for (int img = 0; img < 2; ++img) {
          const size_t base = img ? SL.RBp : SL.R0p, slot = img ? SL.slotB : SL.slot0;
          std::map<int, int> seen;
          for (int fi = 0; fi < NF; ++fi)
            for (size_t i = 0; i < (size_t)NF * 1024; ++i) {
              const float v = bufs[0][base + (size_t)fi * slot + i];
              CHECK(v >= 1.0f && v <= (float)(F * F) && v == std::floor(v), "c=%d F=%d image %d: foreign value %g in a K2 chunk", c, F, img, v);
              const int code = (int)v - 1, fin = code / F, fout = code % F;
              CHECK((img ? fout : fin) / 32 == fi, "c=%d F=%d image %d: K2[%d][%d] in chunk %d", c, F, img, fin, fout, fi);
              ++seen[code];
            }
          CHECK((int)seen.size() == F * F, "c=%d F=%d image %d: %zu distinct K2 elements", c, F, img, seen.size());
          for (const auto& kv : seen) CHECK(kv.second == 1, "K2 element %d appears %d times", kv.first, kv.second);
        }
      }
    }
  }
  if (fails) { std::fprintf(stderr, "%d check(s) failed\n", fails); return 1; }
  std::printf("PACK_SANITIZE_OK\n");
  return 0;
}

// AddressSanitizer / UBSan run of the host-side parsers (CPU only; GPU ASan is not available on the pool):
// the HEIF box reader and the HEVC entropy decoder are fed mutated inputs.  Build + run: tools/asan/run.sh
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "heif_file.h"
#include "heif_mi355x.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd()
{
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 32);
}
static std::vector<uint8_t> slurp(const char* path)
{
  std::vector<uint8_t> v;
  FILE* f = std::fopen(path, "rb");
  if (!f) return v;
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  v.resize(n > 0 ? (size_t)n : 0);
  if (n > 0 && std::fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
  std::fclose(f);
  return v;
}
static void mutate(std::vector<uint8_t>& b, size_t span)
{
  if (b.empty()) return;
  if (span > b.size()) span = b.size();
  const int n = 1 + rnd() % 5;
  for (int i = 0; i < n; i++) {
    const size_t at = rnd() % span;
    switch (rnd() % 4) {
      case 0: b[at] ^= (uint8_t)(1u << (rnd() % 8)); break;
      case 1: b[at] = (uint8_t)rnd(); break;
      case 2: if (at + 4 <= b.size()) { const uint32_t v = rnd() % 3 ? 0xFFFFFFFFu : rnd(); std::memcpy(&b[at], &v, 4); } break;
      default: b.resize(at + 1); span = span > b.size() ? b.size() : span; break;
    }
  }
}

// (common.cpp's test hooks reach into the device side of the library, which this host-only build leaves out)
extern "C" void hm_chain_test_knobs(int, int) {}

int main(int argc, char** argv)
{
  int iterations = 2000;
  if (const char* e = std::getenv("HM_FUZZ_ITER")) iterations = std::atoi(e);
  long heif_ok = 0, heif_err = 0, hevc_ok = 0, hevc_err = 0, stream_ok = 0, stream_err = 0;
  for (int a = 1; a < argc; a++) {
    const std::string path = argv[a];
    const std::vector<uint8_t> seed = slurp(argv[a]);
    if (seed.empty()) { std::fprintf(stderr, "cannot read %s\n", argv[a]); return 2; }
    const bool is_heif = path.size() > 5 && path.substr(path.size() - 5) == ".heic";
    for (int it = 0; it < iterations; it++) {
      std::vector<uint8_t> b = seed;
      if (it) mutate(b, is_heif ? 4096 : b.size());
      if (is_heif) {
        hm::HeifFile f;
        hm::HeifError err;
        if (!f.parse(b.data(), b.size(), err)) { heif_err++; continue; }
        heif_ok++;
        for (const auto& kv : f.items()) {
          const uint32_t id = kv.first;
          const hm::Item* item = &kv.second;
          if (item->type == "grid") { hm::GridInfo g; f.grid_info(id, g, err); }
          else if (item->type == "hvc1") {
            std::vector<uint8_t> data;
            if (f.hevc_data(id, data, err)) {
              uint8_t* blob = nullptr; size_t n = 0;
              if (hm_hevc_parse(data.data(), data.size(), 0, &blob, &n) == 0) { hevc_ok++; hm_free(blob); }
              else hevc_err++;
            }
          }
        }
      }
      else {
        uint8_t* blob = nullptr; size_t n = 0;
        // every second mutation through the concealing parse (HM_PARSE_CONCEAL: damaged slice data is taken back to whole CTUs and
        // filled in - the paths that truncate record lists)
        hm_parse_options po;
        po.annexb = path.find(".hevc") != std::string::npos ? 1 : 0; po.threads = 1;
        po.record_order = (it & 2 ? HM_RECORDS_SPLIT : HM_RECORDS_AUTO) | (it & 1 ? HM_PARSE_CONCEAL : 0);
        if (hm_hevc_parse_opts(b.data(), b.size(), &po, &blob, &n) == 0) {
          hevc_ok++;
          // the command stream as foreign input: the validator must accept the parser's output and survive anything
          if (hm_stream_validate(blob, n) != 0) {
            std::fprintf(stderr, "validator rejects a parser stream (%s, mutation %d): %s\n", argv[a], it, hm_last_error());
            if (const char* dump = std::getenv("HM_FUZZ_DUMP")) { if (FILE* f = std::fopen(dump, "wb")) { std::fwrite(b.data(), 1, b.size(), f); std::fclose(f); } }
            return 3;
          }
          std::vector<uint8_t> s(blob, blob + n);
          for (int k = 0; k < 8; k++) {
            std::vector<uint8_t> m = s;
            mutate(m, m.size());
            (hm_stream_validate(m.data(), m.size()) == 0 ? stream_ok : stream_err)++;
            const size_t cut = rnd() % (m.size() + 1);
            hm_stream_validate(m.data(), cut);
          }
          hm_free(blob);
        }
        else hevc_err++;
      }
    }
  }
  std::printf("heif ok %ld err %ld | hevc ok %ld err %ld | mutated command streams accepted %ld rejected %ld\n", heif_ok, heif_err, hevc_ok, hevc_err,
              stream_ok, stream_err);
  return 0;
}

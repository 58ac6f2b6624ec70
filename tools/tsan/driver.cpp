// tools/tsan/run.sh: parses every file given (length-prefixed NAL records) with argv[1] host threads.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "heif_mi355x.h"
// (common.cpp's test hooks reach into the device side of the library, which this host-only build leaves out)
extern "C" void hm_chain_test_knobs(int, int) {}

int main(int argc, char** argv)
{
  const int threads = argc > 1 ? atoi(argv[1]) : 4;
  int ok = 0, refused = 0;
  for (int i = 2; i < argc; i++) {
    FILE* f = fopen(argv[i], "rb");
    if (!f) continue;
    std::vector<uint8_t> d;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(f);
    uint8_t* blob = nullptr;
    size_t size = 0;
    hm_parse_options po;
    po.annexb = 0; po.threads = threads; po.record_order = HM_RECORDS_AUTO;
    if (hm_hevc_parse_opts(d.data(), d.size(), &po, &blob, &size) == HM_OK) { ok++; hm_free(blob); }
    else refused++;
  }
  printf("parsed %d, refused %d\n", ok, refused);
  return 0;
}

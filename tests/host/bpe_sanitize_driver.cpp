// Host driver of the CPU sanitizer target (`python -m fitclip_amd.build --host-asan`): fitclip_amd/csrc/bpe.cpp compiled with
// g++ -fsanitize=address,undefined and driven through its C ABI (include/fitclip_hip.h: fc_bpe_*) - no GPU, no HIP.  Test
// infrastructure (tests/test_sanitize.py): the fixtures' texts must come out with the fixtures' ids, and a seeded corpus of
// hostile inputs - random bytes, truncated UTF-8, 100 KB tokens, out-of-range ids, malformed merge files - must pass through
// without a sanitizer report (any report aborts the process: -fno-sanitize-recover).
//   bpe_asan <merges.gz> texts <file>        one text per line -> "ids: a b c ..." per line (context 77, truncate)
//   bpe_asan <merges.gz> fuzz <seed> <n>     n random inputs through encode / tokenize / decode -> "fuzz ok <calls>"
//   bpe_asan - create <file>...              fc_bpe_create on every file -> "rc <code>" per file
#include "../../include/fitclip_hip.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

namespace fc {
int fail(int code, const char* fmt, ...) {  // (api.hip owns the real one: thread-local message + code)
  (void)fmt;
  return code;
}
}  // namespace fc

static unsigned long long rng_state = 1;
static unsigned rnd() {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (unsigned)(rng_state >> 11);
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string mode = argv[2];
  if (mode == "create") {
    for (int i = 3; i < argc; ++i) {
      fc_bpe* t = nullptr;
      const int rc = fc_bpe_create(argv[i], 77, &t);
      printf("rc %d\n", rc);
      if (rc == 0) fc_bpe_destroy(t);
    }
    return 0;
  }
  fc_bpe* t = nullptr;
  if (fc_bpe_create(argv[1], 77, &t) != 0) { fprintf(stderr, "cannot load %s\n", argv[1]); return 3; }
  if (mode == "texts" && argc == 4) {
    std::ifstream in(argv[3]);
    std::string line;
    std::vector<int64_t> row(77);
    while (std::getline(in, line)) {
      const char* p = line.c_str();
      if (fc_bpe_tokenize(t, &p, 1, 1, row.data()) != 0) return 4;
      printf("ids:");
      for (int64_t v : row) printf(" %lld", (long long)v);
      printf("\n");
    }
  } else if (mode == "fuzz" && argc == 5) {
    rng_state = strtoull(argv[3], nullptr, 10) * 2654435761ull + 88172645463325252ull;
    const int n = atoi(argv[4]), vocab = fc_bpe_vocab_size(t);
    long calls = 0;
    static const char* frag[] = {"\xC3", "\xE2\x82", "\xF0\x9F\x98", "\xF4\x90\x80\x80", "\xED\xA0\x80", "\xC0\x80", "<|startoftext|>",
                                 "<|endoftext|>", "'s", "'LL", "\xE2\x80\x83", "\t\n ", "\xEF\xBF\xBD", "123", "\xD9\xA3"};
    for (int it = 0; it < n; ++it) {
      std::string s;
      const int kind = rnd() % 6;
      const int len = kind == 5 ? 100000 + (int)(rnd() % 5000) : (int)(rnd() % 300);
      for (int k = 0; k < len; ++k) {
        if (kind == 0) s.push_back((char)(1 + rnd() % 255));                         // any non-NUL byte
        else if (kind == 1) s += frag[rnd() % (sizeof(frag) / sizeof(frag[0]))];      // truncated / odd UTF-8, specials
        else if (kind == 2) s.push_back((char)(0x80 + rnd() % 0x80));                 // continuation / lead bytes only
        else if (kind == 5) s.push_back((char)('a' + rnd() % 3));                     // one 100 KB token
        else s.push_back(" abc\xC3\xA9'12"[rnd() % 10]);
      }
      std::vector<int64_t> ids(s.size() + 8);
      const int got = fc_bpe_encode(t, s.c_str(), ids.data(), (int)ids.size());
      ++calls;
      if (got < 0) return 5;
      // a capacity smaller than the result, and a null output with capacity 0
      (void)fc_bpe_encode(t, s.c_str(), ids.data(), (int)(rnd() % 4));
      (void)fc_bpe_encode(t, s.c_str(), nullptr, 0);
      std::vector<int64_t> row(2 * 77);
      const char* two[2] = {s.c_str(), ""};
      if (fc_bpe_tokenize(t, two, 2, 1, row.data()) != 0) return 6;
      (void)fc_bpe_tokenize(t, two, 2, 0, row.data());                                // too long without truncate: an error code
      // decode: what was encoded, then ids out of range in both directions
      const int m = got < (int)ids.size() ? got : (int)ids.size();
      std::vector<char> out(8 * (size_t)m + 16);
      (void)fc_bpe_decode(t, ids.data(), m, out.data(), (int)out.size());
      (void)fc_bpe_decode(t, ids.data(), m, out.data(), (int)(rnd() % 8));            // too small: length only
      int64_t bad[4] = {(int64_t)vocab, -1, (int64_t)1 << 40, (int64_t)(rnd() % (unsigned)vocab)};
      (void)fc_bpe_decode(t, bad, 4, out.data(), (int)out.size());
      (void)fc_bpe_decode(t, bad + 3, 1, out.data(), (int)out.size());
      calls += 8;
    }
    printf("fuzz ok %ld\n", calls);
  } else {
    fc_bpe_destroy(t);
    return 2;
  }
  fc_bpe_destroy(t);
  return 0;
}

// CLIP byte-level BPE tokenizer, host C++ behind the C ABI (fc_bpe_*).
//
// Restates `SimpleTokenizer` (aligner/encoder/slip.py:75-164; the same algorithm as `clip.tokenize`, which the plugin
// calls at aligner/encoder/clip_video_text_encoder.py:64-65) for a LOCAL `bpe_simple_vocab_16e6.txt.gz`-style file:
//   vocabulary = 256 byte characters (published order) | the same + "</w>" | one entry per merge line | SOT | EOT,
//   ids by list position; pieces of the text are found with the CLIP pattern
//       <|startoftext|> | <|endoftext|> | 's | 't | 're | 've | 'm | 'll | 'd | [\p{L}]+ | [\p{N}] | [^\s\p{L}\p{N}]+
//   (ordered alternation, case-insensitive), their UTF-8 bytes are merged greedily by rank, and the ids are framed
//   as SOT ... EOT, cut to the context length (EOT put back in the last slot: `truncate=True`) and zero padded.
// Text CLEANING (html.unescape twice, white-space collapse, str.lower - slip.py:60-72,138) stays in the Python caller:
// those are the reference's own library calls; this file receives the cleaned, lower-cased UTF-8.
//
// The byte <-> printable-character table of the reference is a bijection, so the merge arithmetic is done directly on
// byte strings (a merge file is translated once at load time); "</w>" is the same four ASCII characters on both sides.
#include "../../include/fitclip_hip.h"

#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace fc {
int fail(int code, const char* fmt, ...);
}

namespace {

struct CpRange {
  uint32_t lo, hi;
};
#include "unicode_ranges.inc"

bool in_ranges(const CpRange* r, int n, uint32_t cp) {
  int lo = 0, hi = n - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) / 2;
    if (cp < r[mid].lo) hi = mid - 1;
    else if (cp > r[mid].hi) lo = mid + 1;
    else return true;
  }
  return false;
}
inline bool is_letter(uint32_t cp) { return in_ranges(kLetterRanges, kLetterCount, cp); }
inline bool is_number(uint32_t cp) { return in_ranges(kNumberRanges, kNumberCount, cp); }
inline bool is_space(uint32_t cp) { return in_ranges(kSpaceRanges, kSpaceCount, cp); }

// one code point of (valid) UTF-8 at s[i]; returns its length in bytes
inline int decode_utf8(const unsigned char* s, size_t n, size_t i, uint32_t* cp) {
  const unsigned char c = s[i];
  if (c < 0x80) { *cp = c; return 1; }
  if ((c >> 5) == 6 && i + 1 < n) { *cp = ((c & 0x1Fu) << 6) | (s[i + 1] & 0x3Fu); return 2; }
  if ((c >> 4) == 14 && i + 2 < n) { *cp = ((c & 0x0Fu) << 12) | ((s[i + 1] & 0x3Fu) << 6) | (s[i + 2] & 0x3Fu); return 3; }
  if ((c >> 3) == 30 && i + 3 < n) {
    *cp = ((c & 0x07u) << 18) | ((s[i + 1] & 0x3Fu) << 12) | ((s[i + 2] & 0x3Fu) << 6) | (s[i + 3] & 0x3Fu);
    return 4;
  }
  *cp = 0xFFFD;  // malformed input: treated as one symbol-class character per byte
  return 1;
}
inline void append_utf8(std::string& out, uint32_t cp) {
  if (cp < 0x80) out.push_back((char)cp);
  else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
  else if (cp < 0x10000) {
    out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F)));
  } else {
    out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
    out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F)));
  }
}

// simple case folding of the few characters the pattern's literals can meet under IGNORECASE
inline uint32_t fold(uint32_t cp) {
  if (cp >= 'A' && cp <= 'Z') return cp + 32;
  if (cp == 0x17F) return 's';   // LATIN SMALL LETTER LONG S
  if (cp == 0x212A) return 'k';  // KELVIN SIGN
  return cp;
}

struct PairHash {
  size_t operator()(const std::pair<std::string, std::string>& p) const {
    return std::hash<std::string>()(p.first) * 1000003u ^ std::hash<std::string>()(p.second);
  }
};

const char kSot[] = "<|startoftext|>";
const char kEot[] = "<|endoftext|>";
constexpr int kFullMerges = 49152 - 256 - 2;  // merge lines kept from the published file (slip.py:80)

}  // namespace

struct fc_bpe {
  int context_length = 77;
  int byte_order[256];    // position of byte b in the published byte list
  int cp_to_byte[512];    // printable code point (< 324) -> byte, -1 otherwise
  std::vector<std::string> id_token;
  std::unordered_map<std::string, int> token_id;
  std::unordered_map<std::pair<std::string, std::string>, int, PairHash> rank;
  std::unordered_map<std::string, std::vector<int>> cache;
  int sot = 0, eot = 0;

  // literal `lit` (ASCII, lower case) at s[i:], compared under simple case folding; returns bytes consumed or 0
  static size_t match_literal(const unsigned char* s, size_t n, size_t i, const char* lit) {
    size_t j = i;
    for (const char* p = lit; *p; ++p) {
      if (j >= n) return 0;
      uint32_t cp;
      const int len = decode_utf8(s, n, j, &cp);
      if (fold(cp) != (uint32_t)(unsigned char)*p) return 0;
      j += len;
    }
    return j - i;
  }

  void merge_word(const std::string& word, std::vector<int>& out) {
    auto hit = cache.find(word);
    if (hit != cache.end()) {
      out.insert(out.end(), hit->second.begin(), hit->second.end());
      return;
    }
    std::vector<std::string> sym;
    sym.reserve(word.size());
    for (size_t i = 0; i < word.size(); ++i) sym.emplace_back(1, word[i]);
    sym.back() += "</w>";
    while (sym.size() > 1) {
      int best = INT32_MAX;
      size_t at = 0;
      for (size_t k = 0; k + 1 < sym.size(); ++k) {
        auto it = rank.find({sym[k], sym[k + 1]});
        if (it != rank.end() && it->second < best) { best = it->second; at = k; }
      }
      if (best == INT32_MAX) break;
      const std::string first = sym[at], second = sym[at + 1];
      std::vector<std::string> merged;
      merged.reserve(sym.size());
      for (size_t k = 0; k < sym.size();) {
        if (k + 1 < sym.size() && sym[k] == first && sym[k + 1] == second) {
          merged.push_back(first + second);
          k += 2;
        } else {
          merged.push_back(sym[k]);
          k += 1;
        }
      }
      sym.swap(merged);
    }
    std::vector<int> ids;
    ids.reserve(sym.size());
    for (auto& s : sym) ids.push_back(token_id.at(s));
    out.insert(out.end(), ids.begin(), ids.end());
    cache.emplace(word, std::move(ids));
  }

  void encode(const char* text, std::vector<int>& out) {
    const unsigned char* s = reinterpret_cast<const unsigned char*>(text);
    const size_t n = strlen(text);
    size_t i = 0;
    static const char* kContractions[] = {"'s", "'t", "'re", "'ve", "'m", "'ll", "'d"};
    while (i < n) {
      size_t len = match_literal(s, n, i, kSot);
      if (len) { out.push_back(sot); i += len; continue; }
      len = match_literal(s, n, i, kEot);
      if (len) { out.push_back(eot); i += len; continue; }
      for (const char* c : kContractions) {
        len = match_literal(s, n, i, c);
        if (len) break;
      }
      if (!len) {
        uint32_t cp;
        int l = decode_utf8(s, n, i, &cp);
        if (is_letter(cp)) {
          size_t j = i + l;
          while (j < n) {
            const int l2 = decode_utf8(s, n, j, &cp);
            if (!is_letter(cp)) break;
            j += l2;
          }
          len = j - i;
        } else if (is_number(cp)) {
          len = l;
        } else if (is_space(cp)) {
          i += l;  // white space only separates pieces
          continue;
        } else {
          size_t j = i + l;
          while (j < n) {
            const int l2 = decode_utf8(s, n, j, &cp);
            if (is_space(cp) || is_letter(cp) || is_number(cp)) break;
            j += l2;
          }
          len = j - i;
        }
      }
      const std::string piece(text + i, len);
      if (piece == kSot) out.push_back(sot);        // the reference seeds its cache with the two specials
      else if (piece == kEot) out.push_back(eot);
      else merge_word(piece, out);
      i += len;
    }
  }
};

using fc::fail;

extern "C" {

int fc_bpe_create(const char* merges_gz_path, int32_t context_length, fc_bpe** out) {
  if (!merges_gz_path || !out || context_length < 2) return fail(FC_EINVAL, "fc_bpe_create: bad argument");
  {
    // zlib reads a file WITHOUT the gzip header transparently (as plain text): any garbage would load as a 514-entry vocabulary.
    // The merges file is a gzip member (bpe_simple_vocab_16e6.txt.gz): anything else is refused here.
    FILE* raw = fopen(merges_gz_path, "rb");
    if (!raw) return fail(FC_EINVAL, "fc_bpe_create: cannot open %s", merges_gz_path);
    unsigned char magic[2] = {0, 0};
    const size_t n = fread(magic, 1, 2, raw);
    fclose(raw);
    if (n != 2 || magic[0] != 0x1f || magic[1] != 0x8b)
      return fail(FC_EINVAL, "fc_bpe_create: %s is not a gzip file (no 1f 8b magic)", merges_gz_path);
  }
  gzFile f = gzopen(merges_gz_path, "rb");
  if (!f) return fail(FC_EINVAL, "fc_bpe_create: cannot open %s", merges_gz_path);
  std::string text;
  char buf[1 << 16];
  int got;
  while ((got = gzread(f, buf, sizeof(buf))) > 0) text.append(buf, got);
  int zerr = Z_OK;
  (void)gzerror(f, &zerr);
  gzclose(f);
  if (got < 0 || (zerr != Z_OK && zerr != Z_STREAM_END))   // (a truncated or corrupt member: never a silently shorter merge list)
    return fail(FC_EINVAL, "fc_bpe_create: %s is not a readable gzip file (zlib error %d)", merges_gz_path, zerr);

  auto* t = new fc_bpe();
  t->context_length = context_length;
  // published byte list: printable Latin-1 bytes stand for themselves, the other 68 move to U+0100.. in byte order
  std::vector<int> bytes;
  for (int b = 0x21; b <= 0x7E; ++b) bytes.push_back(b);
  for (int b = 0xA1; b <= 0xAC; ++b) bytes.push_back(b);
  for (int b = 0xAE; b <= 0xFF; ++b) bytes.push_back(b);
  std::vector<bool> printable(256, false);
  for (int b : bytes) printable[b] = true;
  std::fill(t->cp_to_byte, t->cp_to_byte + 512, -1);
  int spill = 0;
  for (int b = 0; b < 256; ++b) {
    if (printable[b]) t->cp_to_byte[b] = b;
    else { bytes.push_back(b); t->cp_to_byte[256 + spill++] = b; }
  }
  for (int i = 0; i < 256; ++i) t->byte_order[bytes[i]] = i;
  auto add = [&](const std::string& tok) {
    t->token_id[tok] = (int)t->id_token.size();  // duplicates: the later position wins, as dict(zip(vocab, range))
    t->id_token.push_back(tok);
  };
  for (int i = 0; i < 256; ++i) add(std::string(1, (char)bytes[i]));
  for (int i = 0; i < 256; ++i) add(std::string(1, (char)bytes[i]) + "</w>");
  // lines[1 : 1 + kFullMerges] of text.split('\n'); every such line is a vocabulary entry, "a b" lines are merges
  size_t pos = text.find('\n');
  pos = pos == std::string::npos ? text.size() : pos + 1;
  bool ended = pos >= text.size() && (text.empty() || text.back() != '\n');
  for (int line = 0; line < kFullMerges && !ended; ++line) {
    size_t end = text.find('\n', pos);
    if (end == std::string::npos) { end = text.size(); ended = true; }
    // str.split(): pieces separated by runs of white space (the file is ASCII-space separated; code points translated back to bytes)
    std::vector<std::string> parts;
    std::string cur;
    bool bad = false;
    const unsigned char* s = reinterpret_cast<const unsigned char*>(text.data());
    for (size_t i = pos; i < end;) {
      uint32_t cp;
      const int l = decode_utf8(s, end, i, &cp);
      i += l;
      if (is_space(cp) || cp == 0x1C || cp == 0x1D || cp == 0x1E || cp == 0x1F) {
        if (!cur.empty()) { parts.push_back(cur); cur.clear(); }
      } else if (cp < 512 && t->cp_to_byte[cp] >= 0) {
        cur.push_back((char)t->cp_to_byte[cp]);
      } else {
        bad = true;
      }
    }
    if (!cur.empty()) parts.push_back(cur);
    if (bad) { delete t; return fail(FC_EINVAL, "fc_bpe_create: line %d of %s has characters outside the byte alphabet", line + 2, merges_gz_path); }
    std::string joined;
    for (auto& p : parts) joined += p;
    if (parts.size() == 2) t->rank[{parts[0], parts[1]}] = line;
    add(joined);
    pos = end + 1;
  }
  t->sot = (int)t->id_token.size();
  add(kSot);
  t->eot = (int)t->id_token.size();
  add(kEot);
  *out = t;
  return FC_OK;
}

void fc_bpe_destroy(fc_bpe* t) { delete t; }

int32_t fc_bpe_vocab_size(const fc_bpe* t) { return t ? (int32_t)t->token_id.size() : 0; }
int32_t fc_bpe_sot(const fc_bpe* t) { return t ? t->sot : -1; }
int32_t fc_bpe_eot(const fc_bpe* t) { return t ? t->eot : -1; }

int32_t fc_bpe_encode(fc_bpe* t, const char* text_utf8, int64_t* out_ids, int32_t capacity) {
  if (!t || !text_utf8 || (!out_ids && capacity > 0)) return fail(FC_EINVAL, "fc_bpe_encode: bad argument");
  std::vector<int> ids;
  t->encode(text_utf8, ids);
  for (int i = 0; i < (int)ids.size() && i < capacity; ++i) out_ids[i] = ids[i];
  return (int32_t)ids.size();
}

int fc_bpe_tokenize(fc_bpe* t, const char* const* texts_utf8, int32_t n, int32_t truncate, int64_t* out_ids) {
  if (!t || (n > 0 && (!texts_utf8 || !out_ids)) || n < 0) return fail(FC_EINVAL, "fc_bpe_tokenize: bad argument");
  const int L = t->context_length;
  std::vector<int> ids;
  for (int i = 0; i < n; ++i) {
    ids.clear();
    ids.push_back(t->sot);
    t->encode(texts_utf8[i], ids);
    ids.push_back(t->eot);
    if ((int)ids.size() > L) {
      if (!truncate) return fail(FC_EINVAL, "fc_bpe_tokenize: text %d is too long for context length %d", i, L);
      ids.resize(L);
      ids[L - 1] = t->eot;
    }
    int64_t* row = out_ids + (size_t)i * L;
    for (int k = 0; k < L; ++k) row[k] = k < (int)ids.size() ? ids[k] : 0;
  }
  return FC_OK;
}

int32_t fc_bpe_decode(const fc_bpe* t, const int64_t* ids, int32_t n, char* out, int32_t capacity) {
  if (!t || (n > 0 && !ids) || (!out && capacity > 0)) return fail(FC_EINVAL, "fc_bpe_decode: bad argument");
  std::string bytes;
  for (int i = 0; i < n; ++i) {
    if (ids[i] < 0 || ids[i] >= (int64_t)t->id_token.size()) return fail(FC_EINVAL, "fc_bpe_decode: id %lld", (long long)ids[i]);
    bytes += t->id_token[ids[i]];
  }
  // "</w>" -> " " as the reference does after decoding (the replacement is made on the decoded text)
  std::string text;
  for (size_t i = 0; i < bytes.size();) {
    if (bytes.compare(i, 4, "</w>") == 0) { text.push_back(' '); i += 4; }
    else { text.push_back(bytes[i]); i += 1; }
  }
  if ((int)text.size() + 1 <= capacity) memcpy(out, text.c_str(), text.size() + 1);
  return (int32_t)text.size();
}

}  // extern "C"

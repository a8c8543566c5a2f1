#include "automaton.hpp"

#include <cstring>

namespace scalce {

bool Automaton::load_bin(const void *blob, size_t n) {
  // [int16 len][int32 count] + count little-endian integers of ceil(len/4) bytes whose most
  // significant 2-bit digit is the first base (reads.cpp:342-364)
  const uint8_t *p = static_cast<const uint8_t *>(blob);
  size_t pos = 0;
  patterns.clear();
  while (pos < n) {
    if (pos + 6 > n) { error = "truncated group header in core table"; return false; }
    int16_t ln; int32_t cnt;
    std::memcpy(&ln, p + pos, 2);
    std::memcpy(&cnt, p + pos + 2, 4);
    pos += 6;
    if (ln <= 0 || ln > 32 || cnt < 0) { error = "core length outside 1..32 in core table"; return false; }
    const size_t nb = (size_t(ln) + 3) / 4;
    if (pos + nb * size_t(cnt) > n) { error = "truncated group in core table"; return false; }
    for (int32_t i = 0; i < cnt; i++, pos += nb) {
      uint64_t x = 0;
      std::memcpy(&x, p + pos, nb);
      std::string s(size_t(ln), 'A');
      for (int j = 0; j < ln; j++) s[j] = "ACGT"[(x >> (2 * (ln - 1 - j))) & 3];
      patterns.push_back(std::move(s));
    }
  }
  return build();
}

bool Automaton::load_text(const char *text, size_t n) {
  patterns.clear();
  size_t i = 0;
  auto ws = [](char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; };
  while (i < n) {
    while (i < n && ws(text[i])) i++;
    size_t s = i;
    while (i < n && !ws(text[i])) i++;
    if (i > s) patterns.emplace_back(text + s, i - s);
  }
  return build();
}

bool Automaton::build() {
  struct Node { int32_t child[4]; int32_t output; int32_t level; };
  std::vector<Node> trie(1, Node{{-1, -1, -1, -1}, -1, 0});
  for (size_t p = 0; p < patterns.size(); p++) {
    int cur = 0;
    for (unsigned char ch : patterns[p]) {
      int c = base2(ch);
      if (trie[cur].child[c] < 0) {
        trie[cur].child[c] = int32_t(trie.size());
        trie.push_back(Node{{-1, -1, -1, -1}, -1, trie[cur].level + 1});
      }
      cur = trie[cur].child[c];
    }
    trie[cur].output = int32_t(p);  // a later identical core wins (reads.cpp:264)
  }
  n_states = int(trie.size());
  // BFS numbering, children visited A,C,G,T: rank == the reference's `id` (reads.cpp:275-297)
  std::vector<int32_t> order;
  order.reserve(trie.size());
  std::vector<int32_t> newid(trie.size(), -1);
  order.push_back(0);
  newid[0] = 0;
  for (size_t h = 0; h < order.size(); h++)
    for (int c = 0; c < 4; c++) {
      int32_t ch = trie[order[h]].child[c];
      if (ch >= 0) { newid[ch] = int32_t(order.size()); order.push_back(ch); }
    }
  // bucket rank = rank among core-ending states in id order = emission order of aho_output
  pattern_bucket.assign(patterns.size(), -1);
  bucket_pattern.clear();
  bucket_level.clear();
  std::vector<int32_t> state_bucket(trie.size(), -1);
  min_level = 1 << 30; max_level = 0;
  for (size_t s = 0; s < order.size(); s++) {
    const Node &nd = trie[order[s]];
    if (nd.output >= 0 && s != 0) {
      state_bucket[s] = int32_t(bucket_pattern.size());
      pattern_bucket[nd.output] = state_bucket[s];
      bucket_pattern.push_back(nd.output);
      bucket_level.push_back(nd.level);
      if (nd.level < min_level) min_level = nd.level;
      if (nd.level > max_level) max_level = nd.level;
    }
  }
  n_buckets = int(bucket_pattern.size());
  if (n_buckets == 0) min_level = 0;
  if (max_level > 127) { error = "core longer than 127 bases"; return false; }
  if (n_buckets >= int(kBucketMask)) { error = "too many cores"; return false; }
  bucket_pattern.push_back(0x3FFFFFFF);  // root bucket, dumped last (reads.cpp:491-495,161-164)
  bucket_level.push_back(0);
  // total transition function + longest-suffix core per state, rows filled in BFS order
  next.assign(size_t(n_states) * 4, 0);
  outinfo.assign(size_t(n_states), kNoOut);
  std::vector<int32_t> fail(trie.size(), 0), outst(trie.size(), -1);
  for (size_t s = 0; s < order.size(); s++) {
    const Node &nd = trie[order[s]];
    const int32_t f = fail[s];
    outst[s] = (s != 0 && nd.output >= 0) ? int32_t(s) : (s == 0 ? -1 : outst[f]);
    if (outst[s] >= 0) {
      const int32_t b = state_bucket[outst[s]];
      outinfo[s] = (uint32_t(bucket_level[b]) << kLevelShift) | uint32_t(b);
    }
    for (int c = 0; c < 4; c++) {
      const int32_t ch = nd.child[c];
      if (ch >= 0) {
        const int32_t t = newid[ch];
        next[s * 4 + c] = uint32_t(t);
        fail[t] = (s == 0) ? 0 : int32_t(next[size_t(f) * 4 + c]);
      } else {
        next[s * 4 + c] = (s == 0) ? 0u : next[size_t(f) * 4 + c];
      }
    }
  }
  return true;
}

}  // namespace scalce

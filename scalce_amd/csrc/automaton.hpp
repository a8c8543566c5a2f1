// automaton.hpp -- host-side core table: patterns.bin / text list -> BFS-ordered DFA.
// Replaces read_patterns / read_patterns_from_file / pattern_insert / prepare_aho_automata
// (/root/reference/reads.cpp:253-267, 270-324, 330-410).  The reference builds a pointer trie,
// fail links and then rewrites child[] in place; here the DFA rows are produced directly in BFS
// order so that state id == the reference's BFS `id` and shallow states (the hot ones) are the
// low ids that get staged in LDS.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace scalce {

constexpr uint32_t kNoOut = 0xFFFFFFFFu;
constexpr int kLevelShift = 25;  // outinfo = level << 25 | bucket rank
constexpr uint32_t kBucketMask = (1u << kLevelShift) - 1;

struct Automaton {
  std::vector<std::string> patterns;    // file order: the index stored in .scalcer
  std::vector<uint32_t> next;           // 4 transitions per state
  std::vector<uint32_t> outinfo;        // longest core that is a suffix of the state, or kNoOut
  std::vector<int32_t> bucket_pattern;  // bucket rank -> pattern index; last entry = root (0x3FFFFFFF)
  std::vector<int32_t> bucket_level;    // bucket rank -> core length; root 0
  std::vector<int32_t> pattern_bucket;  // pattern index -> bucket rank, -1 if shadowed by a later duplicate
  int n_states = 0;
  int n_buckets = 0;  // root excluded
  int min_level = 0, max_level = 0;
  std::string error;

  bool load_bin(const void *blob, size_t n);
  bool load_text(const char *text, size_t n);

 private:
  bool build();
};

inline int base2(unsigned char c) {  // getval / _tbl, const.cpp:47-49 (bytes outside 'A'..'z': 0)
  switch (c) {
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 0;
  }
}

}  // namespace scalce

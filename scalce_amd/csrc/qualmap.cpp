// qualmap.cpp -- host-side quality model (product code, not the oracle).
// scalce_qmap_init follows quality_mapping_init after its sampling loop
// (/root/reference/qualities.cpp:99-174): phred offset detection and, under -p, the lossy
// replacement table.  Runs once per file on a 128-bin histogram; uses double / pow() exactly
// like the reference so the table is identical on the same libm.
#include <cmath>
#include <cstring>

#include "../../include/scalce_hip.h"

namespace {
double err_of(int c, int offset) { return std::pow(10, -(c - offset) / 10.0); }
}  // namespace

extern "C" void scalce_qmap_init(scalce_qmap *q, const int32_t stat[128], int lossy_percentage) {
  q->offset = 64;
  for (int c = 33; c < 64; c++)
    if (stat[c]) { q->offset = 33; break; }
  for (int c = 0; c < 128; c++) q->values[c] = c;
  if (!lossy_percentage) return;

  bool taken[128];
  std::memset(taken, 0, sizeof taken);
  // characters with more than 30 % error collapse onto the offset (:117-122)
  for (int c = q->offset; c < 128 && 100 * err_of(c, q->offset) > 30; c++) {
    q->values[c] = q->offset;
    taken[c] = true;
  }
  // visit characters by decreasing count, ties by smaller character (:125-140)
  int rank[128];
  for (int i = 0; i < 128; i++) rank[i] = i;
  for (int i = 1; i < 128; i++) {  // insertion sort on (count desc, char asc): a total order
    int v = rank[i], j = i - 1;
    while (j >= 0 && (stat[rank[j]] < stat[v] || (stat[rank[j]] == stat[v] && rank[j] > v))) {
      rank[j + 1] = rank[j];
      j--;
    }
    rank[j + 1] = v;
  }
  const double pct = lossy_percentage / 100.0;
  for (int i = 0; i < 128 && stat[rank[i]]; i++) {
    const int c = rank[i];
    const int below = c > 0 ? stat[c - 1] : 0, above = c < 127 ? stat[c + 1] : 0;
    if (taken[c] || stat[c] < below || stat[c] < above) continue;  // only untouched local maxima
    const double er = err_of(c, q->offset);
    int left = c, right = c;
    double total = er;
    for (int k = c - 1; k >= 0; k--) {  // grow left while the mean error stays within +pct (:151-158)
      total += err_of(k, q->offset);
      if (taken[k] || total / (c - k + 1) > er + (er * pct)) { left = k + 1; break; }
    }
    total = er;
    for (int k = c + 1; k < 128; k++) {  // grow right while it stays within -pct (:160-167)
      total += err_of(k, q->offset);
      if (taken[k] || total / (k - c + 1) < er - (er * pct)) { right = k - 1; break; }
    }
    for (int k = left; k <= right; k++) { q->values[k] = c; taken[k] = true; }
  }
}

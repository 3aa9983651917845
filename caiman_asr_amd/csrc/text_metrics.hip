// Edit distance for WER / CER (include/caiman_data.h).  The reference calls a Rust extension for this
// (`levenshtein_rs.levenshtein_list`, training/caiman_asr_train/evaluate/metrics.py:20,123); host code, no device
// part: two rolling rows, O(min(n, m)) memory.
#include <algorithm>
#include <vector>

#include "../../include/caiman_data.h"
#include "common.h"

extern "C" int64_t caiman_levenshtein(const int32_t* a, int64_t n, const int32_t* b, int64_t m) {
  if ((n > 0 && !a) || (m > 0 && !b) || n < 0 || m < 0) {
    caiman::set_error("levenshtein: null / negative argument");
    return -1;
  }
  if (n > m) {
    std::swap(a, b);
    std::swap(n, m);
  }
  std::vector<int64_t> row(n + 1);
  for (int64_t j = 0; j <= n; ++j) row[j] = j;
  for (int64_t i = 1; i <= m; ++i) {
    int64_t diag = row[0];
    row[0] = i;
    for (int64_t j = 1; j <= n; ++j) {
      const int64_t up = row[j];
      const int64_t sub = diag + (a[j - 1] != b[i - 1]);
      row[j] = std::min({up + 1, row[j - 1] + 1, sub});
      diag = up;
    }
  }
  return row[n];
}

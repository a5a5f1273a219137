// Host-side sparse-grid Gauss-Hermite table (see spgh.cpp).
#pragma once
#include <cstdint>
#include <vector>

namespace gvi {

struct SparseGrid {
  int d = 0, p = 0;
  int64_t N = 0;
  std::vector<double> Z;     // [N][d] row-major, rows ascending lexicographic
  std::vector<double> w;     // [N], sums to 1, some negative
  std::vector<int8_t> idx;   // [N][d][3] (level, node index, sign)
};

// 0 on success, 1 when (d, p) is outside the tabulated rules.
int spgh_generate(int d, int p, SparseGrid& g);
int64_t spgh_count(int d, int p);

}  // namespace gvi

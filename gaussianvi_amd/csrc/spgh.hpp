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

// table_io.cpp: the reference's cereal quadrature-table file
int table_file_list(const char* path, int64_t cap, int64_t* count, double* dims, double* degs, int64_t* rows);
int table_file_read(const char* path, int d, int p, int64_t N, double* Z, double* w, int64_t* N_found);
int table_file_write(const char* path, int n_entries, const int32_t* dims, const int32_t* degs);

}  // namespace gvi

// Reader / writer for the reference's quadrature-table file (QuadratureWeightsMap serialised by a
// cereal::BinaryOutputArchive: quadrature/saveSparseGHWeightMap.h:14-51, read back at
// quadrature/SparseGaussHermite.h:80-117).  Layout restated from helpers/SerializeEigenMaps.h:195-224 and
// cereal's container rules (size tag = u64; std::tuple elements in order; binary archive = raw little-endian):
//   u64 count
//   count x { f64 dim, f64 deg,                         key   std::tuple<double,double>
//             i32 rows, i32 cols, rows*cols f64,        MatrixXd, element by element ROW-major
//             i32 len,  len f64 }                       VectorXd
// Entries are in unordered_map iteration order, i.e. unspecified: readers scan for the key.
#include <cstdio>
#include <cstring>
#include <vector>

#include "spgh.hpp"

namespace gvi {

namespace {
struct File {
  FILE* f = nullptr;
  explicit File(const char* path, const char* mode) { f = path ? fopen(path, mode) : nullptr; }
  ~File() { if (f) fclose(f); }
  template <typename T> bool get(T& v) { return fread(&v, sizeof(T), 1, f) == 1; }
  template <typename T> bool put(const T& v) { return fwrite(&v, sizeof(T), 1, f) == 1; }
};

struct EntryHeader { double dim, deg; int32_t rows, cols; };

// positions the file at the first matrix element of the next entry
bool read_header(File& io, EntryHeader& h) {
  return io.get(h.dim) && io.get(h.deg) && io.get(h.rows) && io.get(h.cols) && h.rows >= 0 && h.cols >= 0;
}
// a header whose claimed body cannot lie inside any file this reader accepts (2^40 bytes): refused before the product of
// two corrupted 31-bit fields overflows (found by the UBSan build, tests/test_sanitizers.py)
constexpr int64_t kMaxBodyBytes = (int64_t)1 << 40;
bool body_bytes(const EntryHeader& h, int64_t& bytes) {
  if ((int64_t)h.rows > kMaxBodyBytes / 8 / ((int64_t)h.cols > 0 ? (int64_t)h.cols : 1)) return false;
  bytes = (int64_t)h.rows * h.cols * 8;
  return true;
}
bool skip_body(File& io, const EntryHeader& h) {
  int64_t bytes;
  if (!body_bytes(h, bytes)) return false;
  if (fseek(io.f, (long)bytes, SEEK_CUR)) return false;
  int32_t len;
  if (!io.get(len) || len < 0) return false;
  return fseek(io.f, (long)((int64_t)len * 8), SEEK_CUR) == 0;
}
}  // namespace

// 0 ok, 1 cannot open / truncated.  keys (dim, deg, rows) of up to cap entries are returned in file order.
int table_file_list(const char* path, int64_t cap, int64_t* count, double* dims, double* degs, int64_t* rows) {
  File io(path, "rb");
  if (!io.f) return 1;
  uint64_t n;
  if (!io.get(n)) return 1;
  for (uint64_t e = 0; e < n; ++e) {
    EntryHeader h;
    if (!read_header(io, h)) return 1;
    if ((int64_t)e < cap) {
      if (dims) dims[e] = h.dim;
      if (degs) degs[e] = h.deg;
      if (rows) rows[e] = h.rows;
    }
    if (!skip_body(io, h)) return 1;
  }
  *count = (int64_t)n;
  return 0;
}

// 0 ok, 1 io error, 2 key absent, 3 N mismatch.  Z [N][d] row-major, w [N]; either may be null.
int table_file_read(const char* path, int d, int p, int64_t N, double* Z, double* w, int64_t* N_found) {
  File io(path, "rb");
  if (!io.f) return 1;
  uint64_t n;
  if (!io.get(n)) return 1;
  for (uint64_t e = 0; e < n; ++e) {
    EntryHeader h;
    if (!read_header(io, h)) return 1;
    if (h.dim != (double)d || h.deg != (double)p) {          // exact compare, like the map's key equality
      if (!skip_body(io, h)) return 1;
      continue;
    }
    if (N_found) *N_found = h.rows;
    if (!Z && !w) return 0;
    if (h.cols != d || h.rows != N) return 3;
    const size_t cnt = (size_t)h.rows * h.cols;
    if (Z) { if (fread(Z, 8, cnt, io.f) != cnt) return 1; }
    else if (fseek(io.f, (long)(cnt * 8), SEEK_CUR)) return 1;
    int32_t len;
    if (!io.get(len) || len != h.rows) return 1;
    if (w && fread(w, 8, (size_t)len, io.f) != (size_t)len) return 1;
    return 0;
  }
  return 2;
}

// Generates every (dims[e], degs[e]) with the in-tree nwspgr restatement and writes them in the given order.
// 0 ok, 1 io error, 2 a key is outside the tabulated rules.
int table_file_write(const char* path, int n_entries, const int32_t* dims, const int32_t* degs) {
  std::vector<SparseGrid> grids((size_t)n_entries);
  for (int e = 0; e < n_entries; ++e)
    if (spgh_generate(dims[e], degs[e], grids[(size_t)e])) return 2;
  File io(path, "wb");
  if (!io.f) return 1;
  bool ok = io.put((uint64_t)n_entries);
  for (int e = 0; e < n_entries && ok; ++e) {
    const SparseGrid& g = grids[(size_t)e];
    ok = io.put((double)g.d) && io.put((double)g.p) && io.put((int32_t)g.N) && io.put((int32_t)g.d);
    ok = ok && fwrite(g.Z.data(), 8, g.Z.size(), io.f) == g.Z.size();
    ok = ok && io.put((int32_t)g.N) && fwrite(g.w.data(), 8, g.w.size(), io.f) == g.w.size();
  }
  return ok ? 0 : 1;
}

}  // namespace gvi

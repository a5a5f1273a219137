// Smolyak sparse-grid Gauss-Hermite generator (host side of the product; needs no GPU).
//
// Replaces the reference's MATLAB-compiled generator + cereal table
//   quadrature/GH/SparseGH/nwspgr.m:32-134 (nwspgr), :147-169 (SpGrGetSeq), :183-190 (SpGrKronProd)
//   quadrature/generateSpGHWeights.h:23-84, quadrature/saveSparseGHWeightMap.h:14-51.
//
// Not a transcription: nodes are handled as small integer codes (signed rank of the 1-D node value
// among all tabulated values of levels 1..p), duplicates are merged in a hash map keyed by the code
// vector, reflections are enumerated per point, and the final order is an integer lexicographic
// sort -- which equals the reference's `sortrows` on doubles because rank order == value order.
// Contributions to one node are summed in the reference's generation order (q ascending,
// multi-index in SpGrGetSeq order, tensor product last dimension fastest), so merged weights agree
// with the reference's sequential sums to the last bit before the final normalisation.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "gqn_table.inc"
#include "spgh.hpp"

namespace gvi {
namespace {

struct Node1D {
  double value;
  int level, index;
};

// Distinct positive node values of levels 1..p, ascending; rank 0 is the value 0.
struct RankTable {
  std::vector<Node1D> by_rank;                 // by_rank[r] for r >= 0
  std::vector<std::vector<int>> rank_of;       // rank_of[level][index]
  explicit RankTable(int p) {
    std::vector<Node1D> pos;
    for (int l = 1; l <= p; ++l)
      for (int j = 0; j < (l + 1) / 2; ++j) {
        double v = GQN_NODE[GQN_OFF[l - 1] + j];
        if (v != 0.0) pos.push_back({v, l, j});
      }
    std::sort(pos.begin(), pos.end(), [](const Node1D& a, const Node1D& b) {
      return a.value < b.value || (a.value == b.value && a.level < b.level);
    });
    by_rank.push_back({0.0, 1, 0});
    rank_of.assign(p + 1, {});
    for (int l = 1; l <= p; ++l) rank_of[l].assign((l + 1) / 2, 0);
    for (const Node1D& nd : pos) {
      if (by_rank.back().value != nd.value) by_rank.push_back(nd);  // exact-equality merge
      rank_of[nd.level][nd.index] = (int)by_rank.size() - 1;
    }
  }
};

// product of the 1-D weights in long double (GVI_SPGH_EXTENDED)
long double prod_ext(const std::vector<int>& lv, const std::vector<int>& it, int d) {
  long double w = 1.0L;
  for (int a = 0; a < d; ++a) w *= (long double)GQN_WEIGHT[GQN_OFF[lv[a] - 1] + it[a]];
  return w;
}

double binom(int n, int k) {
  if (k < 0 || k > n) return 0.0;
  double r = 1.0;
  for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return std::round(r);
}

// All compositions of `total` into d parts >= 1, first part descending (SpGrGetSeq order).
void compositions(int d, int total, std::vector<int>& cur, int pos,
                  std::vector<std::vector<int>>& out) {
  if (pos == d - 1) {
    cur[pos] = total;
    out.push_back(cur);
    return;
  }
  for (int v = total - (d - 1 - pos); v >= 1; --v) {
    cur[pos] = v;
    compositions(d, total - v, cur, pos + 1, out);
  }
}

}  // namespace

int spgh_generate(int d, int p, SparseGrid& g) {
  if (d < 1 || p < 1 || p > GQN_MAX_LEVEL || d > 64) return 1;
  RankTable rt(p);
  if (rt.by_rank.size() > 32000) return 1;
  const int minq = std::max(0, p - d), maxq = p - 1;

  // positive orthant: code vector -> slot; weights accumulate in generation order
  std::unordered_map<std::u16string, int64_t> slot;
  std::vector<std::u16string> keys;
  // Merged weights.  The reference sums the contributions of one node in double, in generation order (nwspgr.m:88-103);
  // `wpos` reproduces those sums bit for bit and is what every key of the reference's table file gets
  // (quadrature/saveSparseGHWeightMap.h:16-23: dims 1..13 up to degree {25,25,19,13,11,9,8,7,7,7,6,6,6}, dims 14..20 up to
  // 5) -- parity with the reference's table comes first.  The contributions alternate in sign with binomial magnitudes,
  // and OUTSIDE that key range the double sums degrade visibly: at (24,7) they lose ~7 digits and a quadratic psi is
  // integrated to 1e-5 only (measured A/B, tests/test_gpu_parity.py::test_c5_full_table_*: 1.0e-5 with the double sums,
  // 4e-8 with the sums below; the summation order on the device plays no role).  Keys the reference never tabulated are
  // therefore merged and normalised in long double and rounded once (`wext`).  GVI_SPGH_EXTENDED=0 / 1 forces either.
  static const int kRefMaxDeg[13] = {25, 25, 19, 13, 11, 9, 8, 7, 7, 7, 6, 6, 6};
  const bool ref_key = (d <= 13 && p <= kRefMaxDeg[d - 1]) || (d >= 14 && d <= 20 && p <= 5);
  const char* ext_env = std::getenv("GVI_SPGH_EXTENDED");
  const bool extended = ext_env ? std::atoi(ext_env) != 0 : !ref_key;
  std::vector<double> wpos;
  std::vector<long double> wext;
  std::u16string key(d, u'\0');
  std::vector<int> cnt(d), it(d);
  for (int q = minq; q <= maxq; ++q) {
    const double bq = (((maxq - q) & 1) ? -1.0 : 1.0) * binom(d - 1, d + q - p);
    std::vector<std::vector<int>> seqs;
    std::vector<int> cur(d);
    compositions(d, d + q, cur, 0, seqs);
    for (const std::vector<int>& lv : seqs) {
      for (int a = 0; a < d; ++a) { cnt[a] = (lv[a] + 1) / 2; it[a] = 0; }
      while (true) {
        double w = GQN_WEIGHT[GQN_OFF[lv[0] - 1] + it[0]];   // kron(weights, w1D{j}) left to right
        for (int a = 1; a < d; ++a) w = w * GQN_WEIGHT[GQN_OFF[lv[a] - 1] + it[a]];
        for (int a = 0; a < d; ++a) key[a] = (char16_t)rt.rank_of[lv[a]][it[a]];
        auto f = slot.find(key);
        if (f == slot.end()) {
          slot.emplace(key, (int64_t)keys.size());
          keys.push_back(key);
          wpos.push_back(bq * w);
          if (extended) wext.push_back((long double)bq * prod_ext(lv, it, d));
        } else {
          wpos[f->second] += bq * w;
          if (extended) wext[f->second] += (long double)bq * prod_ext(lv, it, d);
        }
        int a = d - 1;                                        // last dimension fastest
        while (a >= 0 && ++it[a] == cnt[a]) it[a--] = 0;
        if (a < 0) break;
      }
    }
  }

  // reflect every non-zero coordinate; total count first
  int64_t N = 0;
  for (const std::u16string& k : keys) {
    int nz = 0;
    for (int a = 0; a < d; ++a) nz += k[a] != 0;
    N += (int64_t)1 << nz;
  }
  std::vector<int16_t> codes((size_t)N * d);
  std::vector<double> w(N);
  std::vector<long double> wl(extended ? N : 0);
  int64_t r = 0;
  std::vector<int> nzpos;
  for (size_t s = 0; s < keys.size(); ++s) {
    const std::u16string& k = keys[s];
    nzpos.clear();
    for (int a = 0; a < d; ++a) if (k[a] != 0) nzpos.push_back(a);
    const int64_t combos = (int64_t)1 << nzpos.size();
    for (int64_t m = 0; m < combos; ++m, ++r) {
      int16_t* row = &codes[(size_t)r * d];
      for (int a = 0; a < d; ++a) row[a] = (int16_t)k[a];
      for (size_t b = 0; b < nzpos.size(); ++b)
        if (m >> b & 1) row[nzpos[b]] = (int16_t)-row[nzpos[b]];
      w[r] = wpos[s];
      if (extended) wl[r] = wext[s];
    }
  }

  // lexicographic order on signed codes == sortrows on the double values
  std::vector<int64_t> order(N);
  for (int64_t i = 0; i < N; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) {
    const int16_t* a = &codes[(size_t)x * d];
    const int16_t* b = &codes[(size_t)y * d];
    for (int c = 0; c < d; ++c)
      if (a[c] != b[c]) return a[c] < b[c];
    return false;
  });

  g.d = d; g.p = p; g.N = N;
  g.Z.resize((size_t)N * d);
  g.w.resize(N);
  g.idx.resize((size_t)N * d * 3);
  double total = 0.0;
  for (int64_t i = 0; i < N; ++i) total += w[order[i]];
  long double total_ext = 0.0L;
  if (extended) for (int64_t i = 0; i < N; ++i) total_ext += wl[order[i]];
  for (int64_t i = 0; i < N; ++i) {
    const int16_t* row = &codes[(size_t)order[i] * d];
    g.w[i] = extended ? (double)(wl[order[i]] / total_ext) : w[order[i]] / total;
    for (int a = 0; a < d; ++a) {
      const int c = row[a], rk = c < 0 ? -c : c;
      const Node1D& nd = rt.by_rank[rk];
      g.Z[(size_t)i * d + a] = c < 0 ? -nd.value : nd.value;
      int8_t* o = &g.idx[((size_t)i * d + a) * 3];
      o[0] = (int8_t)nd.level; o[1] = (int8_t)nd.index; o[2] = (int8_t)(c > 0) - (int8_t)(c < 0);
    }
  }
  return 0;
}

int64_t spgh_count(int d, int p) {
  // Count without materialising the reflected grid.
  if (d < 1 || p < 1 || p > GQN_MAX_LEVEL || d > 64) return -1;
  RankTable rt(p);
  const int minq = std::max(0, p - d), maxq = p - 1;
  std::unordered_map<std::u16string, char> seen;
  std::u16string key(d, u'\0');
  std::vector<int> cnt(d), it(d);
  int64_t N = 0;
  for (int q = minq; q <= maxq; ++q) {
    std::vector<std::vector<int>> seqs;
    std::vector<int> cur(d);
    compositions(d, d + q, cur, 0, seqs);
    for (const std::vector<int>& lv : seqs) {
      for (int a = 0; a < d; ++a) { cnt[a] = (lv[a] + 1) / 2; it[a] = 0; }
      while (true) {
        int nz = 0;
        for (int a = 0; a < d; ++a) { key[a] = (char16_t)rt.rank_of[lv[a]][it[a]]; nz += key[a] != 0; }
        if (seen.emplace(key, 1).second) N += (int64_t)1 << nz;
        int a = d - 1;
        while (a >= 0 && ++it[a] == cnt[a]) it[a--] = 0;
        if (a < 0) break;
      }
    }
  }
  return N;
}

}  // namespace gvi

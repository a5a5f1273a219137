// Sign-orbit decomposition of a symmetric sparse Gauss-Hermite table (host side).
//
// nwspgr('GQN', d, k, sym = 1) builds the grid in the positive orthant and reflects every non-zero coordinate, copying the
// weight (quadrature/GH/SparseGH/nwspgr.m:108-126).  The table is therefore a disjoint union of SIGN ORBITS: a support
// c_0 < ... < c_{s-1} (s <= k - 1 non-zero coordinates), magnitudes m_0..m_{s-1} > 0, one weight, and all 2^s sign
// patterns.  At (12,5): 17 217 points = 1 975 orbits + the origin.
// moments_orbit_kernel (kernels_orbit.hpp) walks an orbit per lane instead of a point per lane.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace gvi {

constexpr int ORBIT_SMAX = 6;          // largest support the kernel is instantiated for (degree k <= 7)

struct OrbitHost {
  bool ok = false;
  int smax = 0;
  double w0 = 0.0;                                // weight of the origin (0 if the table has none)
  int64_t count[ORBIT_SMAX + 1] = {0};            // orbits per support size
  // orbits sorted by support size DESCENDING, each size class padded to a multiple of 64 with zero-weight orbits
  int64_t norb_p = 0;
  std::vector<uint64_t> cpk;                      // [norb_p] coordinates, one byte each (c_0 in the low byte)
  // [norb_p] ten bits per coordinate: R_i = 1 + d + c_i d - c_i (c_i - 1) / 2 - c_i, the index of (c_i, 0) in the packed
  // accumulator layout [m0 | m1[d] | upper triangle by rows] -- entry (c_i, c_j), i <= j, is R_i + c_j
  std::vector<uint64_t> rpk;
  std::vector<double> mag;                        // [smax][norb_p]
  std::vector<double> w;                          // [norb_p]
  std::vector<int32_t> tile_s, tile_first;        // per 64-orbit tile: support size, first orbit index
};

// Z [N][d] row-major, w [N].  verify: check that every point of the table belongs to an orbit whose representative (all
// non-zero coordinates positive) carries the same weight -- needed for caller-supplied tables; the in-tree generator
// constructs the grid by reflection.
inline OrbitHost build_orbits(int d, int64_t N, const double* Z, const double* w, bool verify) {
  OrbitHost o;
  if (d > 32) return o;                                         // (rpk: ten bits per row base)
  struct Rep { int64_t row; int s; };
  std::vector<Rep> reps;
  int64_t covered = 0;
  bool have_origin = false;
  for (int64_t i = 0; i < N; ++i) {
    const double* z = Z + (size_t)i * d;
    int s = 0;
    bool pos = true;
    for (int a = 0; a < d; ++a) {
      if (z[a] != 0.0) { ++s; pos = pos && z[a] > 0.0; }
    }
    if (!pos) continue;
    if (s > ORBIT_SMAX) return o;
    if (s == 0) { have_origin = true; o.w0 = w[i]; covered += 1; continue; }
    reps.push_back({i, s});
    covered += (int64_t)1 << s;
  }
  if (covered != N) return o;                                   // some sign pattern is missing (or duplicated)
  if (verify) {
    // key = |z| bit patterns of the row; every point must find its representative with an identical weight
    auto hash_row = [&](const double* z) {
      uint64_t h = 1469598103934665603ull;
      for (int a = 0; a < d; ++a) {
        double v = z[a] < 0 ? -z[a] : z[a];
        if (v == 0.0) v = 0.0;                                    // -0.0 -> +0.0
        uint64_t b;
        std::memcpy(&b, &v, 8);
        h = (h ^ b) * 1099511628211ull;
      }
      return h;
    };
    std::unordered_map<uint64_t, int64_t> by_key;
    by_key.reserve(reps.size() * 2);
    for (const Rep& r : reps) by_key.emplace(hash_row(Z + (size_t)r.row * d), r.row);
    for (int64_t i = 0; i < N; ++i) {
      const double* z = Z + (size_t)i * d;
      bool zero = true;
      for (int a = 0; a < d; ++a) zero = zero && z[a] == 0.0;
      if (zero) continue;
      auto it = by_key.find(hash_row(z));
      if (it == by_key.end()) return o;
      const double* zr = Z + (size_t)it->second * d;
      for (int a = 0; a < d; ++a)
        if ((z[a] < 0 ? -z[a] : z[a]) != zr[a]) return o;
      if (w[i] != w[it->second]) return o;
    }
  }
  (void)have_origin;
  std::stable_sort(reps.begin(), reps.end(), [](const Rep& a, const Rep& b) { return a.s > b.s; });
  for (const Rep& r : reps) { o.count[r.s]++; o.smax = std::max(o.smax, r.s); }
  if (o.smax == 0) { o.ok = true; return o; }                   // only the origin (degree 1)
  o.norb_p = 0;
  for (int s = o.smax; s >= 1; --s) o.norb_p += (o.count[s] + 63) / 64 * 64;
  o.cpk.assign(o.norb_p, 0);
  o.rpk.assign(o.norb_p, 0);
  o.mag.assign((size_t)o.smax * o.norb_p, 0.0);
  o.w.assign(o.norb_p, 0.0);
  int64_t pos = 0;
  size_t ri = 0;
  for (int s = o.smax; s >= 1; --s) {
    const int64_t cnt = o.count[s], padded = (cnt + 63) / 64 * 64;
    for (int64_t t = 0; t < padded / 64; ++t) { o.tile_s.push_back(s); o.tile_first.push_back((int32_t)(pos + t * 64)); }
    // Strided order inside a size class: the table is in lexicographic order, where 64 consecutive orbits share their
    // leading coordinates and all 64 lanes of a tile would add into the same accumulator entries (one LDS atomic
    // instruction then takes ~3 cycles PER LANE, tools/ubench/lds_atomic.hip).  Position q takes orbit (q * P) mod cnt
    // with P ~ cnt / 64 coprime to cnt, so every tile samples the whole class.
    int64_t P = 1;
    if (cnt > 64) {
      P = cnt / 64 + 1;
      auto gcd = [](int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; };
      while (gcd(P, cnt) != 1) ++P;
    }
    const size_t class_first = ri;
    ri += (size_t)cnt;
    for (int64_t q = 0; q < cnt; ++q) {
      const Rep& rep = reps[class_first + (size_t)((q * P) % cnt)];
      const double* z = Z + (size_t)rep.row * d;
      uint64_t pk = 0, rk = 0;
      int j = 0;
      for (int a = 0; a < d; ++a)
        if (z[a] != 0.0) {
          pk |= (uint64_t)a << (8 * j);
          const uint64_t R = (uint64_t)(1 + d + a * d - a * (a - 1) / 2 - a);
          rk |= R << (j < 3 ? 10 * j : 32 + 10 * (j - 3));
          o.mag[(size_t)j * o.norb_p + pos + q] = z[a];
          ++j;
        }
      o.cpk[pos + q] = pk;
      o.rpk[pos + q] = rk;
      o.w[pos + q] = w[rep.row];
    }
    pos += padded;
  }
  if (o.norb_p >= ((int64_t)1 << 28)) return OrbitHost();      // the kernel addresses a record by a 32-bit byte offset
  o.ok = true;
  return o;
}

// contiguous split of the tiles into `nchunk` chunks of about equal cost (half-orbit points + fixed per-orbit work);
// bounds [nchunk + 1] are tile indices
inline std::vector<int32_t> orbit_chunk_bounds(const OrbitHost& o, int nchunk) {
  const int nt = (int)o.tile_s.size();
  std::vector<double> cum(nt + 1, 0.0);
  for (int t = 0; t < nt; ++t) cum[t + 1] = cum[t] + 27.0 * (double)(1 << (o.tile_s[t] - 1)) + 150.0;
  std::vector<int32_t> b(nchunk + 1, nt);
  b[0] = 0;
  int t = 0;
  for (int c = 1; c < nchunk; ++c) {
    const double target = cum[nt] * c / nchunk;
    while (t < nt && cum[t + 1] <= target) ++t;
    b[c] = t;
  }
  return b;
}

}  // namespace gvi

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
  // per 64-lane tile: support size, index of lane 0's first orbit, orbits per lane (G) and the index stride between a
  // lane's consecutive orbits.  SUPPORT-MAJOR classes (G > 1): a lane owns G orbits of ONE support -- they differ in
  // magnitudes and weight only (at (12,5): 4 / 8 / 6 orbits per support of size 3 / 2 / 1) -- walks them one after the other
  // with the support's columns of H in registers and adds their sum to the accumulators ONCE: the LDS atomics, which bound
  // the walk once its integer overhead is gone, drop from 17.6 k to 10.6 k entries per factor.  Orbit g of the lane at
  // slot q is entry tile_first + lane + g * tile_gstride (cpk / rpk are repeated in every plane).
  std::vector<int32_t> tile_s, tile_first, tile_g, tile_gstride;
  int32_t cbase[ORBIT_SMAX + 2] = {0}, cgrp[ORBIT_SMAX + 2] = {0}, cstride[ORBIT_SMAX + 2] = {0};   // per class: first entry, G, stride
};

// cost of one tile in shader cycles of an unimpeded wave (walk stamps of the timing build, m = 6: 3100 per s = 4 tile,
// 5800 / 3800 / 2400 per s = 3 / 2 / 1 tile of 4 / 4 / 6 orbits per lane): what the split of a support's orbits over
// several lanes and the chunk bounds balance
inline double orbit_tile_cost(int s, int g) {
  static const double walk[ORBIT_SMAX + 1] = {0, 300, 800, 1300, 2500, 5200, 10800};
  return g * walk[s] + 600.0;
}

// Z [N][d] row-major, w [N].  verify: check that every point of the table belongs to an orbit whose representative (all
// non-zero coordinates positive) carries the same weight -- needed for caller-supplied tables; the in-tree generator
// constructs the grid by reflection.
// group_smax: largest support size stored support-major (0: one orbit per lane everywhere)
inline OrbitHost build_orbits(int d, int64_t N, const double* Z, const double* w, bool verify, int group_smax = 3) {
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
  auto support_key = [&](const Rep& r) {
    uint64_t pk = 0;
    int j = 0;
    const double* z = Z + (size_t)r.row * d;
    for (int a = 0; a < d; ++a)
      if (z[a] != 0.0) { pk |= (uint64_t)a << (8 * j); ++j; }
    return pk;
  };
  // slots of every class: a slot = the orbits one lane walks (all of one support)
  struct ClassPlan { int g = 1; std::vector<std::vector<int64_t>> slots; };     // slot -> rep indices
  std::vector<ClassPlan> plan(o.smax + 1);
  {
    size_t ri = 0;
    for (int s = o.smax; s >= 1; --s) {
      const size_t cnt = (size_t)o.count[s];
      ClassPlan& cp = plan[s];
      std::vector<std::vector<int64_t>> groups;
      bool uniform = false;
      if (s <= group_smax && cnt > 0) {
        std::unordered_map<uint64_t, size_t> where;
        for (size_t i = 0; i < cnt; ++i) {
          const uint64_t key = support_key(reps[ri + i]);
          auto it = where.find(key);
          if (it == where.end()) { where.emplace(key, groups.size()); groups.emplace_back(); groups.back().push_back((int64_t)(ri + i)); }
          else groups[it->second].push_back((int64_t)(ri + i));
        }
        uniform = true;
        for (const auto& gr : groups) uniform = uniform && gr.size() == groups[0].size();
      }
      if (uniform && groups[0].size() > 1) {
        const int G = (int)groups[0].size();
        int best_f = 1;
        double best = 1e300;
        for (int f = 1; f <= G; ++f) {
          if (G % f) continue;
          const double tiles = (double)((groups.size() * (size_t)f + 63) / 64);
          const double cost = tiles * orbit_tile_cost(s, G / f);
          if (cost < best) { best = cost; best_f = f; }
        }
        cp.g = G / best_f;
        for (const auto& gr : groups)
          for (int part = 0; part < best_f; ++part) cp.slots.emplace_back(gr.begin() + (size_t)part * cp.g, gr.begin() + (size_t)(part + 1) * cp.g);
      } else {
        cp.g = 1;
        for (size_t i = 0; i < cnt; ++i) cp.slots.push_back({(int64_t)(ri + i)});
      }
      ri += cnt;
    }
  }
  o.norb_p = 0;
  for (int s = o.smax; s >= 1; --s) o.norb_p += (int64_t)plan[s].g * (int64_t)((plan[s].slots.size() + 63) / 64 * 64);
  o.cpk.assign(o.norb_p, 0);
  o.rpk.assign(o.norb_p, 0);
  o.mag.assign((size_t)o.smax * o.norb_p, 0.0);
  o.w.assign(o.norb_p, 0.0);
  int64_t pos = 0;
  for (int s = o.smax; s >= 1; --s) {
    const ClassPlan& cp = plan[s];
    const int64_t cnt = (int64_t)cp.slots.size(), padded = (cnt + 63) / 64 * 64;
    o.cbase[s] = (int32_t)pos; o.cgrp[s] = cp.g; o.cstride[s] = (int32_t)padded;
    for (int64_t t = 0; t < padded / 64; ++t) {
      o.tile_s.push_back(s); o.tile_first.push_back((int32_t)(pos + t * 64));
      o.tile_g.push_back(cp.g); o.tile_gstride.push_back((int32_t)padded);
    }
    // Strided order inside a size class: the table is in lexicographic order, where 64 consecutive orbits share their
    // leading coordinates and all 64 lanes of a tile would add into the same accumulator entries (one LDS atomic
    // instruction then takes ~3 cycles PER LANE, tools/ubench/lds_atomic.hip).  Position q takes slot (q * P) mod cnt
    // with P ~ cnt / 64 coprime to cnt, so every tile samples the whole class.
    int64_t P = 1;
    if (cnt > 64) {
      P = cnt / 64 + 1;
      auto gcd = [](int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; };
      while (gcd(P, cnt) != 1) ++P;
    }
    for (int64_t q = 0; q < cnt; ++q) {
      const std::vector<int64_t>& slot = cp.slots[(size_t)((q * P) % cnt)];
      for (int g = 0; g < cp.g; ++g) {
        const Rep& rep = reps[(size_t)slot[(size_t)g]];
        const double* z = Z + (size_t)rep.row * d;
        const int64_t e = pos + (int64_t)g * padded + q;
        uint64_t pk = 0, rk = 0;
        int j = 0;
        for (int a = 0; a < d; ++a)
          if (z[a] != 0.0) {
            pk |= (uint64_t)a << (8 * j);
            const uint64_t R = (uint64_t)(1 + d + a * d - a * (a - 1) / 2 - a);
            rk |= R << (j < 3 ? 10 * j : 32 + 10 * (j - 3));
            o.mag[(size_t)j * o.norb_p + e] = z[a];
            ++j;
          }
        o.cpk[e] = pk;
        o.rpk[e] = rk;
        o.w[e] = w[rep.row];
      }
    }
    pos += (int64_t)cp.g * padded;
  }
  if (o.norb_p >= ((int64_t)1 << 28)) return OrbitHost();      // the kernel addresses a record by a 32-bit byte offset
  o.ok = true;
  return o;
}

// contiguous split of the tiles into `nchunk` chunks (bounds [nchunk + 1] are tile indices) that minimises the cost of the
// most expensive chunk: bisection on that cost, each chunk filled greedily.  (A proportional split leaves one wave of a
// four-tile table with two tiles and another with none.)
inline std::vector<int32_t> orbit_chunk_bounds(const OrbitHost& o, int nchunk) {
  const int nt = (int)o.tile_s.size();
  std::vector<double> cost(nt);
  double total = 0.0, biggest = 0.0;
  for (int t = 0; t < nt; ++t) { cost[t] = orbit_tile_cost(o.tile_s[t], o.tile_g[t]); total += cost[t]; biggest = std::max(biggest, cost[t]); }
  auto fill = [&](double cap, std::vector<int32_t>* out) {
    int t = 0, used = 0;
    if (out) out->assign((size_t)nchunk + 1, nt);
    if (out) (*out)[0] = 0;
    while (t < nt) {
      if (used == nchunk) return false;
      double sum = 0.0;
      while (t < nt && sum + cost[t] <= cap * (1.0 + 1e-12)) sum += cost[t++];
      ++used;
      if (out) (*out)[(size_t)used] = t;
    }
    return true;
  };
  double lo = std::max(biggest, total / std::max(nchunk, 1)), hi = std::max(total, biggest);
  for (int it = 0; it < 60 && hi - lo > 1e-9 * hi; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (fill(mid, nullptr)) hi = mid; else lo = mid;
  }
  std::vector<int32_t> b;
  fill(hi, &b);
  return b;
}

}  // namespace gvi

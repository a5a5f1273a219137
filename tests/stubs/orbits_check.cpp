// Host-side check of the sign-orbit decomposition (gaussianvi_amd/csrc/orbits.hpp) that moments_orbit_kernel walks:
//   1. expanding every orbit to its 2^s sign images reproduces the table (points and weights, exactly);
//   2. tiles are uniform in support size, padded orbits carry zero weight, chunk bounds cover the tiles once;
//   3. the half-orbit Gray-code walk with the +- identities (the arithmetic of kernels_orbit.hpp, restated on the host)
//      gives the same z-space moments as the plain sum over all points;
//   4. a table with a perturbed weight or a missing point is refused.
// usage: orbits_check d p [d p ...]   -- prints "ok d p N norb smax err" per table, exit code 1 on any failure
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../gaussianvi_amd/csrc/orbits.hpp"
#include "../../gaussianvi_amd/csrc/spgh.hpp"

using namespace gvi;

static int fail(const char* what, int d, int p) {
  std::printf("FAIL %s d=%d p=%d\n", what, d, p);
  return 1;
}

static int check(int d, int p) {
  SparseGrid g;
  if (spgh_generate(d, p, g)) return fail("generate", d, p);
  const int64_t N = g.N;
  OrbitHost o = build_orbits(d, N, g.Z.data(), g.w.data(), true);
  if (!o.ok) return fail("build_orbits", d, p);
  if (o.smax > std::min(d, p - 1)) return fail("smax", d, p);
  // 1. expansion
  struct Row { std::vector<double> z; double w; };
  std::vector<Row> rows;
  rows.reserve(N);
  int64_t norb = 0;
  const int nt = (int)o.tile_s.size();
  for (int t = 0; t < nt; ++t) {
    const int s = o.tile_s[t];
    if (t > 0 && o.tile_s[t] > o.tile_s[t - 1]) return fail("tile order", d, p);
    if (o.tile_g[t] < 1 || o.tile_g[t] != o.cgrp[s] || o.tile_gstride[t] != o.cstride[s]) return fail("tile group", d, p);
    for (int lane = 0; lane < 64; ++lane)
    for (int gi = 0; gi < o.tile_g[t]; ++gi) {
      const int64_t q = (int64_t)o.tile_first[t] + lane + (int64_t)gi * o.tile_gstride[t];
      if (gi > 0 && o.w[q] != 0.0 && (o.cpk[q] != o.cpk[q - (int64_t)gi * o.tile_gstride[t]] || o.rpk[q] != o.rpk[q - (int64_t)gi * o.tile_gstride[t]]))
        return fail("a lane's orbits share one support", d, p);
      if (o.w[q] == 0.0) {                       // padding
        bool zero = true;
        for (int j = 0; j < o.smax; ++j) zero = zero && o.mag[(size_t)j * o.norb_p + q] == 0.0;
        if (!zero) return fail("padding", d, p);
        continue;
      }
      ++norb;
      int c[ORBIT_SMAX];
      for (int j = 0; j < s; ++j) {
        c[j] = (int)((o.cpk[q] >> (8 * j)) & 255u);
        if (c[j] >= d || (j > 0 && c[j] <= c[j - 1])) return fail("support", d, p);
        const int R = (int)((o.rpk[q] >> (j < 3 ? 10 * j : 32 + 10 * (j - 3))) & 1023u);
        if (R != 1 + d + c[j] * d - c[j] * (c[j] - 1) / 2 - c[j]) return fail("row base", d, p);
        if (!(o.mag[(size_t)j * o.norb_p + q] > 0.0)) return fail("magnitude", d, p);
      }
      for (int j = s; j < o.smax; ++j)
        if (o.mag[(size_t)j * o.norb_p + q] != 0.0) return fail("unused magnitude", d, p);
      for (int bits = 0; bits < (1 << s); ++bits) {
        Row r{std::vector<double>(d, 0.0), o.w[q]};
        for (int j = 0; j < s; ++j) r.z[c[j]] = ((bits >> j) & 1 ? -1.0 : 1.0) * o.mag[(size_t)j * o.norb_p + q];
        rows.push_back(std::move(r));
      }
    }
  }
  if (o.w0 != 0.0) rows.push_back(Row{std::vector<double>(d, 0.0), o.w0});
  if ((int64_t)rows.size() != N) return fail("expanded count", d, p);
  std::sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.z < b.z; });
  for (int64_t i = 0; i < N; ++i) {
    for (int a = 0; a < d; ++a)
      if (rows[i].z[a] != g.Z[(size_t)i * d + a]) return fail("expanded point", d, p);
    if (rows[i].w != g.w[i]) return fail("expanded weight", d, p);
  }
  // 2. chunk bounds
  for (int nchunk : {1, 2, 3, 7, nt}) {
    if (nchunk > nt) continue;
    const std::vector<int32_t> b = orbit_chunk_bounds(o, nchunk);
    if ((int)b.size() != nchunk + 1 || b[0] != 0 || b[nchunk] != nt) return fail("bounds ends", d, p);
    for (int cidx = 0; cidx < nchunk; ++cidx)
      if (b[cidx + 1] < b[cidx]) return fail("bounds order", d, p);
  }
  // 3. the walk
  const int M = d >= 2 ? d / 2 : 1;
  std::mt19937_64 rng(17 + d * 31 + p);
  std::normal_distribution<double> nd;
  std::vector<double> H((size_t)d * M), u0(M), sg(M);
  for (auto& v : H) v = nd(rng);
  for (auto& v : u0) v = nd(rng);
  for (int r = 0; r < M; ++r) sg[r] = (r % 3 == 1) ? -1.0 : 1.0;
  const int NPAIR = (d + 1) * (d + 2) / 2;
  auto pidx = [&](int a, int b) { return 1 + d + a * d - a * (a - 1) / 2 + (b - a); };
  std::vector<long double> ref(NPAIR, 0.0L);
  for (int64_t i = 0; i < N; ++i) {
    const double* z = &g.Z[(size_t)i * d];
    long double psi = 0;
    for (int r = 0; r < M; ++r) {
      long double v = u0[r];
      for (int a = 0; a < d; ++a) v += (long double)H[(size_t)a * M + r] * z[a];
      psi += sg[r] * v * v;
    }
    const long double c = g.w[i] * psi;
    ref[0] += c;
    for (int a = 0; a < d; ++a) {
      ref[1 + a] += c * z[a];
      for (int b = a; b < d; ++b) ref[pidx(a, b)] += c * z[a] * z[b];
    }
  }
  std::vector<double> acc(NPAIR, 0.0), su0(M);
  double k0 = 0.0;
  for (int r = 0; r < M; ++r) { su0[r] = sg[r] * u0[r]; k0 = std::fma(su0[r], u0[r], k0); }
  double m0 = o.w0 * k0;
  for (int t = 0; t < nt; ++t) {
    const int S = o.tile_s[t];
    for (int lane = 0; lane < 64; ++lane)
    for (int gi0 = 0; gi0 < o.tile_g[t]; ++gi0) {
      const int64_t q = (int64_t)o.tile_first[t] + lane + (int64_t)gi0 * o.tile_gstride[t];
      int c[ORBIT_SMAX], sig[ORBIT_SMAX];
      double mg[ORBIT_SMAX];
      for (int j = 0; j < S; ++j) {
        c[j] = (int)((o.cpk[q] >> (8 * j)) & 255u);
        mg[j] = o.mag[(size_t)j * o.norb_p + q];
        sig[j] = j == S - 1 ? 1 : -1;
      }
      std::vector<double> v(M);
      for (int r = 0; r < M; ++r) {
        v[r] = mg[S - 1] * H[(size_t)c[S - 1] * M + r];
        for (int j = 0; j < S - 1; ++j) v[r] = std::fma(-mg[j], H[(size_t)c[j] * M + r], v[r]);
      }
      double E0 = 0.0, O[ORBIT_SMAX] = {0}, E[ORBIT_SMAX][ORBIT_SMAX] = {{0}};
      const int NH = 1 << (S - 1);
      for (int gi = 0; gi < NH; ++gi) {
        double qv = 0.0, l = 0.0;
        for (int r = 0; r < M; ++r) { qv = std::fma(sg[r] * v[r], v[r], qv); l = std::fma(su0[r], v[r], l); }
        const double cp = qv + k0;
        E0 += cp;
        for (int i = 0; i < S; ++i) {
          O[i] += sig[i] * l;
          for (int j = i + 1; j < S; ++j) E[i][j] += sig[i] * sig[j] * cp;
        }
        if (gi + 1 < NH) {
          const int jn = __builtin_ctz(gi + 1);
          sig[jn] = -sig[jn];
          const double t2 = (sig[jn] > 0 ? 2.0 : -2.0) * mg[jn];
          for (int r = 0; r < M; ++r) v[r] = std::fma(t2, H[(size_t)c[jn] * M + r], v[r]);
        }
      }
      const double wp = o.w[q] + o.w[q], w4 = wp + wp;
      m0 = std::fma(wp, E0, m0);
      for (int i = 0; i < S; ++i) {
        const double wm = wp * mg[i];
        acc[1 + c[i]] += w4 * mg[i] * O[i];
        acc[pidx(c[i], c[i])] += wm * mg[i] * E0;
        for (int j = i + 1; j < S; ++j) acc[pidx(c[i], c[j])] += wm * mg[j] * E[i][j];
      }
    }
  }
  acc[0] = m0;
  double err = 0.0, scale = 0.0;
  for (int e = 0; e < NPAIR; ++e) { scale = std::max(scale, (double)fabsl(ref[e])); }
  for (int e = 0; e < NPAIR; ++e) err = std::max(err, (double)fabsl(acc[e] - ref[e]) / scale);
  if (!(err < 1e-10)) { std::printf("walk error %.3e\n", err); return fail("walk", d, p); }
  // 4. refusals
  if (N > 3) {
    std::vector<double> w2 = g.w;
    w2[N - 1] *= 1.0 + 1e-12;                     // last row: (+, .., +)-most point of some orbit; its images keep the old weight
    if (build_orbits(d, N, g.Z.data(), w2.data(), true).ok) return fail("perturbed weight accepted", d, p);
    if (build_orbits(d, N - 1, g.Z.data(), g.w.data(), true).ok) return fail("missing point accepted", d, p);
    std::vector<double> Z2 = g.Z;
    for (int a = 0; a < d; ++a)
      if (Z2[a] != 0.0) { Z2[a] *= 1.0 + 1e-13; break; }
    if (build_orbits(d, N, Z2.data(), g.w.data(), true).ok) return fail("perturbed node accepted", d, p);
  }
  std::printf("ok %d %d %lld %lld %d %.3e\n", d, p, (long long)N, (long long)norb, o.smax, err);
  return 0;
}

int main(int argc, char** argv) {
  int bad = 0;
  for (int i = 1; i + 1 < argc; i += 2) bad |= check(std::atoi(argv[i]), std::atoi(argv[i + 1]));
  return bad;
}

// Host unit test of the shape-general line-FFT engine (csrc/fft_mixed.h): the
// planner and the stage arithmetic (mix_butterfly, float-reciprocal index
// math, padded LDS layout) against a float64 DFT, forward and inverse, for
// every kind of size the planner serves.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../tike_amd/csrc/fft_mixed.h"

template <bool INV>
static void stage(int R, const cf* a, cf* b, const cf* tw, int n, int Ns) {
  const int nb = n / R, step = n / (Ns * R);
  const float rcp_ns = 1.0f / (float)Ns;
  for (int jj = 0; jj < nb; ++jj) switch (R) {
      case 1: mix_butterfly<1, INV>(a, b, tw, nb, Ns, step, rcp_ns, jj); break;
      case 2: mix_butterfly<2, INV>(a, b, tw, nb, Ns, step, rcp_ns, jj); break;
      case 3: mix_butterfly<3, INV>(a, b, tw, nb, Ns, step, rcp_ns, jj); break;
      case 4: mix_butterfly<4, INV>(a, b, tw, nb, Ns, step, rcp_ns, jj); break;
      case 5: mix_butterfly<5, INV>(a, b, tw, nb, Ns, step, rcp_ns, jj); break;
      case 6: mix_butterfly<6, INV>(a, b, tw, nb, Ns, step, rcp_ns, jj); break;
      case 10: mix_butterfly<10, INV>(a, b, tw, nb, Ns, step, rcp_ns, jj); break;
      case 12: mix_butterfly<12, INV>(a, b, tw, nb, Ns, step, rcp_ns, jj); break;
      case 20: mix_butterfly<20, INV>(a, b, tw, nb, Ns, step, rcp_ns, jj); break;
      case 24: mix_butterfly<24, INV>(a, b, tw, nb, Ns, step, rcp_ns, jj); break;
      case 7: mix_butterfly<7, INV>(a, b, tw, nb, Ns, step, rcp_ns, jj); break;
      case 8: mix_butterfly<8, INV>(a, b, tw, nb, Ns, step, rcp_ns, jj); break;
      case 11: mix_butterfly<11, INV>(a, b, tw, nb, Ns, step, rcp_ns, jj); break;
      case 13: mix_butterfly<13, INV>(a, b, tw, nb, Ns, step, rcp_ns, jj); break;
      case 16: mix_butterfly<16, INV>(a, b, tw, nb, Ns, step, rcp_ns, jj); break;
      default: abort();
    }
}

template <bool INV>
static double check(int n) {
  MixPlan p;
  if (!mix_make_plan(n, &p)) return 1e9;
  int prod = 1;
  for (int s = 0; s < p.nst; ++s) prod *= p.radix[s];
  if (prod != n) return 1e9;
  std::vector<cf> a(p.ls + 8), b(p.ls + 8), tw(n);
  std::vector<std::complex<double>> x(n);
  for (int k = 0; k < n; ++k)
    tw[k] = mk((float)std::cos(-2.0 * M_PI * k / n), (float)std::sin(-2.0 * M_PI * k / n));
  for (int i = 0; i < n; ++i) {
    float re = (float)rand() / RAND_MAX - 0.5f, im = (float)rand() / RAND_MAX - 0.5f;
    a[mix_pad(i)] = mk(re, im);
    x[i] = {re, im};
  }
  cf *pa = a.data(), *pb = b.data();
  int Ns = 1;
  for (int s = 0; s < p.nst; ++s) {
    stage<INV>(p.radix[s], pa, pb, tw.data(), n, Ns);
    Ns *= p.radix[s];
    cf* t = pa;
    pa = pb;
    pb = t;
  }
  double err = 0, nrm = 0;
  for (int k = 0; k < n; ++k) {
    std::complex<double> s = 0;
    for (int j = 0; j < n; ++j)
      s += x[j] * std::polar(1.0, (INV ? 2.0 : -2.0) * M_PI * (double)((long)j * k % n) / n);
    const cf v = pa[mix_pad(k)];
    err += std::norm(s - std::complex<double>(v.x, v.y));
    nrm += std::norm(s);
  }
  return std::sqrt(err / nrm);
}

int main() {
  const int sizes[] = {1,   2,   3,   5,   6,   7,   9,   12,  15,  24,   45,   49,   96,  121,
                       160, 169, 192, 320, 384, 640, 768, 1000, 1536, 2048, 2187, 3072, 4096,
                       11 * 13 * 7, 32, 64, 128, 256, 1024, 675, 3125};
  double worst = 0;
  for (int n : {96, 192, 384, 768, 320, 640, 1000, 2048, 4096, 1536}) {
    MixPlan q;
    mix_make_plan(n, &q);
    printf("plan %d:", n);
    for (int s = 0; s < q.nst; ++s) printf(" %d", q.radix[s]);
    printf("\n");
  }
  for (int n : sizes) {
    const double e = std::fmax(check<false>(n), check<true>(n));
    if (e > 1.5e-6) printf("n %d: normwise error %.3e\n", n, e);
    worst = std::fmax(worst, e);
  }
  MixPlan p;
  const bool refused = !mix_make_plan(127, &p) && !mix_make_plan(34, &p) &&
                       !mix_make_plan(4097, &p) && !mix_make_plan(0, &p);
  printf("max normwise err %.3e, refusals %s\n", worst, refused ? "ok" : "WRONG");
  return worst < 1.5e-6 && refused ? 0 : 1;
}

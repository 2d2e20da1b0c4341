// Host unit test of the in-register DFT butterflies against a direct O(R^2) DFT.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>

#include "../../tike_amd/csrc/fft_radix.h"

template <int R, bool INV>
static double check() {
  cf v[R];
  std::complex<double> x[R];
  for (int i = 0; i < R; ++i) {
    float a = (float)rand() / RAND_MAX - 0.5f, b = (float)rand() / RAND_MAX - 0.5f;
    v[i] = mk(a, b);
    x[i] = {a, b};
  }
  Dft<R, INV>::run(v);
  double err = 0;
  for (int k = 0; k < R; ++k) {
    std::complex<double> s = 0;
    for (int n = 0; n < R; ++n)
      s += x[n] * std::polar(1.0, (INV ? 2.0 : -2.0) * M_PI * n * k / R);
    err = std::fmax(err, std::abs(s - std::complex<double>(v[k].x, v[k].y)));
  }
  return err;
}

int main() {
  double e = 0;
  e = std::fmax(e, check<2, false>());
  e = std::fmax(e, check<2, true>());
  e = std::fmax(e, check<4, false>());
  e = std::fmax(e, check<4, true>());
  e = std::fmax(e, check<8, false>());
  e = std::fmax(e, check<8, true>());
  e = std::fmax(e, check<16, false>());
  e = std::fmax(e, check<16, true>());
  e = std::fmax(e, check<3, false>());
  e = std::fmax(e, check<3, true>());
  e = std::fmax(e, check<5, false>());
  e = std::fmax(e, check<5, true>());
  e = std::fmax(e, check<7, false>());
  e = std::fmax(e, check<7, true>());
  e = std::fmax(e, check<11, false>());
  e = std::fmax(e, check<11, true>());
  e = std::fmax(e, check<13, false>());
  e = std::fmax(e, check<13, true>());
  e = std::fmax(e, check<6, false>());
  e = std::fmax(e, check<6, true>());
  e = std::fmax(e, check<10, false>());
  e = std::fmax(e, check<10, true>());
  e = std::fmax(e, check<12, false>());
  e = std::fmax(e, check<12, true>());
  e = std::fmax(e, check<20, false>());
  e = std::fmax(e, check<20, true>());
  e = std::fmax(e, check<24, false>());
  e = std::fmax(e, check<24, true>());
  e = std::fmax(e, check<32, false>());
  e = std::fmax(e, check<32, true>());
  printf("max err %.3e\n", e);
  return e < 5e-6 ? 0 : 1;
}

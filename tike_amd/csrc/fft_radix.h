// In-register radix-2/4/8/16 DFT butterflies (complex float32).
//
// Natural-order input -> natural-order output.  INV selects the sign of the
// exponent: false = exp(-2*pi*i*nk/R) (forward), true = exp(+...) (inverse,
// unscaled).  Plain C++ so the same code is unit-tested on the host
// (tests/csrc/test_fft_radix.cpp) and compiled for gfx950.
#pragma once

#if defined(__HIPCC__)
#define TK_HD __host__ __device__ __forceinline__
#else
#define TK_HD inline
#endif

struct alignas(8) cf {
  float x, y;
};

TK_HD cf mk(float x, float y) {
  cf r;
  r.x = x;
  r.y = y;
  return r;
}
TK_HD cf operator+(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
TK_HD cf operator-(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
#if defined(__HIP_DEVICE_COMPILE__)
// Device: written on 2-vectors so that the compiler selects the packed fp32
// pair v_pk_mul_f32 + v_pk_fma_f32 (half swap and sign in the instruction's
// op_sel / neg modifiers) instead of four scalar operations plus moves.
typedef float tk_v2f __attribute__((ext_vector_type(2)));
TK_HD cf operator*(cf a, cf b) {
  const tk_v2f A = __builtin_bit_cast(tk_v2f, a), B = __builtin_bit_cast(tk_v2f, b);
  const tk_v2f t = A * B.xx;
  const tk_v2f nB = {-B.y, B.y};
  return __builtin_bit_cast(cf, __builtin_elementwise_fma(A.yx, nB, t));
}
#else
TK_HD cf operator*(cf a, cf b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
#endif
TK_HD cf operator*(cf a, float s) { return mk(a.x * s, a.y * s); }
TK_HD cf conjf(cf a) { return mk(a.x, -a.y); }
TK_HD float norm2(cf a) { return a.x * a.x + a.y * a.y; }

// multiply by -i (forward) or +i (inverse)
template <bool INV>
TK_HD cf mul_mi(cf a) {
  return INV ? mk(-a.y, a.x) : mk(a.y, -a.x);
}
// multiply by a unit twiddle given as its FORWARD value (c - i s form stored as (c, -s))
template <bool INV>
TK_HD cf mul_tw(cf a, cf w) {
  return INV ? a * conjf(w) : a * w;
}

// rp = t + (-+i) d,  rm = t - (-+i) d   (the radix-4 cross terms).
// Device: two packed FMAs on the half-swapped d with a (+-1, -+1) sign pair --
// exact (a product with +-1 is exact, so this IS the sum), and without the
// per-half moves the compiler needs to recombine `t + mk(d.y, -d.x)` from two
// packed adds whose halves carry different signs.
template <bool INV>
TK_HD void addsub_mi(cf t, cf d, cf& rp, cf& rm) {
#if defined(__HIP_DEVICE_COMPILE__)
  const tk_v2f T = __builtin_bit_cast(tk_v2f, t), D = __builtin_bit_cast(tk_v2f, d);
  const tk_v2f sp = INV ? tk_v2f{-1.0f, 1.0f} : tk_v2f{1.0f, -1.0f};
  const tk_v2f sm = INV ? tk_v2f{1.0f, -1.0f} : tk_v2f{-1.0f, 1.0f};
  rp = __builtin_bit_cast(cf, __builtin_elementwise_fma(D.yx, sp, T));
  rm = __builtin_bit_cast(cf, __builtin_elementwise_fma(D.yx, sm, T));
#else
  const cf r = mul_mi<INV>(d);
  rp = t + r;
  rm = t - r;
#endif
}

template <int R, bool INV>
struct Dft;

template <bool INV>
struct Dft<2, INV> {
  static TK_HD void run(cf* v) {
    cf t = v[0] - v[1];
    v[0] = v[0] + v[1];
    v[1] = t;
  }
};

template <bool INV>
struct Dft<4, INV> {
  static TK_HD void run(cf* v) {
    cf t0 = v[0] + v[2], t1 = v[0] - v[2];
    cf t2 = v[1] + v[3], d = v[1] - v[3];
    v[0] = t0 + t2;
    v[2] = t0 - t2;
    addsub_mi<INV>(t1, d, v[1], v[3]);
  }
};

template <bool INV>
struct Dft<8, INV> {
  static TK_HD void run(cf* v) {
    const float h = 0.70710678118654752440f;
    cf e[4] = {v[0], v[2], v[4], v[6]};
    cf o[4] = {v[1], v[3], v[5], v[7]};
    Dft<4, INV>::run(e);
    Dft<4, INV>::run(o);
    // o[k] *= w8^k : w8 = (1 - i)/sqrt2 (forward)
    cf o1 = INV ? mk((o[1].x - o[1].y) * h, (o[1].x + o[1].y) * h)
                : mk((o[1].x + o[1].y) * h, (o[1].y - o[1].x) * h);
    cf o3 = INV ? mk((-o[3].x - o[3].y) * h, (o[3].x - o[3].y) * h)
                : mk((o[3].y - o[3].x) * h, (-o[3].x - o[3].y) * h);
    v[0] = e[0] + o[0];
    v[4] = e[0] - o[0];
    v[1] = e[1] + o1;
    v[5] = e[1] - o1;
    addsub_mi<INV>(e[2], o[2], v[2], v[6]);
    v[3] = e[3] + o3;
    v[7] = e[3] - o3;
  }
};

template <bool INV>
struct Dft<16, INV> {
  static TK_HD void run(cf* v) {
    // n = 4*n1 + n2, k = k1 + 4*k2
    const float c1 = 0.92387953251128675613f;  // cos(pi/8)
    const float s1 = 0.38268343236508977173f;  // sin(pi/8)
    const float h = 0.70710678118654752440f;
    cf a[4][4];  // a[n2][k1]
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
      cf t[4] = {v[n2], v[4 + n2], v[8 + n2], v[12 + n2]};
      Dft<4, INV>::run(t);
#pragma unroll
      for (int k1 = 0; k1 < 4; ++k1) a[n2][k1] = t[k1];
    }
    // twiddles w16^(n2*k1), forward values (cos, -sin)
    const cf w1 = mk(c1, -s1), w2 = mk(h, -h), w3 = mk(s1, -c1);
    const cf w6 = mk(-h, -h), w9 = mk(-c1, s1);
    a[1][1] = mul_tw<INV>(a[1][1], w1);
    a[1][2] = mul_tw<INV>(a[1][2], w2);
    a[1][3] = mul_tw<INV>(a[1][3], w3);
    a[2][1] = mul_tw<INV>(a[2][1], w2);
    // a[2][2] takes w16^4 = -i: folded into the k1 = 2 butterfly below
    a[2][3] = mul_tw<INV>(a[2][3], w6);
    a[3][1] = mul_tw<INV>(a[3][1], w3);
    a[3][2] = mul_tw<INV>(a[3][2], w6);
    a[3][3] = mul_tw<INV>(a[3][3], w9);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
      cf t[4] = {a[0][k1], a[1][k1], a[2][k1], a[3][k1]};
      if (k1 == 2) {
        // radix 4 with t[2] still to be multiplied by -+i
        cf t0, t1;
        addsub_mi<INV>(t[0], t[2], t0, t1);
        const cf t2 = t[1] + t[3], d = t[1] - t[3];
        t[0] = t0 + t2;
        t[2] = t0 - t2;
        addsub_mi<INV>(t1, d, t[1], t[3]);
      } else {
        Dft<4, INV>::run(t);
      }
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) v[k1 + 4 * k2] = t[k2];
    }
  }
};

template <bool INV>
struct Dft<32, INV> {
  static TK_HD void run(cf* v) {
    // X[k] = E[k] + w32^k O[k], X[k+16] = E[k] - w32^k O[k]
    const float c[16] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654757f, 0.55557023301960229f, 0.38268343236508984f, 0.19509032201612833f, 0.0f, -0.19509032201612819f, -0.38268343236508973f, -0.55557023301960196f, -0.70710678118654746f, -0.83146961230254535f, -0.92387953251128674f, -0.98078528040323043f};
    const float s[16] = {0.0f, 0.19509032201612825f, 0.38268343236508978f, 0.55557023301960218f, 0.70710678118654746f, 0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f, 1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254546f, 0.70710678118654757f, 0.55557023301960218f, 0.38268343236508989f, 0.19509032201612861f};
    cf e[16], o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      e[i] = v[2 * i];
      o[i] = v[2 * i + 1];
    }
    Dft<16, INV>::run(e);
    Dft<16, INV>::run(o);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const cf w = mk(c[k], -s[k]);  // forward twiddle exp(-2 pi i k / 32)
      const cf t = mul_tw<INV>(o[k], w);
      v[k] = e[k] + t;
      v[k + 16] = e[k] - t;
    }
  }
};

// ---- odd radices (round 6: detector sizes with factors 3, 5, 7, 11, 13 --
// 96, 192, 320, 384, 640, 768 ... -- through fft_mixed.h)
template <bool INV>
struct Dft<3, INV> {
  static TK_HD void run(cf* v) {
    const float s = 0.86602540378443864676f;  // sin(2 pi / 3)
    const cf t1 = v[1] + v[2];
    const cf t2 = mk(v[0].x - 0.5f * t1.x, v[0].y - 0.5f * t1.y);
    const cf t3 = (v[1] - v[2]) * s;
    v[0] = v[0] + t1;
    addsub_mi<INV>(t2, t3, v[1], v[2]);
  }
};

template <bool INV>
struct Dft<5, INV> {
  static TK_HD void run(cf* v) {
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    const cf a1 = v[1] + v[4], a2 = v[2] + v[3];
    const cf b1 = v[1] - v[4], b2 = v[2] - v[3];
    const cf r1 = mk(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
    const cf r2 = mk(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
    const cf i1 = mk(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y);
    const cf i2 = mk(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y);
    v[0] = v[0] + a1 + a2;
    addsub_mi<INV>(r1, i1, v[1], v[4]);
    addsub_mi<INV>(r2, i2, v[2], v[3]);
  }
};

// cos / sin of 2 pi k / P
template <int P>
struct PrimeTab;
template <>
struct PrimeTab<7> {
  static constexpr float c[7] = {1.f, 0.62348980185873359f, -0.22252093395631434f, -0.90096886790241903f, -0.90096886790241915f, -0.22252093395631459f, 0.62348980185873337f};
  static constexpr float s[7] = {0.f, 0.7818314824680298f, 0.97492791218182362f, 0.43388373911755823f, -0.43388373911755801f, -0.97492791218182362f, -0.78183148246802991f};
};
template <>
struct PrimeTab<11> {
  static constexpr float c[11] = {1.f, 0.84125353283118121f, 0.41541501300188644f, -0.142314838273285f, -0.65486073394528499f, -0.95949297361449737f, -0.95949297361449748f, -0.65486073394528521f, -0.14231483827328523f, 0.41541501300188605f, 0.84125353283118121f};
  static constexpr float s[11] = {0.f, 0.54064081745559756f, 0.90963199535451833f, 0.9898214418809328f, 0.75574957435425827f, 0.28173255684142967f, -0.28173255684142939f, -0.75574957435425816f, -0.98982144188093268f, -0.90963199535451855f, -0.54064081745559744f};
};
template <>
struct PrimeTab<13> {
  static constexpr float c[13] = {1.f, 0.88545602565320991f, 0.56806474673115592f, 0.12053668025532301f, -0.35460488704253545f, -0.74851074817110119f, -0.97094181742605201f, -0.97094181742605212f, -0.7485107481711013f, -0.3546048870425359f, 0.1205366802553232f, 0.56806474673115481f, 0.88545602565321002f};
  static constexpr float s[13] = {0.f, 0.46472317204376851f, 0.82298386589365635f, 0.99270887409805397f, 0.93501624268541483f, 0.66312265824079519f, 0.23931566428755768f, -0.23931566428755743f, -0.66312265824079497f, -0.93501624268541472f, -0.99270887409805397f, -0.82298386589365702f, -0.4647231720437684f};
};

// Odd prime P by the symmetric form: with a_r = v_r + v_{P-r}, b_r = v_r - v_{P-r},
// X_q = v_0 + sum_r cos(2 pi r q / P) a_r  -+ i sum_r sin(2 pi r q / P) b_r,
// X_{P-q} the same with the other sign: (P-1)^2 / 2 real multiplies per component.
template <int P, bool INV>
struct DftPrime {
  static TK_HD void run(cf* v) {
    constexpr int H = (P - 1) / 2;
    cf a[H], b[H];
#pragma unroll
    for (int r = 1; r <= H; ++r) {
      a[r - 1] = v[r] + v[P - r];
      b[r - 1] = v[r] - v[P - r];
    }
    cf x0 = v[0];
#pragma unroll
    for (int r = 0; r < H; ++r) x0 = x0 + a[r];
    cf out[P];
    out[0] = x0;
#pragma unroll
    for (int q = 1; q <= H; ++q) {
      cf re = v[0], im = mk(0.f, 0.f);
#pragma unroll
      for (int r = 1; r <= H; ++r) {
        const float c = PrimeTab<P>::c[(r * q) % P], s = PrimeTab<P>::s[(r * q) % P];
        re.x += c * a[r - 1].x;
        re.y += c * a[r - 1].y;
        im.x += s * b[r - 1].x;
        im.y += s * b[r - 1].y;
      }
      addsub_mi<INV>(re, im, out[q], out[P - q]);
    }
#pragma unroll
    for (int k = 0; k < P; ++k) v[k] = out[k];
  }
};
template <bool INV>
struct Dft<7, INV> : DftPrime<7, INV> {};
template <bool INV>
struct Dft<11, INV> : DftPrime<11, INV> {};
template <bool INV>
struct Dft<13, INV> : DftPrime<13, INV> {};

// ---- composite radices 6, 10, 12, 20, 24 (round 6): one Cooley-Tukey step in
// registers, R = R1 * R2 with input n = R2 n1 + n2 and output k = k1 + R1 k2:
// R2 DFTs of length R1, the twiddles w_R^(n2 k1) from a literal table, R1 DFTs
// of length R2.  A 384-point line is two LDS stages (24 x 16) instead of three
// (3 x 8 x 16): fewer barriers, and the index arithmetic of a butterfly is
// shared by 24 elements.
template <int R>
struct UnitTab;
template <>
struct UnitTab<6> {
  static constexpr float c[6] = {1.0f, 0.5f, -0.5f, -1.0f, -0.5f, 0.5f};
  static constexpr float s[6] = {0.0f, 0.8660254037844386f, 0.86602540378443871f, 0.0f, -0.86602540378443837f, -0.8660254037844386f};
};
template <>
struct UnitTab<10> {
  static constexpr float c[10] = {1.0f, 0.80901699437494745f, 0.30901699437494745f, -0.30901699437494734f, -0.80901699437494734f, -1.0f, -0.80901699437494756f, -0.30901699437494756f, 0.30901699437494723f, 0.80901699437494734f};
  static constexpr float s[10] = {0.0f, 0.58778525229247314f, 0.95105651629515353f, 0.95105651629515364f, 0.58778525229247325f, 0.0f, -0.58778525229247303f, -0.95105651629515353f, -0.95105651629515364f, -0.58778525229247336f};
};
template <>
struct UnitTab<12> {
  static constexpr float c[12] = {1.0f, 0.86602540378443871f, 0.5f, 0.0f, -0.5f, -0.86602540378443871f, -1.0f, -0.86602540378443882f, -0.5f, 0.0f, 0.5f, 0.86602540378443837f};
  static constexpr float s[12] = {0.0f, 0.5f, 0.8660254037844386f, 1.0f, 0.86602540378443871f, 0.5f, 0.0f, -0.5f, -0.86602540378443837f, -1.0f, -0.8660254037844386f, -0.5f};
};
template <>
struct UnitTab<20> {
  static constexpr float c[20] = {1.0f, 0.95105651629515353f, 0.80901699437494745f, 0.58778525229247314f, 0.30901699437494745f, 0.0f, -0.30901699437494734f, -0.58778525229247303f, -0.80901699437494734f, -0.95105651629515353f, -1.0f, -0.95105651629515375f, -0.80901699437494756f, -0.58778525229247325f, -0.30901699437494756f, 0.0f, 0.30901699437494723f, 0.58778525229247292f, 0.80901699437494734f, 0.95105651629515353f};
  static constexpr float s[20] = {0.0f, 0.3090169943749474f, 0.58778525229247314f, 0.80901699437494745f, 0.95105651629515353f, 1.0f, 0.95105651629515364f, 0.80901699437494745f, 0.58778525229247325f, 0.30901699437494751f, 0.0f, -0.3090169943749469f, -0.58778525229247303f, -0.80901699437494734f, -0.95105651629515353f, -1.0f, -0.95105651629515364f, -0.80901699437494756f, -0.58778525229247336f, -0.30901699437494762f};
};
template <>
struct UnitTab<24> {
  static constexpr float c[24] = {1.0f, 0.96592582628906831f, 0.86602540378443871f, 0.70710678118654757f, 0.5f, 0.25881904510252074f, 0.0f, -0.25881904510252063f, -0.5f, -0.70710678118654746f, -0.86602540378443871f, -0.9659258262890682f, -1.0f, -0.96592582628906831f, -0.86602540378443882f, -0.70710678118654791f, -0.5f, -0.25881904510252063f, 0.0f, 0.2588190451025203f, 0.5f, 0.70710678118654735f, 0.86602540378443837f, 0.96592582628906809f};
  static constexpr float s[24] = {0.0f, 0.25881904510252074f, 0.5f, 0.70710678118654746f, 0.8660254037844386f, 0.96592582628906831f, 1.0f, 0.96592582628906831f, 0.86602540378443871f, 0.70710678118654757f, 0.5f, 0.25881904510252102f, 0.0f, -0.25881904510252079f, -0.5f, -0.70710678118654713f, -0.86602540378443837f, -0.96592582628906831f, -1.0f, -0.96592582628906842f, -0.8660254037844386f, -0.70710678118654768f, -0.5f, -0.25881904510252157f};
};

template <int R1, int R2, bool INV>
struct DftCT {
  static TK_HD void run(cf* v) {
    constexpr int R = R1 * R2;
    cf a[R2][R1];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) {
      cf t[R1];
#pragma unroll
      for (int n1 = 0; n1 < R1; ++n1) t[n1] = v[R2 * n1 + n2];
      Dft<R1, INV>::run(t);
#pragma unroll
      for (int k1 = 0; k1 < R1; ++k1)
        a[n2][k1] = (n2 * k1) % R == 0
                        ? t[k1]
                        : mul_tw<INV>(t[k1], mk(UnitTab<R>::c[(n2 * k1) % R],
                                                -UnitTab<R>::s[(n2 * k1) % R]));
    }
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
      cf t[R2];
#pragma unroll
      for (int n2 = 0; n2 < R2; ++n2) t[n2] = a[n2][k1];
      Dft<R2, INV>::run(t);
#pragma unroll
      for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = t[k2];
    }
  }
};
template <bool INV>
struct Dft<6, INV> : DftCT<2, 3, INV> {};
template <bool INV>
struct Dft<10, INV> : DftCT<2, 5, INV> {};
template <bool INV>
struct Dft<12, INV> : DftCT<4, 3, INV> {};
template <bool INV>
struct Dft<20, INV> : DftCT<4, 5, INV> {};
template <bool INV>
struct Dft<24, INV> : DftCT<8, 3, INV> {};

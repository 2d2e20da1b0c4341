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

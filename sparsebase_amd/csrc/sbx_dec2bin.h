// sbx_dec2bin.h — exact decimal -> binary floating point for the Matrix Market parser.
//
// value = w * 10^q10 with w < 2^128 (up to 38 significant decimal digits) is converted to
// the nearest double / float, ties to even — the result strtod / strtof (and therefore
// `istream >> double/float`, which the reference reader uses, io/mtx_reader.cc:331) produce.
// No floating-point arithmetic is involved: the value is formed exactly in multi-limb
// integers (w * 5^q for q >= 0; a 64-bit restoring division w * 2^s / 5^-q with the
// remainder as sticky bit for q < 0) and rounded once, so there are no double-rounding or
// "hard case" fallbacks.  Compiles for the device (HIP) and for the host (g++: the CPU unit
// test tests/test_dec2bin.py checks it against strtod/strtof on millions of random inputs).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SBX_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define SBX_HD inline
#endif

#define SBX_POW5_MAX 400   /* |q10| up to here is converted exactly; beyond: 0 / inf by magnitude */
#define SBX_POW5_LIMBS 16  /* 5^400 has 929 bits */
#define SBX_DECIMAL_MAX_DIGITS 38

// Behind the 5^k limbs the table holds, for q = SBX_EL_QMIN .. SBX_EL_QMAX, the 128-bit truncated significand of 5^q
// (high word, low word) that the Eisel-Lemire fast path multiplies by (Lemire, "Number parsing at a gigabyte per
// second", 2021; the construction is the one of its reference implementation): for q >= 0 the top 128 bits of 5^q,
// for q < 0 those of 2^b / 5^-q + 1.
#define SBX_EL_QMIN (-342)
#define SBX_EL_QMAX 308
#define SBX_EL_ENTRIES (SBX_EL_QMAX - SBX_EL_QMIN + 1)
#define SBX_POW5_WORDS ((SBX_POW5_MAX + 1) * SBX_POW5_LIMBS)
#define SBX_TABLE_WORDS (SBX_POW5_WORDS + 2 * SBX_EL_ENTRIES)

static inline void sbx_el_table_fill(const uint64_t *pow5, uint64_t *el);

// table[k * SBX_POW5_LIMBS + i] = limb i (little endian) of 5^k, k = 0..SBX_POW5_MAX; SBX_TABLE_WORDS words in all
static inline void sbx_pow5_table_fill(uint64_t *table) {
  for (int i = 0; i < SBX_POW5_LIMBS; i++) table[i] = 0;
  table[0] = 1;
  for (int k = 1; k <= SBX_POW5_MAX; k++) {
    const uint64_t *prev = table + (size_t)(k - 1) * SBX_POW5_LIMBS;
    uint64_t *cur = table + (size_t)k * SBX_POW5_LIMBS;
    unsigned __int128 carry = 0;
    for (int i = 0; i < SBX_POW5_LIMBS; i++) {
      const unsigned __int128 t = (unsigned __int128)prev[i] * 5u + carry;
      cur[i] = (uint64_t)t;
      carry = t >> 64;
    }
  }
  sbx_el_table_fill(table, table + SBX_POW5_WORDS);
}

// (host only, once per handle: plain multi-limb shifts and a restoring division)
static inline void sbx_el_table_fill(const uint64_t *pow5, uint64_t *el) {
  enum { NL = 30 };  // 2^b has up to 2 * 795 + 128 + 1 bits
  auto bitlen = [](const uint64_t *a, int n) {
    while (n > 0 && a[n - 1] == 0) n--;
    return n == 0 ? 0 : 64 * n - __builtin_clzll(a[n - 1]);
  };
  auto top128 = [&](const uint64_t *a, int bits, uint64_t *hi, uint64_t *lo) {  // bits [bits-128, bits) of a, zero-filled below 0
    auto bit = [&](int i) -> uint64_t { return i < 0 ? 0 : (a[i >> 6] >> (i & 63)) & 1u; };
    uint64_t h = 0, l = 0;
    for (int i = 0; i < 64; i++) h = (h << 1) | bit(bits - 1 - i);
    for (int i = 0; i < 64; i++) l = (l << 1) | bit(bits - 65 - i);
    *hi = h;
    *lo = l;
  };
  for (int q = SBX_EL_QMIN; q <= SBX_EL_QMAX; q++) {
    uint64_t *out = el + 2 * (size_t)(q - SBX_EL_QMIN);
    if (q >= 0) {
      uint64_t a[NL] = {0};
      for (int i = 0; i < SBX_POW5_LIMBS; i++) a[i] = pow5[(size_t)q * SBX_POW5_LIMBS + i];
      top128(a, bitlen(a, NL), &out[0], &out[1]);  // normalised so that bit 127 is set, truncated
      continue;
    }
    const uint64_t *p = pow5 + (size_t)(-q) * SBX_POW5_LIMBS;
    const int z = bitlen(p, SBX_POW5_LIMBS);  // smallest z with 2^z >= 5^-q (5^k is no power of two)
    const int b = q >= -27 ? z + 127 : 2 * z + 128;
    // quo = floor(2^b / p) by restoring division, then + 1
    uint64_t quo[NL] = {0}, rem[SBX_POW5_LIMBS + 1] = {0};
    for (int i = b; i >= 0; i--) {
      uint64_t carry = i == b ? 1u : 0u;  // rem = rem * 2 + bit i of 2^b
      for (int j = 0; j <= SBX_POW5_LIMBS; j++) {
        const uint64_t nc = rem[j] >> 63;
        rem[j] = (rem[j] << 1) | carry;
        carry = nc;
      }
      bool ge = rem[SBX_POW5_LIMBS] != 0;
      if (!ge) {
        ge = true;
        for (int j = SBX_POW5_LIMBS - 1; j >= 0; j--)
          if (rem[j] != p[j]) { ge = rem[j] > p[j]; break; }
      }
      if (ge) {
        uint64_t borrow = 0;
        for (int j = 0; j <= SBX_POW5_LIMBS; j++) {
          const uint64_t pj = j < SBX_POW5_LIMBS ? p[j] : 0;
          const uint64_t d = rem[j] - pj - borrow;
          borrow = (rem[j] < pj || (rem[j] == pj && borrow)) ? 1u : 0u;
          rem[j] = d;
        }
        quo[i >> 6] |= (uint64_t)1 << (i & 63);
      }
    }
    for (int j = 0; j < NL; j++)
      if (++quo[j] != 0) break;  // + 1
    top128(quo, bitlen(quo, NL), &out[0], &out[1]);  // (q >= -27: exactly 128 bits; below: halved until it fits)
  }
}

namespace sbx_d2b {

SBX_HD int clz64(uint64_t x) { return __builtin_clzll(x); }

// number of significant limbs / bits of a little-endian multi-limb integer
SBX_HD int limbs_of(const uint64_t *a, int n) {
  while (n > 0 && a[n - 1] == 0) n--;
  return n;
}
SBX_HD int bits_of(const uint64_t *a, int n) {
  n = limbs_of(a, n);
  return n == 0 ? 0 : 64 * n - clz64(a[n - 1]);
}

// Rounds the exact value  mant * 2^e2  (+ something in (0, 2^e2) if sticky)  with mant in [2^63, 2^64)
// to a binary format with P significand bits (incl. the hidden one), minimum normal exponent EMIN and
// maximum EMAX (IEEE: double 53/-1022/1023, float 24/-126/127).  Returns the IEEE bit pattern without sign.
template <int P, int EMIN, int EMAX, int EXP_BITS>
SBX_HD uint64_t round_pack(uint64_t mant, int e2, bool sticky) {
  // value = mant * 2^e2 = 1.xxx * 2^(e2 + 63)
  int exp = e2 + 63;                 // unbiased exponent of the leading bit
  int drop = 64 - P;                 // bits to drop for a normal number
  if (exp < EMIN) drop += EMIN - exp;  // subnormal: fewer significand bits
  uint64_t kept;
  bool round_bit, rest;
  if (drop >= 65) {                  // far below the smallest subnormal
    kept = 0;
    round_bit = false;
    rest = true;
  } else if (drop == 64) {
    kept = 0;
    round_bit = (mant >> 63) & 1u;
    rest = sticky || (mant & 0x7FFFFFFFFFFFFFFFull) != 0;
  } else {
    kept = mant >> drop;
    round_bit = (mant >> (drop - 1)) & 1u;
    rest = sticky || (mant & ((1ull << (drop - 1)) - 1ull)) != 0;
  }
  if (round_bit && (rest || (kept & 1u))) kept++;  // nearest, ties to even
  if (exp < EMIN) {
    // subnormal (or rounded up to the smallest normal: the pattern below is then exactly that)
    return kept;  // biased exponent field 0 (or 1 if kept reached 2^(P-1))
  }
  if (kept >> P) {  // significand overflowed to P+1 bits
    kept >>= 1;
    exp++;
  }
  if (exp > EMAX) return (uint64_t)((1u << EXP_BITS) - 1u) << (P - 1);  // infinity
  return ((uint64_t)(exp - EMIN + 1) << (P - 1)) | (kept & ((1ull << (P - 1)) - 1ull));
}

// dst (nd limbs, zeroed here) = src (ns limbs) << shift
SBX_HD void shl_multi(uint64_t *dst, int nd, const uint64_t *src, int ns, int shift) {
  for (int i = 0; i < nd; i++) dst[i] = 0;
  const int limb = shift >> 6, bit = shift & 63;
  for (int i = 0; i < ns; i++) {
    if (i + limb < nd) dst[i + limb] |= src[i] << bit;
    if (bit && i + limb + 1 < nd) dst[i + limb + 1] |= src[i] >> (64 - bit);
  }
}

// exact (mant, e2, sticky) with mant in [2^63, 2^64) for w * 10^q10, w = w_hi * 2^64 + w_lo != 0,
// |q10| <= SBX_POW5_MAX
SBX_HD void exact_scale(uint64_t w_hi, uint64_t w_lo, int q10, const uint64_t *pow5, uint64_t *mant, int *e2,
                        bool *sticky) {
  constexpr int NL = SBX_POW5_LIMBS + 2;
  const uint64_t w[2] = {w_lo, w_hi};
  const int wl = w_hi ? 2 : 1;
  if (q10 >= 0) {
    // N = w * 5^q, value = N * 2^q
    const uint64_t *p5 = pow5 + (size_t)q10 * SBX_POW5_LIMBS;
    uint64_t n[NL];
    for (int i = 0; i < NL; i++) n[i] = 0;
    for (int j = 0; j < wl; j++) {
      unsigned __int128 carry = 0;
      for (int i = 0; i < SBX_POW5_LIMBS; i++) {
        const unsigned __int128 t = (unsigned __int128)p5[i] * w[j] + n[i + j] + carry;
        n[i + j] = (uint64_t)t;
        carry = t >> 64;
      }
      n[SBX_POW5_LIMBS + j] += (uint64_t)carry;
    }
    const int nl = limbs_of(n, NL);
    const int lz = clz64(n[nl - 1]);
    uint64_t top = n[nl - 1] << lz;  // top 64 bits of N
    bool st = false;
    if (nl >= 2) {
      if (lz) top |= n[nl - 2] >> (64 - lz);
      st = lz ? (n[nl - 2] << lz) != 0 : n[nl - 2] != 0;
      for (int i = 0; i + 2 < nl; i++) st |= n[i] != 0;
    }
    *mant = top;
    *sticky = st;
    *e2 = q10 + (64 * nl - lz) - 64;  // N = top * 2^(bits - 64)
    return;
  }
  // q < 0: value = w / (5^k * 2^k).  Restoring division with 64 quotient bits.
  const int k = -q10;
  const uint64_t *d = pow5 + (size_t)k * SBX_POW5_LIMBS;
  const int dl = limbs_of(d, SBX_POW5_LIMBS);
  const int dbits = 64 * dl - clz64(d[dl - 1]);
  const int wbits = 64 * wl - clz64(w[wl - 1]);
  // align the two operands to the same bit length, so that rem / dd is in (1/2, 2):
  // sh >= 0: rem = w << sh, dd = d;  sh < 0 (5^k shorter than w): rem = w, dd = d << -sh
  const int sh = dbits - wbits;
  uint64_t rem[NL], dd[NL];
  if (sh >= 0) {
    shl_multi(rem, NL, w, wl, sh);
    shl_multi(dd, NL, d, dl, 0);
  } else {
    shl_multi(rem, NL, w, wl, 0);
    shl_multi(dd, NL, d, dl, -sh);
  }
  const int ml = dl > wl ? dl : wl;
  const int rl = ml + 1;  // limbs that can be non-zero (rem < 2 * dd at every step)
  uint64_t q = 0;
  int produced = 0;  // quotient bits from the leading one on
  int steps = 0;
  while (produced < 64) {
    bool ge = true;
    for (int i = rl - 1; i >= 0; i--) {
      if (rem[i] != dd[i]) {
        ge = rem[i] > dd[i];
        break;
      }
    }
    if (ge) {
      unsigned __int128 borrow = 0;
      for (int i = 0; i < rl; i++) {
        const unsigned __int128 t = (unsigned __int128)rem[i] - dd[i] - borrow;
        rem[i] = (uint64_t)t;
        borrow = (t >> 64) & 1u;
      }
    }
    if (produced > 0 || ge) {
      q = (q << 1) | (ge ? 1u : 0u);
      produced++;
    }
    steps++;
    if (produced < 64) {  // rem <<= 1
      for (int i = rl - 1; i > 0; i--) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 63);
      rem[0] <<= 1;
    }
  }
  bool st = false;
  for (int i = 0; i < rl; i++) st |= rem[i] != 0;
  *mant = q;
  *sticky = st;
  // quotient bit j (j = 0 first step) has weight 2^-j relative to rem0 / dd; q collects `steps` bits:
  // rem0 / dd = q * 2^-(steps - 1) (+ remainder), and w / d = (rem0 / dd) * 2^-sh
  *e2 = -(steps - 1) - sh - k;
}

template <int P, int EMIN, int EMAX, int EXP_BITS>
SBX_HD uint64_t convert(uint64_t w_hi, uint64_t w_lo, int q10, const uint64_t *pow5) {
  if ((w_hi | w_lo) == 0) return 0;
  if (q10 > SBX_POW5_MAX) return (uint64_t)((1u << EXP_BITS) - 1u) << (P - 1);  // >= 10^401: infinity in both formats
  if (q10 < -SBX_POW5_MAX) return 0;  // < 10^38 * 10^-401: below half the smallest subnormal
  uint64_t mant;
  int e2;
  bool sticky;
  exact_scale(w_hi, w_lo, q10, pow5, &mant, &e2, &sticky);
  return round_pack<P, EMIN, EMAX, EXP_BITS>(mant, e2, sticky);
}

}  // namespace sbx_d2b

// ---- token -> (sign, w, q10): the characters `istream >> double` consumes (libstdc++ num_get: digits, one
// '.', an exponent) — no hex floats, no inf/nan.  status: 0 ok, 1 malformed, 2 more than 38 significant
// digits with a non-zero tail (not representable in the 128-bit w; refused rather than rounded twice).
struct sbx_decimal {
  uint64_t w_lo, w_hi;  // the first (up to) 38 significant digits as an integer
  int q10;
  int neg;
  int status;
};

SBX_HD sbx_decimal sbx_parse_decimal(const char *s, int64_t len) {
  sbx_decimal r;
  r.w_lo = 0; r.w_hi = 0; r.q10 = 0; r.neg = 0; r.status = 1;
  unsigned __int128 w = 0;
  int64_t i = 0;
  if (i < len && (s[i] == '+' || s[i] == '-')) { r.neg = s[i] == '-'; i++; }
  int digits = 0;       // significant digits taken into w
  int any = 0;          // any digit seen at all
  int dropped = 0;      // integer-part digits that did not fit (all zero, or status 2)
  int frac_taken = 0;   // fractional digits taken into w (or skipped leading zeros of 0.000x)
  bool tail_nonzero = false;
  for (; i < len && s[i] >= '0' && s[i] <= '9'; i++) {
    any = 1;
    const int dgt = s[i] - '0';
    if (digits == 0 && dgt == 0) continue;  // leading zero
    if (digits < 38) { w = w * 10u + (unsigned)dgt; digits++; }
    else { dropped++; tail_nonzero |= dgt != 0; }
  }
  if (i < len && s[i] == '.') {
    i++;
    for (; i < len && s[i] >= '0' && s[i] <= '9'; i++) {
      any = 1;
      const int dgt = s[i] - '0';
      if (digits == 0 && dgt == 0) { frac_taken++; continue; }  // 0.000x: scale only
      if (digits < 38) { w = w * 10u + (unsigned)dgt; digits++; frac_taken++; }
      else tail_nonzero |= dgt != 0;
    }
  }
  if (!any) return r;
  long e = 0;
  if (i < len && (s[i] == 'e' || s[i] == 'E')) {
    i++;
    int eneg = 0;
    if (i < len && (s[i] == '+' || s[i] == '-')) { eneg = s[i] == '-'; i++; }
    if (!(i < len && s[i] >= '0' && s[i] <= '9')) return r;  // "1e" / "1e+": malformed
    for (; i < len && s[i] >= '0' && s[i] <= '9'; i++)
      if (e < 100000) e = e * 10 + (s[i] - '0');
    if (eneg) e = -e;
  }
  if (i != len) return r;  // trailing garbage inside the token
  long q = e + dropped - frac_taken;
  if (q > 200000) q = 200000;
  if (q < -200000) q = -200000;
  r.q10 = (int)q;
  r.w_lo = (uint64_t)w;
  r.w_hi = (uint64_t)(w >> 64);
  r.status = tail_nonzero ? 2 : 0;
  return r;
}

// token -> signed 64-bit integer (what `istream >> IDType` consumes): status 0 ok, 1 malformed / overflow
SBX_HD int sbx_parse_integer(const char *s, int64_t len, long long *out) {
  int64_t i = 0;
  int neg = 0;
  if (i < len && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; i++; }
  if (i >= len) return 1;
  unsigned long long v = 0;
  for (; i < len; i++) {
    if (s[i] < '0' || s[i] > '9') return 1;
    if (v > 922337203685477580ull) return 1;
    v = v * 10u + (unsigned)(s[i] - '0');
    if (v > 9223372036854775807ull) return 1;
  }
  *out = neg ? -(long long)v : (long long)v;
  return 0;
}

namespace sbx_d2b {
SBX_HD void mul64(uint64_t a, uint64_t b, uint64_t *hi, uint64_t *lo) {
#if defined(__HIP_DEVICE_COMPILE__)
  *lo = a * b;
  *hi = __umul64hi(a, b);
#else
  const unsigned __int128 t = (unsigned __int128)a * b;
  *lo = (uint64_t)t;
  *hi = (uint64_t)(t >> 64);
#endif
}
// Eisel-Lemire: w * 10^q (w != 0, at most 64 bits) to the nearest double / float with one or two 64 x 64 multiplications
// by the tabulated significand of 5^q; returns false in the rare cases it cannot decide (the exact path then does).
// MB explicit significand bits, BIAS the exponent bias + 1 - 1 (1023 / 127), INFP the all-ones exponent, [RLO, RHI] the
// powers of ten where w * 10^q can fall exactly half way between two values.
template <int MB, int BIAS, int INFP, int RLO, int RHI, int QLO, int QHI>
SBX_HD bool eisel_lemire(uint64_t w, int q, const uint64_t *el, uint64_t *bits) {
  if (q < QLO || q > QHI) return false;
  const int lz = clz64(w);
  w <<= lz;
  const uint64_t *t = el + 2 * (size_t)(q - SBX_EL_QMIN);
  const uint64_t pmask = 0xFFFFFFFFFFFFFFFFull >> (MB + 3);
  uint64_t hi, lo;
  mul64(w, t[0], &hi, &lo);
  if ((hi & pmask) == pmask) {  // the MB + 3 bits wanted are not settled by the high word of the table entry alone
    uint64_t hi2, lo2;
    mul64(w, t[1], &hi2, &lo2);
    lo += hi2;
    if (hi2 > lo) hi++;
  }
  if (lo == 0xFFFFFFFFFFFFFFFFull && !(q >= -27 && q <= 55)) return false;
  const int upperbit = (int)(hi >> 63);
  const int sh = upperbit + 64 - MB - 3;
  uint64_t mant = hi >> sh;
  int power2 = (int)((((int64_t)(152170 + 65536) * q) >> 16) + 63) + upperbit - lz + BIAS;
  if (power2 <= 0) {  // subnormal
    if (-power2 + 1 >= 64) { *bits = 0; return true; }
    mant >>= -power2 + 1;
    mant += mant & 1u;
    mant >>= 1;
    power2 = mant < (1ull << MB) ? 0 : 1;
    *bits = ((uint64_t)power2 << MB) | (mant & ~(1ull << MB));
    return true;
  }
  if (lo <= 1 && q >= RLO && q <= RHI && (mant & 3u) == 1u) {  // exactly half way: to even
    if ((mant << sh) == hi) mant &= ~1ull;
  }
  mant += mant & 1u;
  mant >>= 1;
  if (mant >= (2ull << MB)) {
    mant = 1ull << MB;
    power2++;
  }
  mant &= ~(1ull << MB);
  if (power2 >= INFP) { *bits = (uint64_t)INFP << MB; return true; }
  *bits = ((uint64_t)power2 << MB) | mant;
  return true;
}
}  // namespace sbx_d2b

// IEEE bit patterns (sign applied by the caller).
// Fast path (Clinger 1990): when the digits w and the power of ten are both exactly representable in the target
// format, ONE correctly rounded IEEE multiplication or division gives the correctly rounded value of w * 10^q — the
// same bits as the exact multi-limb path below it, at a few instructions instead of a few thousand (most values of
// real files have fewer than 16 significant digits).
SBX_HD uint64_t sbx_decimal_to_double_bits(const sbx_decimal &d, const uint64_t *pow5) {
  if (d.w_hi == 0 && d.w_lo < (1ull << 53) && d.q10 >= -22 && d.q10 <= 22) {
    const double p10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                            1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const double w = (double)d.w_lo;
    const double r = d.q10 < 0 ? w / p10[-d.q10] : w * p10[d.q10];
    uint64_t bits;
    memcpy(&bits, &r, sizeof bits);
    return bits;
  }
  if (d.w_hi == 0 && d.w_lo != 0) {  // up to 19 digits: Eisel-Lemire, exact path only where it cannot decide
    uint64_t bits;
    if (sbx_d2b::eisel_lemire<52, 1023, 0x7FF, -4, 23, SBX_EL_QMIN, SBX_EL_QMAX>(d.w_lo, d.q10, pow5 + SBX_POW5_WORDS, &bits))
      return bits;
  }
  return sbx_d2b::convert<53, -1022, 1023, 11>(d.w_hi, d.w_lo, d.q10, pow5);
}
SBX_HD uint32_t sbx_decimal_to_float_bits(const sbx_decimal &d, const uint64_t *pow5) {
  if (d.w_hi == 0 && d.w_lo < (1ull << 24) && d.q10 >= -10 && d.q10 <= 10) {
    const float p10[11] = {1e0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f};
    const float w = (float)d.w_lo;
    const float r = d.q10 < 0 ? w / p10[-d.q10] : w * p10[d.q10];
    uint32_t bits;
    memcpy(&bits, &r, sizeof bits);
    return bits;
  }
  if (d.w_hi == 0 && d.w_lo != 0) {
    uint64_t bits;
    if (sbx_d2b::eisel_lemire<23, 127, 0xFF, -17, 10, -64, 38>(d.w_lo, d.q10, pow5 + SBX_POW5_WORDS, &bits))
      return (uint32_t)bits;
  }
  return (uint32_t)sbx_d2b::convert<24, -126, 127, 8>(d.w_hi, d.w_lo, d.q10, pow5);
}

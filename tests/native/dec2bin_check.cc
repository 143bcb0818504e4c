// CPU unit test of sparsebase_amd/csrc/sbx_dec2bin.h against strtod / strtof (test infrastructure).
// usage: dec2bin_check <count> <seed>    prints "ok <count>" or the first mismatches
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "sbx_dec2bin.h"

static std::vector<uint64_t> table((size_t)SBX_TABLE_WORDS);

static int check(const std::string &tok, int *bad) {
  const sbx_decimal d = sbx_parse_decimal(tok.data(), (int64_t)tok.size());
  char *end = nullptr;
  const double want_d = strtod(tok.c_str(), &end);
  const bool full = end == tok.c_str() + tok.size();
  if (d.status == 1) {
    if (full && tok.find_first_of("xXpPnNiI") == std::string::npos) { std::printf("parser rejects %s\n", tok.c_str()); (*bad)++; }
    return 0;
  }
  if (d.status == 2) return 0;  // > 38 significant digits with a non-zero tail: refused by design
  const float want_f = strtof(tok.c_str(), nullptr);
  uint64_t got_d = sbx_decimal_to_double_bits(d, table.data()) | ((uint64_t)d.neg << 63);
  uint32_t got_f = sbx_decimal_to_float_bits(d, table.data()) | ((uint32_t)d.neg << 31);
  uint64_t wd; uint32_t wf;
  memcpy(&wd, &want_d, 8); memcpy(&wf, &want_f, 4);
  if (got_d != wd || got_f != wf) {
    if (*bad < 10) std::printf("MISMATCH %s: double %016" PRIx64 " want %016" PRIx64 "  float %08x want %08x (w_hi=%" PRIu64 " w_lo=%" PRIu64 " q=%d)\n",
                               tok.c_str(), got_d, wd, got_f, wf, d.w_hi, d.w_lo, d.q10);
    (*bad)++;
  }
  return 1;
}

int main(int argc, char **argv) {
  const long count = argc > 1 ? atol(argv[1]) : 100000;
  std::mt19937_64 g(argc > 2 ? atoll(argv[2]) : 1);
  sbx_pow5_table_fill(table.data());
  int bad = 0;
  long checked = 0;
  const char *fixed[] = {"0", "-0", "0.0", "1", "-1", "1e0", "1.", ".5", "5e-324", "4.9406564584124654e-324", "2.4703282292062327e-324",
                         "2.4703282292062328e-324", "2.2250738585072014e-308", "2.2250738585072011e-308", "1.7976931348623157e308",
                         "1.7976931348623159e308", "1e309", "1e-400", "1e400", "123456789012345678", "9007199254740993", "9007199254740992",
                         "9007199254740991", "0.1", "0.3", "1e23", "8.5e22", "1.4e-45", "7e-46", "3.4028235e38", "3.4028236e38", "3.5e38",
                         "1.17549435e-38", "16777217", "16777216", "16777215", "0.000001", "100000000000000000000", "1e", "1e+", "abc", "1.5x",
                         "+.e1", "12345678901234567890", "1234567890123456789", "00001.2500e+02", "-3.14159", "6.02214076e23", "1.0000000000000002",
                         "1.00000000000000011102230246251565404236316680908203125", "4.35", "0.5e-323", "1e-323", "9.5e-324"};
  for (const char *f : fixed) checked += check(f, &bad);
  for (long i = 0; i < count; i++) {
    std::string tok;
    const int kind = (int)(g() % 9);
    if (kind == 0) {  // random double printed with 17 significant digits
      uint64_t bits = g();
      double x; memcpy(&x, &bits, 8);
      if (!std::isfinite(x)) continue;
      char buf[64]; snprintf(buf, sizeof buf, "%.17g", x); tok = buf;
    } else if (kind == 1) {  // random float printed with 9 digits
      uint32_t bits = (uint32_t)g();
      float x; memcpy(&x, &bits, 4);
      if (!std::isfinite(x)) continue;
      char buf[64]; snprintf(buf, sizeof buf, "%.9g", (double)x); tok = buf;
    } else if (kind == 2) {  // "moderate" values like real .mtx files
      char buf[64]; snprintf(buf, sizeof buf, "%.*e", (int)(g() % 17), (double)(int64_t)(g() % 2000000 - 1000000) / (double)(g() % 100000 + 1));
      tok = buf;
    } else if (kind == 3) {  // mantissa digits + exponent over the whole range, incl. subnormal / overflow
      const int nd = 1 + (int)(g() % 19);
      for (int k = 0; k < nd; k++) tok += (char)('0' + (k == 0 ? 1 + g() % 9 : g() % 10));
      tok.insert(1 + g() % tok.size(), ".");
      char buf[16]; snprintf(buf, sizeof buf, "e%d", (int)(g() % 700) - 350); tok += buf;
      if (g() & 1) tok = "-" + tok;
    } else if (kind == 4) {  // halfway cases between adjacent doubles / floats (17+ digits exact decimal expansions are refused; use short ones)
      const uint64_t m = (g() % (1ull << 53)) | (1ull << 52);
      const int e = (int)(g() % 40) - 20;
      char buf[64]; snprintf(buf, sizeof buf, "%.19g", std::ldexp((double)m + 0.5 * (g() & 1), e)); tok = buf;
    } else if (kind == 6) {  // 20..38 significant digits: exact decimal expansions of halfway points and their neighbours
      const uint64_t m = (g() % (1ull << 53)) | (1ull << 52);
      const int e = (int)(g() % 60) - 30;
      char buf[128]; snprintf(buf, sizeof buf, "%.*e", 19 + (int)(g() % 19), std::ldexp((double)m + 0.5, e) * (1.0 + (double)(int)(g() % 3 - 1) * 1e-17));
      // printing a double cannot show the halfway point itself; build it from the integer: (2m + 1) * 2^(e-1)
      if (e >= 1 && e <= 10) { unsigned __int128 v = ((unsigned __int128)(2 * m + 1)) << (e - 1); std::string t; while (v) { t.insert(t.begin(), (char)('0' + (int)(v % 10))); v /= 10; } tok = t; }
      else tok = buf;
    } else if (kind == 7) {  // long random digit strings
      const int nd = 20 + (int)(g() % 30);
      for (int k = 0; k < nd; k++) tok += (char)('0' + (k == 0 ? 1 + g() % 9 : g() % 10));
      tok.insert(1 + g() % tok.size(), ".");
      char buf[16]; snprintf(buf, sizeof buf, "e%d", (int)(g() % 640) - 320); tok += buf;
    } else if (kind == 8) {  // around the limits of the one-operation fast path: digits below / at / above 2^53 and 2^24,
                             // powers of ten up to and beyond 10^22 / 10^10
      const int which = (int)(g() % 4);
      uint64_t w = which == 0 ? g() % (1ull << 53) : which == 1 ? (1ull << 53) - 2 + g() % 5 : which == 2 ? g() % (1ull << 24)
                                                                                          : (1ull << 24) - 2 + g() % 5;
      char buf[64]; snprintf(buf, sizeof buf, "%llue%d", (unsigned long long)w, (int)(g() % 53) - 26); tok = buf;
      if (g() & 1) tok = "-" + tok;
    } else {  // plain integers and fixed-point
      char buf[64]; snprintf(buf, sizeof buf, "%lld.%03d", (long long)(g() % 2000000000000ll) - 1000000000000ll, (int)(g() % 1000)); tok = buf;
    }
    checked += check(tok, &bad);
  }
  if (bad) { std::printf("FAILED %d of %ld\n", bad, checked); return 1; }
  std::printf("ok %ld\n", checked);
  return 0;
}

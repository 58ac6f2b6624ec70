// clap.h — clean-aperture arithmetic of the reference, restated: its `Fraction` class (box.cc:51-152, values are
// halved until numerator / denominator fit MAX_FRACTION_VALUE = 0x10000, heif_limits.h:47) and the rounding rules
// of Box_clap (box.cc:3771-3814).  Host only.
#ifndef HM_CLAP_H
#define HM_CLAP_H

#include <stdint.h>

#include <limits>

namespace hm {

struct Fraction {
  int32_t n = 0, d = 1;
  Fraction() = default;
  Fraction(int32_t num, int32_t den) : n(num), d(den) // box.cc:51-69
  {
    constexpr int32_t MAXV = 0x10000;
    while (d > MAXV || d < -MAXV) { n /= 2; d /= 2; }
    while (d > 1 && (n > MAXV || n < -MAXV)) { n /= 2; d /= 2; }
  }
  static Fraction wide(int64_t num, int64_t den) // box.cc:79-89
  {
    while (num < std::numeric_limits<int32_t>::min() || num > std::numeric_limits<int32_t>::max() ||
           den < std::numeric_limits<int32_t>::min() || den > std::numeric_limits<int32_t>::max()) {
      num = (num + (num >= 0 ? 1 : -1)) / 2;
      den = (den + (den >= 0 ? 1 : -1)) / 2;
    }
    Fraction f;
    f.n = (int32_t)num; f.d = (int32_t)den;
    return f;
  }
  Fraction operator+(const Fraction& b) const
  {
    if (d == b.d) return wide((int64_t)n + b.n, d);
    return wide((int64_t)n * b.d + (int64_t)b.n * d, (int64_t)d * b.d);
  }
  Fraction operator-(const Fraction& b) const
  {
    if (d == b.d) return wide((int64_t)n - b.n, d);
    return wide((int64_t)n * b.d - (int64_t)b.n * d, (int64_t)d * b.d);
  }
  Fraction operator+(int v) const { return wide(n + v * (int64_t)d, d); }
  Fraction operator-(int v) const { return wide(n - v * (int64_t)d, d); }
  Fraction operator/(int v) const { return wide(n, (int64_t)d * v); }
  int32_t round_down() const { return n / d; }
  int32_t round() const { return (int32_t)((n + (int64_t)d / 2) / d); }
  bool valid() const { return d != 0; }
};

struct Clap {
  Fraction width, height, hoff, voff;
  int left_rounded(int image_width) const // box.cc:3771-3782
  {
    const Fraction pcX = hoff + Fraction(image_width - 1, 2);
    const Fraction left = pcX - (width - 1) / 2;
    return left.round_down();
  }
  int right_rounded(int image_width) const { return (width - 1 + left_rounded(image_width)).round(); } // box.cc:3784-3789
  int top_rounded(int image_height) const // box.cc:3791-3797
  {
    const Fraction pcY = voff + Fraction(image_height - 1, 2);
    const Fraction top = pcY - (height - 1) / 2;
    return top.round();
  }
  int bottom_rounded(int image_height) const { return (height - 1 + top_rounded(image_height)).round(); } // box.cc:3799-3804
  int width_rounded() const { return width.round(); }
  int height_rounded() const { return height.round(); }
};

} // namespace hm
#endif

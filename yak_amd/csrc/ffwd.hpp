// ffwd.hpp -- n consecutive fp32 additions x = x + s (round to nearest even), in a few steps instead of n.
//
// The reference walks a voxel column with `vc += zstep` once per layer (yak/src/cuda/tsdf_volume.cu:76), so
// the camera-space position of layer z is the result of z ROUNDED additions, not vc0 + z * zstep.  A z-slab
// that starts at layer z0 has to reproduce those z0 roundings exactly.  Doing them one by one costs z0
// dependent additions per column per frame -- for the last rank of a 1024-layer volume, more than the rank's
// own 128 layers of work.  This routine gives the same bits in O(number of binades crossed):
//
// while x stays in one binade [2^e, 2^(e+1)) its ulp u is constant and x is an integer multiple X of u.  With
// s / u = q + r (q integer, 0 <= r < 1) the rounded sum is X + q (r < 1/2) or X + q + 1 (r > 1/2), and on a tie
// the even one of the two: the same integer increment D at every step (after at most one step that makes X
// even).  So k steps inside the binade are X + k * D, exactly.  The steps that leave a binade, cross zero or
// touch subnormals are done as real additions.
//
// Plain C++ (no HIP types): the device code includes it with YAK_FFWD_FN = __device__ __forceinline__, the CPU
// unit test (tests/cpp/test_ffwd.cpp) compiles it with g++ and checks it bit for bit against the loop.
#ifndef YAK_FFWD_HPP_
#define YAK_FFWD_HPP_

#include <stdint.h>
#include <string.h>

#ifndef YAK_FFWD_FN
#define YAK_FFWD_FN static inline
#endif

YAK_FFWD_FN uint32_t yak_ffwd_bits(float v)
{
  uint32_t b;
  memcpy(&b, &v, 4);
  return b;
}

YAK_FFWD_FN float yak_ffwd_float(uint32_t b)
{
  float v;
  memcpy(&v, &b, 4);
  return v;
}

// x after n additions of s, each rounded to nearest even in fp32.  n >= 0.
YAK_FFWD_FN float yak_ffwd_accumulate(float x, float s, int n)
{
  const uint32_t sb = yak_ffwd_bits(s);
  const int es = (int)((sb >> 23) & 0xffu);
  // the step as a signed 24-bit integer times 2^(es - 150); a zero, subnormal, infinite or NaN step takes
  // the plain loop below
  if (n > 0 && s == 0.f)
    return x + s;  // one addition settles the sign of a zero; after that x + 0 = x
  const bool step_ok = es >= 1 && es <= 254;
  const int32_t Ms = (int32_t)((sb & 0x7fffffu) | 0x800000u) * ((sb >> 31) ? -1 : 1);
  while (n > 0)
  {
    const uint32_t xb = yak_ffwd_bits(x);
    const int ex = (int)((xb >> 23) & 0xffu);
    const int sh = ex - es;
    // real additions where the integer picture does not hold: x zero / subnormal / tiny (its ulp would be
    // subnormal) / infinite / NaN, or a step with a larger exponent than x
    if (!step_ok || ex < 30 || ex > 254 || sh < 0)
    {
      if (ex == 255)
        return (x + s) + s;  // inf / NaN: settled after at most two additions
      x = x + s;
      --n;
      continue;
    }
    if (sh > 26)
      return x;  // |s| < ulp(x) / 8: every sum rounds back to x (also just below a power of two, where the
                 // spacing is ulp / 2)
    const int32_t X = (int32_t)((xb & 0x7fffffu) | 0x800000u) * ((xb >> 31) ? -1 : 1);
    if ((xb & 0x7fffffu) == 0 && ((xb ^ sb) >> 31))
    {  // x is a power of two and the step shrinks it: below x the spacing is u / 2, not u
      x = x + s;
      --n;
      continue;
    }
    const int32_t q = Ms >> sh;  // floor(s / u)
    int32_t D;
    if (sh == 0)
      D = q;
    else
    {
      const int32_t rem = Ms & ((1 << sh) - 1), half = 1 << (sh - 1);
      if (rem < half)
        D = q;
      else if (rem > half)
        D = q + 1;
      else if (X & 1)
      {  // a tie from an odd X: this one step goes to the even neighbour, then X stays even
        x = x + s;
        --n;
        continue;
      }
      else
        D = q + (q & 1);  // a tie from an even X: the even one of X + q, X + q + 1
    }
    if (D == 0)
      return x;
    // steps j = 1..k with 2^23 + 1 <= |X + j * D| <= 2^24 - 1 (sum and rounding stay in this binade)
    const int32_t aX = X < 0 ? -X : X, aD = D < 0 ? -D : D;
    const int32_t room = ((X < 0) == (D < 0)) ? 0xffffff - aX : aX - 0x800001;
    int32_t k = 0;
    if (room >= aD)
    {
      k = (int32_t)((float)room / (float)aD);  // both below 2^24: exact operands, correctly rounded quotient
      if (k * (int64_t)aD > room)
        --k;
      if (k > n)
        k = n;
    }
    if (k <= 0)
    {
      x = x + s;
      --n;
      continue;
    }
    const int32_t Xn = X + k * D;  // |Xn| < 2^24
    // Xn * 2^(ex - 150): (float)Xn is exact, the scale is a normal power of two (ex >= 30)
    x = (float)Xn * yak_ffwd_float((uint32_t)(ex - 150 + 127) << 23);
    n -= k;
    // the jump stops where the next sum would leave the binade (or n is used up): that step is a real
    // addition anyway, so take it here instead of going round the loop for it
    if (n > 0)
    {
      x = x + s;
      --n;
    }
  }
  return x;
}

#endif  // YAK_FFWD_HPP_

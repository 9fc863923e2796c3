// yak_ffwd_accumulate (yak_amd/csrc/ffwd.hpp) against the loop it replaces, bit for bit: n rounded fp32
// additions x = x + s, as the reference performs them along a voxel column (tsdf_volume.cu:76).
#include "../../yak_amd/csrc/ffwd.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

static float loop(float x, float s, int n)
{
  volatile float v = x;  // no contraction, no x87 excess precision
  for (int i = 0; i < n; ++i)
    v = v + s;
  return v;
}

static long g_cases = 0, g_bad = 0;

static void check(float x, float s, int n)
{
  const float want = loop(x, s, n), got = yak_ffwd_accumulate(x, s, n);
  ++g_cases;
  if (yak_ffwd_bits(want) != yak_ffwd_bits(got) && !(want != want && got != got))
  {
    if (++g_bad <= 20)
      std::printf("MISMATCH x=%a s=%a n=%d: loop %a (%08x), ffwd %a (%08x)\n", x, s, n, want, yak_ffwd_bits(want), got,
                  yak_ffwd_bits(got));
  }
}

int main()
{
  std::mt19937_64 rng(12345);
  auto uni = [&](double a, double b) { return a + (b - a) * (double)(rng() >> 11) / 9007199254740992.0; };
  auto rnd_float = [&](int emin, int emax) {  // random sign, exponent in [emin, emax], random mantissa
    const uint32_t e = (uint32_t)(emin + (int)(rng() % (uint64_t)(emax - emin + 1)));
    return yak_ffwd_float((uint32_t)(rng() & 1) << 31 | e << 23 | (uint32_t)(rng() & 0x7fffffu));
  };

  // 1. the application's range: camera-space coordinates of a few metres, steps of a voxel size times a
  //    rotation entry, up to 2048 layers
  for (int i = 0; i < 400000; ++i)
    check((float)uni(-8, 8), (float)(uni(-1, 1) * uni(0.001, 0.02)), (int)(rng() % 2049));
  // 2. every exponent relation between x and s, including steps far below and far above x
  for (int i = 0; i < 600000; ++i)
  {
    const int ex = 60 + (int)(rng() % 130);
    const float x = rnd_float(ex, ex);
    const int es = ex - 30 + (int)(rng() % 36);
    check(x, rnd_float(es < 1 ? 1 : es, es < 1 ? 1 : es), (int)(rng() % 1500));
  }
  // 3. ties: steps that are odd multiples of half an ulp of x (and of a quarter, an eighth)
  for (int i = 0; i < 300000; ++i)
  {
    const int ex = 100 + (int)(rng() % 50);
    const float x = rnd_float(ex, ex);
    const float ulp = std::ldexp(1.0f, ex - 150);
    const int odd = 2 * (int)(rng() % 2000) + 1, div = 1 << (1 + (int)(rng() % 3));
    const float s = ((rng() & 1) ? -1.f : 1.f) * (float)odd * ulp / (float)div;
    check(x, s, (int)(rng() % 3000));
  }
  // 4. zero crossings and walks through the small binades
  for (int i = 0; i < 300000; ++i)
  {
    const float s = rnd_float(90, 125);
    const float x = -s * (float)uni(0, 600);
    check(x, s, (int)(rng() % 1200));
  }
  // 5. the edges: zeros of either sign, subnormals, tiny, huge, infinities, NaN, n = 0
  const float specials[] = { 0.f, -0.f, 1e-45f, -1e-45f, 1e-39f, 1.17549435e-38f, 3e-38f, 1e-30f, 1.f, -1.f, 8388608.f, 16777216.f,
                             16777215.f, 3.4e38f, -3.4e38f, INFINITY, -INFINITY, NAN, 0.5f, 1.5f, 0x1.fffffep22f };
  for (float x : specials)
    for (float s : specials)
      for (int n : { 0, 1, 2, 3, 7, 100, 1025 })
        check(x, s, n);
  for (int i = 0; i < 200000; ++i)
    check(rnd_float(1, 40), rnd_float(1, 40), (int)(rng() % 300));    // tiny against tiny
  for (int i = 0; i < 100000; ++i)
    check(rnd_float(0, 254), rnd_float(0, 254), (int)(rng() % 64));   // anything against anything, few steps
  // 6. powers of two (the spacing changes exactly there) with steps of either sign around their ulp
  for (int i = 0; i < 300000; ++i)
  {
    const int ex = 40 + (int)(rng() % 180);
    const float x = ((rng() & 1) ? -1.f : 1.f) * std::ldexp(1.0f, ex - 127);
    const int es = ex - 28 + (int)(rng() % 32);
    check(x, rnd_float(es < 1 ? 1 : es, es < 1 ? 1 : es), (int)(rng() % 700));
    const float ulp = std::ldexp(1.0f, ex - 150);
    check(x, ((rng() & 1) ? -1.f : 1.f) * ulp * (float)(1 + rng() % 9) / (float)(1 << (rng() % 5)), (int)(rng() % 700));
  }
  // 7. overflow to infinity
  for (int i = 0; i < 20000; ++i)
    check(rnd_float(250, 254), rnd_float(245, 254), (int)(rng() % 200));

  std::printf("%ld cases, %ld mismatches\n", g_cases, g_bad);
  if (g_bad == 0)
    std::printf("PASSED\n");
  return g_bad ? 1 : 0;
}

// rt_fastdiv.h -- division by a divisor that is the same for a whole launch, prepared on the host.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define RT_FASTDIV_FN __host__ __device__ inline
#else
#define RT_FASTDIV_FN inline
#endif

namespace rt {

// Division of n < 2^31 by d >= 1: with L = ceil(log2 d) and m = floor(2^(31 + L) / d) + 1 (which fits 32 bits),
// floor(n / d) = (n * m) >> (31 + L) exactly -- the error term n * (m * d - 2^(31 + L)) / (d * 2^(31 + L)) stays below
// 1 / d because n < 2^31 and m * d - 2^(31 + L) <= d <= 2^L.  One multiply (its upper half) and a shift instead of the
// twenty-odd instructions of the general 32-bit division.
struct FastDiv {
    uint32_t d, m, sh;   // q = mulhi(n, m) >> sh with sh = L - 1; d == 1: q = n
};
inline FastDiv fast_div_make(uint32_t d)
{
    FastDiv f{d, 0u, 0u};
    if (d < 2u) return f;
    uint32_t L = 0;
    while ((1ull << L) < d) ++L;
    f.m = (uint32_t)((1ull << (31 + L)) / d + 1ull);
    f.sh = L - 1u;
    return f;
}
RT_FASTDIV_FN uint32_t fast_div(uint32_t n, const FastDiv& f)
{
    const uint32_t q = (uint32_t)(((uint64_t)n * f.m) >> 32) >> f.sh;
    return f.d == 1u ? n : q;
}

} // namespace rt

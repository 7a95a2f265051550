// Host check of rt_fastdiv.h: fast_div(n, fast_div_make(d)) == n / d for n < 2^31 -- every divisor up to 70 000 and a
// spread of larger ones, each against the numerators around its multiples, the range's ends and random values.
#include <cstdint>
#include <cstdio>
#include <initializer_list>
#include <random>

#include "rt_fastdiv.h"

int main()
{
    std::mt19937_64 rng(1);
    uint64_t checked = 0;
    auto check = [&](uint32_t d) -> bool {
        const rt::FastDiv f = rt::fast_div_make(d);
        auto one = [&](uint64_t n) -> bool {
            if (n >= (1ull << 31)) return true;
            ++checked;
            if (rt::fast_div((uint32_t)n, f) == (uint32_t)n / d) return true;
            std::printf("FAIL: %u / %u = %u, fast_div says %u\n", (uint32_t)n, d, (uint32_t)n / d, rt::fast_div((uint32_t)n, f));
            return false;
        };
        const uint64_t top = (1ull << 31) - 1;
        for (uint64_t n : std::initializer_list<uint64_t>{0, 1, (uint64_t)d - 1, (uint64_t)d, (uint64_t)d + 1, top, top - 1, top / d * d, top / d * d - 1})
            if (!one(n)) return false;
        for (int k = 0; k < 64; ++k) {
            const uint64_t q = rng() % (top / d + 1);
            for (uint64_t n : std::initializer_list<uint64_t>{q * d, q * d + d - 1, q * d - 1, rng() % (top + 1)})
                if (n <= top && !one(n)) return false;
        }
        return true;
    };
    for (uint32_t d = 1; d <= 70000; ++d)
        if (!check(d)) return 1;
    for (int k = 0; k < 200000; ++k)
        if (!check((uint32_t)(rng() % ((1ull << 31) - 1)) + 1u)) return 1;
    for (uint32_t L = 1; L < 31; ++L)
        for (uint32_t d : {(1u << L) - 1u, 1u << L, (1u << L) + 1u})
            if (!check(d)) return 1;
    std::printf("ok: %llu divisions\n", (unsigned long long)checked);
    return 0;
}

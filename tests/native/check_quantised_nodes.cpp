// Host-only check of the quantised four-way nodes (raytracer_amd/csrc/bvh_build.cpp, compiled in here): every decoded
// child box must contain the exact one with room for the kernels' decode rounding, child references must be kept, unused
// slots must stay unused.  Usage: check_quantised_nodes <number of boxes> <seed> <kind>; kind 0 = small boxes in a cube,
// 1 = thin / flat boxes and large coordinates, 2 = some boxes with infinite or NaN planes.  Prints a summary; exit code 0 = ok.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <random>
#include <vector>

#include "bvh_build.cpp"

using namespace rt;

int main(int argc, char** argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 100000u;
    const unsigned seed = argc > 2 ? (unsigned)std::atoi(argv[2]) : 1u;
    const int kind = argc > 3 ? std::atoi(argv[3]) : 0;
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> u(0.0f, 1.0f);
    std::vector<PrimBox> boxes(n);
    for (uint32_t i = 0; i < n; ++i) {
        const float scale = kind == 1 ? 5000.0f : 10.0f;
        for (int a = 0; a < 3; ++a) {
            const float c = (u(rng) - 0.5f) * scale + (kind == 1 ? 12345.0f : 0.0f);
            float e = u(rng) * 0.02f;
            if (kind == 1 && (rng() % 3) == 0) e = 0.0f;            // flat on this axis
            boxes[i].lo[a] = c - e;
            boxes[i].hi[a] = c + e;
        }
        if (kind == 2 && (rng() % 1000) == 0) boxes[i].hi[rng() % 3] = std::numeric_limits<float>::infinity();
        if (kind == 2 && (rng() % 1000) == 1) boxes[i].lo[rng() % 3] = std::numeric_limits<float>::quiet_NaN();
    }
    BuildParams bp;
    bp.threads = 4;
    const BuiltTree bt = build_tree(boxes.data(), n, bp);
    uint32_t depth4 = 0;
    const std::vector<Node128> four = collapse_tree4(bt.nodes, depth4);
    const std::vector<NodeQ64> q = quantize_tree4(four, 4);
    if (q.size() != four.size()) { std::printf("size mismatch\n"); return 1; }
    double worstSlack = 0.0;   // largest widening of a plane, in units of the node's extent
    size_t planes = 0, wide = 0;
    for (size_t i = 0; i < four.size(); ++i) {
        const Node128& f = four[i];
        const NodeQ64& g = q[i];
        const float step[3] = {g.sx, g.sy, g.sz};
        const uint32_t words[3][2] = {{g.x01, g.x23}, {g.y01, g.y23}, {g.z01, g.z23}};
        for (int k = 0; k < 4; ++k) {
            if (g.c[k] != f.c[k]) { std::printf("node %zu child %d: reference changed\n", i, k); return 1; }
            if (f.c[k] == kEmptyChild) continue;
            for (int a = 0; a < 3; ++a) {
                const uint32_t w = words[a][k >> 1] >> ((k & 1) * 16);
                const double ql = w & 255u, qh = (w >> 8) & 255u;
                const double lo = (double)g.origin[a] + ql * (double)step[a], hi = (double)g.origin[a] + qh * (double)step[a];
                const double el = f.box[k][2 * a], eh = f.box[k][2 * a + 1];
                if (!std::isfinite(el) || !std::isfinite(eh)) {
                    if (!(lo < -1e38 && hi > 1e38)) { std::printf("node %zu child %d axis %d: a non-finite box must decode to everything\n", i, k, a); return 1; }
                    ++wide;
                    continue;
                }
                if (lo < -1e38 && hi > 1e38) { ++wide; continue; }   // a sibling is not finite on this axis
                double nlo = 1e300, nhi = -1e300;
                for (int m = 0; m < 4; ++m)
                    if (f.c[m] != kEmptyChild) { nlo = std::min(nlo, (double)f.box[m][2 * a]); nhi = std::max(nhi, (double)f.box[m][2 * a + 1]); }
                const double extent = nhi - nlo, margin = std::ldexp(extent, -22);
                // planes on the grid's first point decode exactly (q = 0 adds nothing); all others keep the margin
                const bool okLo = ql == 0 ? lo <= el : lo <= el - margin;
                const bool okHi = hi >= eh + margin || (extent == 0.0 && hi == eh);
                if (!okLo || !okHi) {
                    std::printf("node %zu child %d axis %d: [%.9g, %.9g] decoded as [%.9g, %.9g] (extent %.9g)\n", i, k, a, el, eh, lo, hi, extent);
                    return 1;
                }
                if (extent > 0.0) worstSlack = std::max(worstSlack, std::max(el - lo, hi - eh) / extent);
                ++planes;
            }
        }
    }
    std::printf("ok: %zu nodes, %zu planes checked, %zu on all-space grids, widest plane moved by %.5f of its node's extent\n", four.size(), planes * 2, wide, worstSlack);
    return worstSlack <= 1.0 / 255.0 * 1.01 + 1e-6 ? 0 : 1;
}

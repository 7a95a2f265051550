// Host-only check of the kernels' box test on quantised nodes (raytracer_amd/csrc/rt_trace_inl.h: node4_hits,
// slab_quantised, plane_offsets), restated here in fp32 with the same operations: a child box must never be rejected
// for a ray that passes through it.
//
//   wide decode   : accepts every ray that, in exact arithmetic, passes through the exact fp32 child box -- at any distance
//                   between the ray's origin and the tree.
//   plain decode  : (used while every ray of a wave starts within 256 units of its tree's farthest plane) accepts every ray
//                   that passes through the child box shrunk by 1e-4 per side, the padding TriangleMesh::getAABB puts around
//                   every triangle (shape/TriangleMesh.cpp:177-191): a hit point lies inside the unpadded box.
//
// Usage: check_quantised_decode <boxes> <seed> <distance>; rays start about <distance> units from the boxes they are aimed
// at (at corners, edges and faces, where the test is tight).  Exit code 0 = no rejection that should not be.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "bvh_build.cpp"

using namespace rt;

namespace {

struct Ray { float o[3], d[3]; };

// slab_quantised + node4_hits for one child, as the kernel computes it (fmaf = the packed fma per half; fmaxf / fminf drop a
// NaN operand like v_max3 / v_min3)
bool decoded_hit(const NodeQ64& g, int k, const Ray& r, bool wide)
{
    const float step[3] = {g.sx, g.sy, g.sz};
    const uint32_t words[3][2] = {{g.x01, g.x23}, {g.y01, g.y23}, {g.z01, g.z23}};
    float tn[3], tf[3];
    for (int a = 0; a < 3; ++a) {
        const float iv = 1.0f / r.d[a];
        const float b = (g.origin[a] - r.o[a]) * iv;
        const float s = step[a] * iv;
        float bn = b, bf = b;
        if (wide) {
            const float kk = 4.76837158203125e-07f;
            bn = fmaf(-fabsf(b), kk, b);
            bf = fmaf(fabsf(b), kk, b);
        }
        const uint32_t w = words[a][k >> 1] >> ((k & 1) * 16);
        float ql = (float)(w & 255u), qh = (float)((w >> 8) & 255u);
        if (std::signbit(iv)) std::swap(ql, qh);      // near_first: the plane the ray meets first
        tn[a] = fmaf(ql, s, bn);
        tf[a] = fmaf(qh, s, bf);
    }
    const float tnear = fmaxf(fmaxf(fmaxf(tn[0], tn[1]), tn[2]), 0.0f);
    const float tfar = fminf(fminf(fminf(tf[0], tf[1]), tf[2]), INFINITY);
    return tnear <= tfar;
}

// the exact slab test on [lo, hi] with the exact direction, in long double
bool exact_hit(const double lo[3], const double hi[3], const Ray& r)
{
    long double tnear = 0.0L, tfar = INFINITY;
    for (int a = 0; a < 3; ++a) {
        const long double o = r.o[a], d = r.d[a];
        if (d == 0.0L) {
            if (o < lo[a] || o > hi[a]) return false;
            continue;
        }
        long double t1 = ((long double)lo[a] - o) / d, t2 = ((long double)hi[a] - o) / d;
        if (t1 > t2) std::swap(t1, t2);
        tnear = std::max(tnear, t1);
        tfar = std::min(tfar, t2);
    }
    return tnear <= tfar;
}

} // namespace

int main(int argc, char** argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 20000u;
    const unsigned seed = argc > 2 ? (unsigned)std::atoi(argv[2]) : 1u;
    const double dist = argc > 3 ? std::atof(argv[3]) : 10.0;
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> u(0.0f, 1.0f);
    std::vector<PrimBox> boxes(n);
    for (uint32_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            const float c = (u(rng) - 0.5f) * 10.0f;
            const float e = u(rng) * 0.02f + 1e-4f;      // padded like a triangle's box
            boxes[i].lo[a] = c - e;
            boxes[i].hi[a] = c + e;
        }
    BuildParams bp;
    bp.threads = 4;
    const BuiltTree bt = build_tree(boxes.data(), n, bp);
    uint32_t depth4 = 0;
    const std::vector<Node128> four = collapse_tree4(bt.nodes, depth4);
    const std::vector<NodeQ64> q = quantize_tree4(four, 4);

    size_t rays = 0, exactHits = 0, shrunkHits = 0, wideMiss = 0, plainMiss = 0, plainMissOfExact = 0;
    const bool plainApplies = dist <= 240.0;      // the kernels use the plain decode only within 256 units of the tree
    for (size_t i = 0; i < four.size(); ++i) {
        const Node128& f = four[i];
        for (int k = 0; k < 4; ++k) {
            if (f.c[k] == kEmptyChild) continue;
            double lo[3], hi[3], slo[3], shi[3];
            bool shrunkOk = true;
            for (int a = 0; a < 3; ++a) {
                lo[a] = f.box[k][2 * a]; hi[a] = f.box[k][2 * a + 1];
                slo[a] = lo[a] + 1e-4; shi[a] = hi[a] - 1e-4;
                if (slo[a] > shi[a]) shrunkOk = false;
            }
            for (int rep = 0; rep < 6; ++rep) {
                // a target on the box: every coordinate at the low plane, the high plane or inside (corners, edges, faces),
                // moved by a few ulps either way so that the ray grazes the box from inside or outside
                float target[3];
                for (int a = 0; a < 3; ++a) {
                    const int where = (int)(rng() % 3);
                    float v = where == 0 ? (float)lo[a] : where == 1 ? (float)hi[a] : (float)(lo[a] + (hi[a] - lo[a]) * u(rng));
                    for (int m = (int)(rng() % 5) - 2; m != 0; m += m < 0 ? 1 : -1) v = std::nextafter(v, m < 0 ? -INFINITY : INFINITY);
                    target[a] = v;
                }
                Ray r;
                double dir[3], len = 0;
                for (int a = 0; a < 3; ++a) { dir[a] = u(rng) - 0.5; len += dir[a] * dir[a]; }
                if (rng() % 8 == 0) dir[rng() % 3] = 0.0;      // rays along an axis' planes
                len = std::sqrt(len);
                if (!(len > 0)) continue;
                for (int a = 0; a < 3; ++a) r.o[a] = (float)(target[a] - dir[a] / len * dist);
                double l2 = 0;
                for (int a = 0; a < 3; ++a) { r.d[a] = (float)(target[a] - (double)r.o[a]); l2 += (double)r.d[a] * r.d[a]; }
                l2 = std::sqrt(l2);
                if (!(l2 > 0)) continue;
                for (int a = 0; a < 3; ++a) r.d[a] = (float)(r.d[a] / l2);
                ++rays;
                const bool ex = exact_hit(lo, hi, r);
                const bool sh = shrunkOk && exact_hit(slo, shi, r);
                const bool dw = decoded_hit(q[i], k, r, true), dp = decoded_hit(q[i], k, r, false);
                exactHits += ex;
                shrunkHits += sh;
                if (ex && !dw) {
                    if (wideMiss < 5) std::printf("wide decode rejects node %zu child %d: o = (%.9g, %.9g, %.9g) d = (%.9g, %.9g, %.9g)\n", i, k, r.o[0], r.o[1], r.o[2], r.d[0], r.d[1], r.d[2]);
                    ++wideMiss;
                }
                if (sh && !dp) ++plainMiss;
                if (ex && !dp) ++plainMissOfExact;
            }
        }
    }
    std::printf("%s: %zu rays at distance %g, %zu pass through the exact box (%zu through the shrunk one); rejected: wide decode %zu, plain decode %zu of the shrunk (%zu of the exact)\n",
                (wideMiss == 0 && (!plainApplies || plainMiss == 0)) ? "ok" : "FAILED", rays, dist, exactHits, shrunkHits, wideMiss, plainMiss, plainMissOfExact);
    return (wideMiss == 0 && (!plainApplies || plainMiss == 0)) ? 0 : 1;
}

#include "rt_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace rt {

namespace {

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; a++) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -lo[a]; }
    }
    void grow(const Box& b) {
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); }
    }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

constexpr int kBins = 16;

struct Builder {
    std::vector<Box> boxes;          // per input sphere, padded outwards
    std::vector<float> centroid;     // per input sphere x3
    std::vector<int32_t> order;
    std::vector<float> nodes;
    int max_depth = 44;
    int depth_seen = 0;
    int64_t sweep_max = 0;           // ranges up to this size are split by the exact sweep, larger ones by bins
    int leaf_max = 4;                // a range of at most this many spheres may become a leaf
    float node_cost = 0.0f;          // cost of visiting an inner node in units of one sphere test; 0: split down to leaf_max

    static int ceil_log2(int64_t v) {
        int l = 0;
        while ((int64_t(1) << l) < v) l++;
        return l;
    }

    int32_t leaf_code(int64_t first, int64_t count) const {
        return ~(int32_t)((first << kLeafBits) | (count - 1));
    }

    // Exact surface-area-heuristic split of order[first, first+count): the range is sorted by
    // centroid along each axis in turn and every position is priced; the range is left sorted
    // along the winning axis.  Returns the first index of the right part, or -1.
    int64_t sweep_split(int64_t first, int64_t count) {
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1;
        int64_t best_k = -1;
        std::vector<float> right_area((size_t)count);
        for (int a = 0; a < 3; a++) {
            std::sort(order.begin() + first, order.begin() + first + count, [&](int32_t x, int32_t y) {
                float cx = centroid[3 * (size_t)x + a], cy = centroid[3 * (size_t)y + a];
                return cx < cy || (cx == cy && x < y);
            });
            Box acc; acc.reset();
            for (int64_t k = count - 1; k > 0; k--) { acc.grow(boxes[order[first + k]]); right_area[(size_t)k] = acc.half_area(); }
            acc.reset();
            for (int64_t k = 1; k < count; k++) {
                acc.grow(boxes[order[first + k - 1]]);
                float cost = acc.half_area() * (float)k + right_area[(size_t)k] * (float)(count - k);
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_k = k; }
            }
        }
        if (best_axis < 0) return -1;
        if (best_axis != 2)
            std::sort(order.begin() + first, order.begin() + first + count, [&](int32_t x, int32_t y) {
                float cx = centroid[3 * (size_t)x + best_axis], cy = centroid[3 * (size_t)y + best_axis];
                return cx < cy || (cx == cy && x < y);
            });
        return first + best_k;
    }

    // Builds the subtree over order[first, first+count); returns its child code and bounds.
    int32_t build(int64_t first, int64_t count, int depth, Box& bounds) {
        bounds.reset();
        for (int64_t i = first; i < first + count; i++) bounds.grow(boxes[order[i]]);
        if (count <= leaf_max && (node_cost <= 0.0f || count <= 2)) return leaf_code(first, count);

        float clo[3], chi[3];
        for (int a = 0; a < 3; a++) { clo[a] = std::numeric_limits<float>::infinity(); chi[a] = -clo[a]; }
        for (int64_t i = first; i < first + count; i++)
            for (int a = 0; a < 3; a++) {
                float c = centroid[3 * (size_t)order[i] + a];
                clo[a] = std::min(clo[a], c); chi[a] = std::max(chi[a], c);
            }

        int64_t mid = -1;
        // small ranges: the exact sweep (every split position on every axis) instead of bins
        if (count <= sweep_max && depth + ceil_log2((count + leaf_max - 1) / leaf_max) + 1 < max_depth) {
            mid = sweep_split(first, count);
        }
        bool balanced = mid > first || depth + ceil_log2((count + leaf_max - 1) / leaf_max) + 1 >= max_depth;   // "balanced": skip the binned search
        if (!balanced) {
            float best_cost = std::numeric_limits<float>::infinity();
            int best_axis = -1, best_bin = -1;
            for (int a = 0; a < 3; a++) {
                float ext = chi[a] - clo[a];
                if (!(ext > 0.0f)) continue;
                Box bb[kBins]; int64_t cnt[kBins];
                for (int b = 0; b < kBins; b++) { bb[b].reset(); cnt[b] = 0; }
                float scale = (float)kBins / ext;
                for (int64_t i = first; i < first + count; i++) {
                    int b = (int)((centroid[3 * (size_t)order[i] + a] - clo[a]) * scale);
                    b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                    bb[b].grow(boxes[order[i]]); cnt[b]++;
                }
                float right_area[kBins]; int64_t right_cnt[kBins];
                Box acc; acc.reset(); int64_t n = 0;
                for (int b = kBins - 1; b > 0; b--) {
                    acc.grow(bb[b]); n += cnt[b];
                    right_area[b] = n ? acc.half_area() : 0.0f; right_cnt[b] = n;
                }
                acc.reset(); n = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    acc.grow(bb[b]); n += cnt[b];
                    if (n == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = acc.half_area() * (float)n + right_area[b + 1] * (float)right_cnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                }
            }
            // surface-area heuristic: a leaf costs one sphere test per sphere; splitting costs the
            // node visit plus each side's tests weighted by the chance of entering it
            if (count <= leaf_max && node_cost > 0.0f && best_axis >= 0) {
                float split_cost = node_cost + best_cost / std::max(bounds.half_area(), 1e-30f);
                if ((float)count <= split_cost) return leaf_code(first, count);
            }
            if (count <= leaf_max && best_axis < 0) return leaf_code(first, count);
            if (best_axis >= 0) {
                float ext = chi[best_axis] - clo[best_axis];
                float scale = (float)kBins / ext;
                auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](int32_t id) {
                    int b = (int)((centroid[3 * (size_t)id + best_axis] - clo[best_axis]) * scale);
                    b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                    return b <= best_bin;
                });
                mid = it - order.begin();
            }
        }
        if (mid <= first || mid >= first + count) {
            // median split on the widest centroid axis (also the depth-bounded fallback);
            // ties broken by sphere index so the build is deterministic
            int axis = 0;
            for (int a = 1; a < 3; a++) if (chi[a] - clo[a] > chi[axis] - clo[axis]) axis = a;
            mid = first + count / 2;
            std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                             [&](int32_t x, int32_t y) {
                                 float cx = centroid[3 * (size_t)x + axis], cy = centroid[3 * (size_t)y + axis];
                                 return cx < cy || (cx == cy && x < y);
                             });
        }

        int32_t me = (int32_t)(nodes.size() / 16);
        nodes.resize(nodes.size() + 16);
        depth_seen = std::max(depth_seen, depth + 1);
        Box lb, rb;
        int32_t lc = build(first, mid - first, depth + 1, lb);
        int32_t rc = build(mid, first + count - mid, depth + 1, rb);
        float* q = nodes.data() + 16 * (size_t)me;
        q[0] = lb.lo[0]; q[1] = lb.lo[1]; q[2] = lb.lo[2]; q[3] = lb.hi[0];
        q[4] = lb.hi[1]; q[5] = lb.hi[2]; q[6] = rb.lo[0]; q[7] = rb.lo[1];
        q[8] = rb.lo[2]; q[9] = rb.hi[0]; q[10] = rb.hi[1]; q[11] = rb.hi[2];
        std::memcpy(q + 12, &lc, 4);
        std::memcpy(q + 13, &rc, 4);
        q[14] = 0.0f; q[15] = 0.0f;
        return me;
    }
};

}  // namespace

BvhResult build_bvh(const double* centers, const double* radii, int64_t n, int max_depth, int leaf_max, float node_cost) {
    BvhResult res;
    if (n <= 0) return res;
    Builder b;
    b.max_depth = max_depth;
    b.leaf_max = leaf_max < 1 ? 1 : (leaf_max > kLeafCap ? kLeafCap : leaf_max);
    b.node_cost = node_cost;
    // the sweep sorts each range three times: worth it where the build is over in milliseconds
    b.sweep_max = n <= 65536 ? 4096 : 0;
    b.boxes.resize((size_t)n);
    b.centroid.resize(3 * (size_t)n);
    b.order.resize((size_t)n);
    const float ninf = -std::numeric_limits<float>::infinity(), pinf = -ninf;
    for (int64_t i = 0; i < n; i++) {
        // cuda/sphere.h:30-33 (centre -+ radius), padded so that single-precision rounding in
        // the slab test can never cut off a sphere the double-precision scan would hit
        double r = std::fabs(radii[i]) * (1.0 + 1e-5) + 1e-30;
        for (int a = 0; a < 3; a++) {
            double c = centers[3 * i + a];
            b.boxes[i].lo[a] = std::nextafter((float)(c - r), ninf);
            b.boxes[i].hi[a] = std::nextafter((float)(c + r), pinf);
            b.centroid[3 * (size_t)i + a] = (float)c;
        }
        b.order[i] = (int32_t)i;
    }
    b.nodes.reserve((size_t)n * 8);
    Box bounds;
    res.root = b.build(0, n, 0, bounds);
    res.depth = b.depth_seen;
    res.nodes = std::move(b.nodes);
    res.order = std::move(b.order);
    return res;
}

}  // namespace rt

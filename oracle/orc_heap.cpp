// oracle/orc_heap.cpp -- TEST INFRASTRUCTURE ONLY (part of liborc.so).
// WeightedFilter::leastLikely (src/beliefs/particle_filters/WeightedFilter.cpp:193-238) restated with the same standard
// container: a std::priority_queue of (weight, index) under a comparator that looks at the weight only, filled with the
// first n elements, then every element (the first n again) replaces the top if it is lighter; the n indices are popped
// off the top.  With ties -- the incubator belief only ever asks this of uniform weights -- the order that comes out is
// whatever libstdc++'s heap does, so it is asked of libstdc++'s heap.
#include <queue>
#include <utility>
#include <vector>

namespace {
using queue_elements = std::pair<double, int>;
struct Less {
    bool operator()(queue_elements l, queue_elements r) const { return l.first < r.first; }
};
}  // namespace

extern "C" void orc_least_likely(const double* w, int size, int n, int* out)
{
    std::priority_queue<queue_elements, std::vector<queue_elements>, Less> q;
    int i = 0;
    for (; i < n; ++i) q.push({w[i], i});
    for (i = 0; i < size; ++i) {
        if (w[i] < q.top().first) {
            q.pop();
            q.push({w[i], i});
        }
    }
    for (i = 0; i < n; ++i) {
        out[i] = q.top().second;
        q.pop();
    }
}

// Host-side inner loops of the minibatch selectors (tike_amd/cluster.py).
//
// The reference selects minibatches with two quadratic procedures
// (src/tike/cluster.py:302-377 "wobbly center", :465-637 "compact"): a
// farthest-point round robin and a pairwise-swap refinement, each step of
// which is a pass over all N positions.  As NumPy expressions that is five
// temporaries and a Python iteration per step (2.8 s / 3.5 s for 10 000
// positions, minutes for the 80 000 of a node-sized job); here the same
// steps run as single fused passes.  The arithmetic is the arithmetic of the
// NumPy expressions they replace -- float32 differences, squares, sums and
// square roots in that order, sequential float32 sums for the means, float64
// for the swap gains, first index wins every tie -- so the labels are the
// same labels (tests/test_host_golden_cpu.py pins them to reference-run
// fixtures).  Built with -ffp-contract=off: a fused multiply-add would round
// differently from NumPy.  No GPU involved; plain C ABI like the rest of
// include/tike_amd.h.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/tike_amd.h"

namespace {

constexpr int kBadArgument = TIKE_ERR_ARG;

// float32 mean of the rows of `points` owned by `cluster`, summed in index
// order (what ndarray.mean(axis=0) does for an (m, 2) float32 array)
inline void mean_of(const float* points, const int64_t* owner, int64_t n, int64_t cluster,
                    float* cx, float* cy) {
  float sx = 0.0f, sy = 0.0f;
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (owner[i] == cluster) {
      sx += points[2 * i];
      sy += points[2 * i + 1];
      ++m;
    }
  }
  *cx = sx / (float)m;
  *cy = sy / (float)m;
}

}  // namespace

extern "C" int tike_cluster_farthest_fill(const float* points, long n, long* owner_,
                                          int num_cluster, long turns) {
  if (!points || !owner_ || n < 1 || num_cluster < 1 || turns < 0) return kBadArgument;
  static_assert(sizeof(long) == sizeof(int64_t), "LP64");
  int64_t* owner = reinterpret_cast<int64_t*>(owner_);
  // the free points, ascending: the first of several farthest ones wins
  std::vector<int64_t> free_;
  free_.reserve(n);
  for (int64_t i = 0; i < n; ++i)
    if (owner[i] < 0) free_.push_back(i);
  if ((int64_t)free_.size() < turns) return kBadArgument;
  for (long turn = 0; turn < turns; ++turn) {
    const int64_t cluster = turn % num_cluster;
    float cx, cy;
    mean_of(points, owner, n, cluster, &cx, &cy);
    float reach = -1.0f;
    size_t at = 0;
    for (size_t f = 0; f < free_.size(); ++f) {
      const float dx = points[2 * free_[f]] - cx;
      const float dy = points[2 * free_[f] + 1] - cy;
      const float d = std::sqrt(dx * dx + dy * dy);
      if (d > reach) {
        reach = d;
        at = f;
      }
    }
    owner[free_[at]] = cluster;
    free_.erase(free_.begin() + at);
  }
  return 0;
}

extern "C" int tike_cluster_swap_sweep(const double* dist, long n, int k, long* label_,
                                       const long* best_, const long* order_, double* regret,
                                       int* moved) {
  if (!dist || !label_ || !best_ || !order_ || !regret || !moved || n < 1 || k < 1)
    return kBadArgument;
  int64_t* label = reinterpret_cast<int64_t*>(label_);
  const int64_t* best = reinterpret_cast<const int64_t*>(best_);
  const int64_t* order = reinterpret_cast<const int64_t*>(order_);
  *moved = 0;
  // column-major copy of the distances and each point's distance to its own
  // centroid: the pass over q below then reads three contiguous streams
  std::vector<double> col((size_t)n * k), own_d(n), gain(n);
  for (int64_t i = 0; i < n; ++i) {
    for (int c = 0; c < k; ++c) col[(size_t)c * n + i] = dist[i * k + c];
    own_d[i] = dist[i * k + label[i]];
  }
  for (long o = 0; o < n; ++o) {
    const int64_t p = order[o];
    if (!(regret[p] < 0)) continue;
    const int64_t home = label[p];
    const double* dp = dist + p * k;
    const double* to_home = &col[(size_t)home * n];
    const double here = dp[home];
    // gain of exchanging p with q, in the order NumPy evaluates it:
    //   ((d(p, home) + d(q, own(q))) - d(p, own(q))) - d(q, home)
    for (int64_t i = 0; i < n; ++i) gain[i] = ((here + own_d[i]) - dp[label[i]]) - to_home[i];
    double top = 0.0;
    int64_t q = -1;
    for (int64_t i = 0; i < n; ++i) {
      if (gain[i] > top && label[i] != home) {
        top = gain[i];
        q = i;
      }
    }
    if (q < 0) continue;
    *moved = 1;
    label[p] = label[q];
    label[q] = home;
    own_d[p] = dist[p * k + label[p]];
    own_d[q] = dist[q * k + label[q]];
    regret[q] = dist[q * k + best[q]] - own_d[q];
    regret[p] = dist[p * k + best[p]] - own_d[p];
  }
  return 0;
}

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
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/tike_amd.h"

namespace {

constexpr int kBadArgument = TIKE_ERR_ARG;

// d2[f] = squared float32 distance of free point f to (cx, cy) -- -1 for a
// tombstone (NaN coordinates) or a NaN distance, which must never win -- and
// the largest of them.  Non-negative floats order like their bit patterns, so
// the maximum is an INTEGER reduction, which the compiler vectorises without
// any licence to reorder float arithmetic (there is none here to reorder).
__attribute__((target_clones("avx2", "default")))
float squared_distances(const float* __restrict__ px, const float* __restrict__ py, float cx,
                        float cy, float* __restrict__ d2, size_t count) {
  int32_t top = -1;
  for (size_t f = 0; f < count; ++f) {
    const float dx = px[f] - cx, dy = py[f] - cy;
    float d = dx * dx + dy * dy;
    d = d == d ? d : -1.0f;
    d2[f] = d;
    int32_t bits;
    __builtin_memcpy(&bits, &d, sizeof(bits));
    top = bits > top ? bits : top;  // (-1.0f is a negative integer)
  }
  float out;
  __builtin_memcpy(&out, &top, sizeof(out));
  return top < 0 ? -1.0f : out;
}

}  // namespace

// Round 5: the same turns in O(members + free) instead of O(n + free + a
// memmove of the free list) each.  What fixes the labels is kept to the bit:
//   * the mean of a cluster is the float32 sum of its members in ASCENDING
//     index order (what ndarray.mean(axis=0) does for an (m, 2) float32
//     array) -- the members are kept as a sorted list, a newcomer is inserted
//     at its place;
//   * the winner is the FIRST free point whose float32 distance
//     sqrt(dx^2 + dy^2) is the largest: sqrt is monotonic, so that is the
//     first point whose rounded distance equals sqrt(max d^2) -- one
//     vectorisable pass for the largest SQUARED distance, then a scan that
//     takes a square root only of the few candidates within rounding of it;
//   * the free points stay in ascending order: a claimed one becomes a
//     tombstone (NaN coordinates never win a comparison) and the arrays are
//     compacted, order preserved, when half of them are tombstones.
// (80 000 positions in 80 clusters: 8.1 s -> well under a second.)
extern "C" int tike_cluster_farthest_fill(const float* points, long n, long* owner_,
                                          int num_cluster, long turns) {
  if (!points || !owner_ || n < 1 || num_cluster < 1 || turns < 0) return kBadArgument;
  static_assert(sizeof(long) == sizeof(int64_t), "LP64");
  int64_t* owner = reinterpret_cast<int64_t*>(owner_);
  std::vector<std::vector<int64_t>> members(num_cluster);
  std::vector<float> fx, fy, d2;
  std::vector<int64_t> fid;
  fx.reserve(n);
  fy.reserve(n);
  fid.reserve(n);
  for (int64_t i = 0; i < n; ++i) {
    if (owner[i] < 0) {
      fx.push_back(points[2 * i]);
      fy.push_back(points[2 * i + 1]);
      fid.push_back(i);
    } else if (owner[i] < num_cluster) {
      members[owner[i]].push_back(i);  // ascending by construction
    }
  }
  size_t alive = fid.size();
  if ((int64_t)alive < turns) return kBadArgument;
  d2.resize(fid.size());
  const float nan = std::nanf("");
  for (long turn = 0; turn < turns; ++turn) {
    const int64_t cluster = turn % num_cluster;
    std::vector<int64_t>& mine = members[cluster];
    float sx = 0.0f, sy = 0.0f;
    for (const int64_t i : mine) {
      sx += points[2 * i];
      sy += points[2 * i + 1];
    }
    const float cx = sx / (float)mine.size(), cy = sy / (float)mine.size();
    const size_t count = fid.size();
    const float* __restrict__ px = fx.data();
    const float* __restrict__ py = fy.data();
    float* __restrict__ pd = d2.data();
    // (a NaN never wins: tombstones, the mean of an empty cluster)
    const float top = squared_distances(px, py, cx, cy, pd, count);
    size_t at = count;
    if (top >= 0.0f) {
      const float reach = std::sqrt(top);
      const float near = top * 0.99999f;  // wider than any rounding of sqrt
      for (size_t f = 0; f < count; ++f) {
        if (pd[f] >= near && std::sqrt(pd[f]) == reach) {
          at = f;
          break;
        }
      }
    }
    if (at == count) {  // no distance compared greater than -1: the first free point
      for (size_t f = 0; f < count; ++f) {
        if (fid[f] >= 0) {
          at = f;
          break;
        }
      }
    }
    const int64_t winner = fid[at];
    owner[winner] = cluster;
    mine.insert(std::upper_bound(mine.begin(), mine.end(), winner), winner);
    fid[at] = -1;
    fx[at] = fy[at] = nan;
    --alive;
    if (alive * 2 < fid.size() && fid.size() > 64) {
      size_t w = 0;
      for (size_t f = 0; f < fid.size(); ++f) {
        if (fid[f] >= 0) {
          fid[w] = fid[f];
          fx[w] = fx[f];
          fy[w] = fy[f];
          ++w;
        }
      }
      fid.resize(w);
      fx.resize(w);
      fy.resize(w);
    }
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

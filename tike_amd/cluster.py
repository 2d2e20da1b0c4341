"""Host-side batch selection (reference src/tike/cluster.py).

Once-per-run host code, out of the accelerated scope; restated so that
``reconstruct`` accepts the reference's ``batch_method`` names.
"""
import ctypes

import numpy as np


def _native():
    """The C entries of csrc/cluster_host.cpp (loaded with the library)."""
    from ._lib import check, lib
    return lib, check


def _farthest_fill(points, owner, num_cluster, turns):
    """`turns` rounds of "cluster turn % num_cluster claims the free point
    farthest from its current mean" on `owner` (-1 = free), in place.  For
    float32 (N, 2) populations -- scan positions -- the rounds run in the
    library (tike_cluster_farthest_fill: one fused pass per round, the same
    float32 arithmetic); anything else takes the NumPy expressions below."""
    if points.dtype == np.float32 and points.ndim == 2 and points.shape[1] == 2:
        lib, check = _native()
        flat = np.ascontiguousarray(points)
        check(lib.tike_cluster_farthest_fill(
            flat.ctypes.data_as(ctypes.c_void_p), len(flat),
            owner.ctypes.data_as(ctypes.c_void_p), num_cluster, turns),
            "cluster farthest fill")
        return
    for turn in range(turns):
        cluster = turn % num_cluster
        free = np.flatnonzero(owner < 0)
        centre = points[owner == cluster].mean(axis=0, keepdims=True)
        reach = np.linalg.norm(points[free] - centre, axis=1)
        owner[free[np.argmax(reach)]] = cluster  # first of the farthest


def _check_cluster_count(num_cluster):
    if not 0 < num_cluster < 0xFFFF:
        raise ValueError(
            f"The number of clusters must be 0 < {num_cluster} < 65536.")


def wobbly_center(population, num_cluster):
    """Equal-size clusters that each span the whole population
    (maximal-heterogeneity clustering, Mishra et al. arXiv:1709.01423; the
    reference's `cluster.wobbly_center`, cluster.py:302-377, whose labels this
    reproduces: the `num_cluster` points nearest the global centroid seed the
    clusters, then the clusters take turns, each claiming the free point
    farthest from its own current mean)."""
    points = np.asarray(population)
    _check_cluster_count(num_cluster)
    count = len(points)
    if num_cluster == 1 or num_cluster >= count:
        return contiguous(points, num_cluster)
    owner = np.full(count, -1, dtype=np.int64)
    spread = np.linalg.norm(points - points.mean(axis=0, keepdims=True), axis=1)
    owner[np.argpartition(spread, num_cluster, axis=0)[:num_cluster]] = (
        np.arange(num_cluster))
    _farthest_fill(points, owner, num_cluster, count - num_cluster)
    return [np.flatnonzero(owner == c) for c in range(num_cluster)]


def wobbly_center_random_bootstrap(population, num_cluster,
                                   boot_fraction=0.95):
    """`wobbly_center` with a random start (the reference's batch method of
    the same name, cluster.py:380-462, whose labels this reproduces): a
    `boot_fraction` of the points -- rounded down to a multiple of
    `num_cluster`, drawn without replacement from the legacy ``np.random``
    generator -- is dealt to the clusters in turn; the clusters then take
    turns claiming the free point farthest from their own current mean."""
    points = np.asarray(population)
    _check_cluster_count(num_cluster)
    count = len(points)
    if num_cluster == 1 or num_cluster >= count:
        return contiguous(points, num_cluster)
    dealt = int(count * boot_fraction)
    dealt -= dealt % num_cluster
    drawn = np.random.choice(count, size=dealt, replace=False)
    owner = np.full(count, -1, dtype=np.int64)
    owner[drawn] = np.arange(dealt) % num_cluster
    _farthest_fill(points, owner, num_cluster, count - dealt)
    return [np.flatnonzero(owner == c) for c in range(num_cluster)]


def stripes_equal_count(population, num_cluster, dim=0):
    """Index sets of `num_cluster` stripes across dimension `dim` holding
    (nearly) the same number of points each (cluster.py:265-299)."""
    points = np.asarray(population)
    if num_cluster == 1 or num_cluster >= len(points):
        return contiguous(points, num_cluster)
    return np.array_split(np.argsort(points[:, dim]), num_cluster)


class _EqualSizeKMeans:
    """Equal-size k-means ("compact" batches): k-means++ seeds, a capacity-
    limited greedy assignment, then pairwise swaps that lower the summed
    distance to the centroids.  Restates the procedure of the reference's
    `cluster.compact` (cluster.py:465-637) -- same draws from the legacy
    NumPy generator, same tie-breaking -- as three vectorised steps."""

    def __init__(self, points, num_cluster):
        self.x = points
        self.k = num_cluster
        self.n = len(points)
        self.rows = np.arange(self.n)
        self.capacity = np.full(num_cluster, self.n // num_cluster)
        self.capacity[:self.n % num_cluster] += 1
        self.label = np.full(self.n, -1, dtype=np.int64)
        self.dist = np.empty((self.n, num_cluster))

    def _distances(self, centroids):
        for c in range(self.k):
            self.dist[:, c] = np.linalg.norm(centroids[c] - self.x, axis=1)

    def seed(self):
        """k-means++: the first seed uniform, every further one with
        probability proportional to the squared distance to the nearest seed
        (inverse-CDF sampling of one legacy-generator uniform per seed: the
        draws `RandomState.choice` makes)."""
        seeds = np.empty(self.k, dtype=np.int64)
        seeds[0] = np.random.randint(0, self.n, size=1)[0]
        nearest = np.inf
        for c in range(1, self.k):
            gap = np.linalg.norm(self.x - self.x[seeds[c - 1]], axis=1)**2
            nearest = np.minimum(nearest, gap)
            cdf = np.cumsum((nearest / nearest.sum()).astype(np.float64))
            cdf /= cdf[-1]
            seeds[c] = cdf.searchsorted(np.random.random_sample(1),
                                        side="right")[0]
        self.centroids = self.x[seeds]
        self._distances(self.centroids)
        self.label[seeds] = np.arange(self.k)
        return seeds

    def fill(self):
        """Hand the free points to their nearest cluster that still has room,
        most decisive point first (smallest nearest-minus-farthest distance);
        whenever a cluster becomes full the preferences are re-evaluated over
        the clusters that remain."""
        room = self.capacity - np.bincount(self.label[self.label >= 0],
                                           minlength=self.k)
        while True:
            open_ = np.flatnonzero(room > 0)
            free = np.flatnonzero(self.label < 0)
            if not len(open_) or not len(free):
                break
            d = self.dist[:, open_]
            near = open_[np.argmin(d, axis=1)]
            far = open_[np.argmax(d, axis=1)]
            urgency = self.dist[self.rows, near] - self.dist[self.rows, far]
            queue = free[np.argsort(urgency[free])]
            target = near[queue]
            # the cluster whose last free place is taken first ends the round
            stop = len(queue) - 1
            for c in np.unique(target):
                where = np.flatnonzero(target == c)
                if len(where) >= room[c]:
                    stop = min(stop, where[room[c] - 1])
            taken = queue[:stop + 1]
            self.label[taken] = target[:stop + 1]
            room -= np.bincount(target[:stop + 1], minlength=self.k)

    def _sweep(self, best, order, regret):
        lib, check = _native()
        moved = ctypes.c_int(0)
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        dist, best, order = (np.ascontiguousarray(a, dtype=t) for a, t in (
            (self.dist, np.float64), (best, np.int64), (order, np.int64)))
        check(lib.tike_cluster_swap_sweep(
            ptr(dist), self.n, self.k, ptr(self.label), ptr(best), ptr(order),
            ptr(regret), ctypes.byref(moved)), "cluster swap sweep")
        return bool(moved.value)

    def refine(self, max_iter):
        """Swap pairs of points between clusters while that lowers the sum of
        the two points' distances to their centroids, unhappiest point first;
        centroids move after every sweep that swapped something."""
        for _ in range(max_iter):
            self._distances(self.centroids)
            best = np.argmin(self.dist, axis=1)
            regret = self.dist[self.rows, best] - self.dist[self.rows,
                                                            self.label]
            # one sweep, unhappiest point first (tike_cluster_swap_sweep): a
            # point with negative regret is exchanged with the point q of
            # another cluster that maximises the gain
            #   d(p, home) + d(q, own(q)) - d(p, own(q)) - d(q, home) > 0
            # (first q among equals); the regrets of both follow
            moved = self._sweep(best, np.argsort(regret), regret)
            if not moved:
                break
            for c in range(self.k):
                self.centroids[c] = self.x[self.label == c].mean(axis=0)
        return self.label


def compact(population, num_cluster, max_iter=500):
    """Equal-size, spatially compact clusters, largest first (the reference's
    `cluster.compact`, cluster.py:465-637; see `_EqualSizeKMeans`).  Seeds with
    the legacy ``np.random`` generator, as the reference does."""
    points = np.asarray(population)
    _check_cluster_count(num_cluster)
    if num_cluster == 1 or num_cluster >= len(points):
        return contiguous(points, num_cluster)
    model = _EqualSizeKMeans(points, num_cluster)
    model.seed()
    model.fill()
    labels = model.refine(max_iter)
    groups = [np.flatnonzero(labels == c) for c in range(num_cluster)]
    return sorted(groups, key=len, reverse=True)


def contiguous(population, num_cluster):
    """Equal contiguous index ranges in the given order."""
    return np.array_split(np.arange(len(population)), num_cluster)


_METHODS = {
    "wobbly_center": wobbly_center,
    "wobbly_center_random_bootstrap": wobbly_center_random_bootstrap,
    "compact": compact,
    "contiguous": contiguous,
}


def batches_contiguous(scan, batch_method, num_batch):
    """(order, batches): `order` permutes positions so that every batch is a
    contiguous index range of the permuted arrays (what
    cluster.by_scan_stripes_contiguous does for one worker, :176-262)."""
    groups = _METHODS[batch_method](np.asarray(scan), num_batch)
    order = np.concatenate(groups)
    breaks = np.cumsum([len(g) for g in groups])[:-1]
    return order, np.array_split(np.arange(len(order)), breaks)


def spatial_order(scan, leaf=8):
    """Permutation that lists positions leaf by leaf of a k-d tree (median
    splits along the longer side, every leaf exactly `leaf` positions except
    the last), so that each run of `leaf` consecutive positions is a compact
    spatial cluster.  The grouped footprint scatter (tike_scatter_patches,
    tike_psi_preconditioner) sums `leaf` consecutive positions on chip before
    it touches the object; the order changes no result beyond summation
    order."""
    scan = np.asarray(scan, dtype=np.float64)
    out = []

    def split(idx):
        n = len(idx)
        if n <= leaf:
            out.append(idx)
            return
        pts = scan[idx]
        axis = int(np.argmax(np.ptp(pts, axis=0)))
        m = ((n + leaf - 1) // leaf // 2) * leaf  # whole leaves on the left
        part = np.argpartition(pts[:, axis], m - 1)
        split(idx[part[:m]])
        split(idx[part[m:]])

    if len(scan):
        split(np.arange(len(scan)))
        return np.concatenate(out)
    return np.zeros(0, dtype=np.int64)

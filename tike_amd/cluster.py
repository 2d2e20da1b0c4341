"""Host-side batch selection (reference src/tike/cluster.py).

Once-per-run host code, out of the accelerated scope; restated so that
``reconstruct`` accepts the reference's ``batch_method`` names.
"""
import numpy as np


def wobbly_center(population, num_cluster):
    """Maximally heterogeneous equal-size clusters (cluster.py:302-377)."""
    population = np.asarray(population)
    if not 0 < num_cluster < 0xFFFF:
        raise ValueError(
            f"The number of clusters must be 0 < {num_cluster} < 65536.")
    if (num_cluster == 1) or (num_cluster >= len(population)):
        return np.array_split(np.arange(population.shape[0]), num_cluster)
    start = np.argpartition(
        np.linalg.norm(population - np.mean(population, axis=0, keepdims=True),
                       axis=1), num_cluster, axis=0)[:num_cluster]
    UNASSIGNED = 0xFFFF
    labels = np.full(len(population), UNASSIGNED, dtype="uint16")
    labels[start] = range(num_cluster)
    for c in range(len(population) - len(start)):
        c = c % num_cluster
        free = labels == UNASSIGNED
        furthest = np.argmax(
            np.linalg.norm(population[free] - np.mean(
                population[labels == c], axis=0, keepdims=True), axis=1))
        i = np.argmax(np.cumsum(free) == (furthest + 1))
        labels[i] = c
    return [np.flatnonzero(labels == c) for c in range(num_cluster)]


def compact(population, num_cluster, max_iter=500):
    """Equal-size k-means-like clusters (cluster.py:465-637), same greedy
    fill and swap refinement; seeds with legacy ``np.random.choice``."""
    population = np.asarray(population)
    if not 0 < num_cluster < 0xFFFF:
        raise ValueError(
            f"The number of clusters must be 0 < {num_cluster} < 65536.")
    if (num_cluster == 1) or (num_cluster >= len(population)):
        return np.array_split(np.arange(population.shape[0]), num_cluster)
    n = len(population)
    _all = np.arange(n)
    size = np.zeros(num_cluster, dtype="int")
    max_size = np.full(num_cluster, n // num_cluster)
    max_size[:n % num_cluster] += 1
    start = np.zeros(num_cluster, dtype="int")
    start[0] = np.random.choice(_all, size=1, p=None)[0]
    distances = np.inf
    for c in range(1, num_cluster):
        distances = np.minimum(
            distances,
            np.linalg.norm(population - population[start[c - 1]], axis=1)**2)
        start[c] = np.random.choice(_all, size=1,
                                    p=distances / distances.sum())[0]
    centroids = population[start].astype(float)
    UNASSIGNED = 0xFFFF
    labels = np.full(n, UNASSIGNED, dtype="uint16")
    distances = np.empty((n, num_cluster))
    unfilled = list(range(num_cluster))
    unassigned = list(range(n))
    for c in unfilled:
        distances[:, c] = np.linalg.norm(centroids[c] - population, axis=1)
        p = start[c]
        labels[p] = c
        unassigned.remove(p)
        size[c] += 1
    for c in range(num_cluster):
        if size[c] >= max_size[c]:
            unfilled.remove(c)
    while unfilled:
        nearest = np.array(unfilled)[np.argmin(distances[:, unfilled], axis=1)]
        farthest = np.array(unfilled)[np.argmax(distances[:, unfilled],
                                                axis=1)]
        priority = np.array(unassigned)[np.argsort(
            (distances[_all, nearest] - distances[_all, farthest])[unassigned])]
        for p in priority:
            labels[p] = nearest[p]
            unassigned.remove(p)
            size[nearest[p]] += 1
            if size[nearest[p]] >= max_size[nearest[p]]:
                unfilled.remove(nearest[p])
                break
    for _ in range(max_iter):
        swapped = False
        for c in range(num_cluster):
            distances[:, c] = np.linalg.norm(centroids[c] - population, axis=1)
        wanted = np.argmin(distances, axis=1)
        happiness = distances[_all, wanted] - distances[_all, labels]
        for p in np.argsort(happiness):
            if happiness[p] < 0:
                net = (distances[p, labels[p]] + distances[_all, labels] -
                       distances[p, labels] - distances[_all, labels[p]])
                good = np.flatnonzero(
                    np.logical_and(net > 0, labels != labels[p]))
                if good.size > 0:
                    swapped = True
                    o = good[np.argmax(net[good])]
                    labels[o], labels[p] = labels[p], labels[o]
                    happiness[o] = distances[o, wanted[o]] - distances[
                        o, labels[o]]
                    happiness[p] = distances[p, wanted[p]] - distances[
                        p, labels[p]]
        if not swapped:
            break
        for c in range(num_cluster):
            centroids[c] = np.mean(population[labels == c], axis=0)
    indices = [np.flatnonzero(labels == c) for c in range(num_cluster)]
    indices.sort(key=len, reverse=True)
    return indices


def contiguous(population, num_cluster):
    """Equal contiguous index ranges in the given order."""
    return np.array_split(np.arange(len(population)), num_cluster)


_METHODS = {
    "wobbly_center": wobbly_center,
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

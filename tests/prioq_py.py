"""searcher.PriorityQueue (internal/searcher/queue.go:25-290) in plain Python — a second, independent reading of the
4-ary heap used ONLY to cross-check the C oracle's heap (oracle/vg_oracle.c vgo_prioq_*) on small cases and to evaluate
the reference's own queue tests (tests/golden/reference_kats.json, group searcher_priority_queue).  Test
infrastructure, like everything under oracle/."""
import numpy as np

ARITY = 4


class PrioQ:
    def __init__(self, is_max):
        self.is_max = is_max
        self.items = []          # (node, np.float32 distance)

    def __len__(self):
        return len(self.items)

    def top(self):               # TopItem queue.go:37-42
        return self.items[0] if self.items else None

    def min_item(self):          # MinItem :46-57
        if not self.items:
            return None
        best = self.items[0]
        for it in self.items[1:]:
            if it[1] < best[1]:
                best = it
        return best

    def push(self, node, dist):  # PushItem :59-62
        self.items.append((node, np.float32(dist)))
        self._up(len(self.items) - 1)

    def push_bounded(self, node, dist, capacity):   # PushItemBounded :67-92
        dist = np.float32(dist)
        if len(self.items) < capacity:
            self.push(node, dist)
            return True
        top = self.items[0][1]
        if (dist < top) if self.is_max else (dist > top):
            self.items[0] = (node, dist)
            self._down(0)
            return True
        return False

    def try_push_bounded(self, node, dist, max_size):   # TryPushBounded :190-215
        dist = np.float32(dist)
        if len(self.items) < max_size:
            self.push(node, dist)
            return True
        top = self.items[0][1]
        if (dist >= top) if self.is_max else (dist <= top):
            return False
        self.items[0] = (node, dist)
        self._down(0)
        return True

    def pop(self):               # PopItem :113-128
        if not self.items:
            return None
        it = self.items[0]
        last = self.items.pop()
        if self.items:
            self.items[0] = last
            self._down(0)
        return it

    def reset(self):
        self.items = []

    def _up(self, i):            # siftUp :161-183
        it = self.items[i]
        while i > 0:
            p = (i - 1) // ARITY
            pd = self.items[p][1]
            if (it[1] <= pd) if self.is_max else (it[1] >= pd):
                break
            self.items[i] = self.items[p]
            i = p
        self.items[i] = it

    def _down(self, i):          # siftDown :221-290
        n = len(self.items)
        it = self.items[i]
        while True:
            fc = ARITY * i + 1
            if fc >= n:
                break
            best, bd = fc, self.items[fc][1]
            for c in range(fc + 1, min(fc + ARITY, n)):
                cd = self.items[c][1]
                if (cd > bd) if self.is_max else (cd < bd):
                    best, bd = c, cd
            if (it[1] >= bd) if self.is_max else (it[1] <= bd):
                break
            self.items[i] = self.items[best]
            i = best
        self.items[i] = it


def brute_search(dists, k, mode, mask=None):
    """hnsw.go:2075-2101 (mode 0) / :2240-2263 (mode 1) over precomputed distances, then the pops."""
    q = PrioQ(True)
    for i, d in enumerate(dists):
        if mask is not None and not mask[i]:
            continue
        if mode == 1:
            q.try_push_bounded(i, d, k)
        elif len(q) < k:
            q.push(i, d)
        elif np.float32(d) < q.top()[1]:
            q.pop()
            q.push(i, d)
    out = []
    while len(q):
        out.append(q.pop())
    out.reverse()
    return np.array([x[0] for x in out], np.uint32), np.array([x[1] for x in out], np.float32)

"""A host model of the scan's finishing stage (csrc/scan_kernels.hip, MODE_FIN — DESIGN 4.1b) replayed in plain Python: the rules
the kernel relies on, each of which was a bug or a near miss while it was built.
  * the k-th largest of the first-panel maxima of ANY set of waves is a valid lower bound of the global k-th best key;
  * a list may hold everything its wave saw before it adopted the published threshold; compactions raise a wave's threshold to its
    list's k-th best key and never lower it;
  * the hand-over keeps keys >= the wave's final threshold (a compaction's threshold IS a key of the list; the published one is
    a key - 1), and whatever a wave drops can never be among the k best;
  * lists never exceed their capacity: at most CAP - 32 keys in front of any push."""
import numpy as np
import pytest

CAP, PANEL = 128, 32


def _key(score: np.float32, row: int) -> int:
    u = int(np.float32(score + np.float32(0.0)).view(np.uint32))
    u = (~u & 0xFFFFFFFF) if u & 0x80000000 else (u | 0x80000000)
    return (u << 32) | (0xFFFFFFFF - row)


class Wave:
    def __init__(self, k):
        self.k, self.tau, self.list, self.max_len = k, 0, [], 0

    def compact(self):
        self.list.sort(reverse=True)
        del self.list[self.k:]
        if len(self.list) >= self.k and self.list[self.k - 1] > self.tau:      # never downwards
            self.tau = self.list[self.k - 1]

    def panel(self, keys, adopted):
        if not adopted and self.tau == 0 and len(self.list) <= CAP - 64:      # no threshold yet: the whole panel, plain stores
            self.list += keys
        else:
            assert len(self.list) <= CAP - 32, "the invariant in front of a push"
            self.list += [x for x in keys if x > self.tau]
            if len(self.list) > CAP - 32:
                self.compact()
        self.max_len = max(self.max_len, len(self.list))
        assert len(self.list) <= CAP


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("k", [1, 2, 20, 64])
def test_finishing_stage_model_returns_the_true_top_k(seed, k):
    rng = np.random.default_rng(seed * 131 + k)
    n_waves = int(rng.integers(8, 40))
    panels_per_wave = int(rng.integers(1, 12))
    n = n_waves * panels_per_wave * PANEL
    scores = rng.standard_normal(n).astype(np.float32)
    if seed % 3 == 0:
        scores[rng.integers(0, n, n // 7)] = scores[0]                 # many exact ties: the row decides
    if seed % 4 == 1:
        scores = np.sort(scores)                                       # ascending: every panel beats the last, compactions galore
    keys = [_key(scores[r], r) for r in range(n)]
    want = sorted(keys, reverse=True)[:k]
    suppliers = rng.permutation(n_waves)[: max(1, n_waves // 3)]       # the waves whose first panels supply the thresholds
    first_max = [max(keys[w * panels_per_wave * PANEL:][:PANEL]) for w in suppliers]
    kth = sorted(first_max, reverse=True)[k - 1] if len(first_max) >= k else 0
    published = kth - 1 if kth else 0
    assert published < want[-1] or published == 0 or len(want) < k    # a valid lower bound of the global k-th best key
    dense = []
    for w in range(n_waves):
        wave = Wave(k)
        adopt_at = int(rng.integers(0, panels_per_wave + 2))           # panel index after which the wave sees the published threshold (maybe never)
        adopted = False
        for p in range(panels_per_wave):
            if p >= adopt_at and not adopted:
                adopted = True
                wave.tau = max(wave.tau, published)
            lo = (w * panels_per_wave + p) * PANEL
            wave.panel(keys[lo:lo + PANEL], adopted)
        dense += [x for x in wave.list if x >= wave.tau]               # the hand-over: >=, not >
        dropped = [x for x in wave.list if x < wave.tau]
        assert all(x < want[-1] for x in dropped) or len(want) < k
    assert sorted(dense, reverse=True)[:k] == want


def test_the_strict_comparison_would_lose_a_compacted_lists_own_kth_best():
    # one wave, k = 1, ascending scores: every compaction makes the best key so far the threshold — '>' at the hand-over drops it
    keys = [_key(np.float32(i), i) for i in range(8 * PANEL)]
    wave = Wave(1)
    for p in range(8):
        wave.panel(keys[p * PANEL:(p + 1) * PANEL], adopted=False)
    wave.compact()
    assert wave.tau == max(keys)
    assert [x for x in wave.list if x > wave.tau] == [] and [x for x in wave.list if x >= wave.tau] == [max(keys)]

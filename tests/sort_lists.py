"""Key lists for the candidate-order tests (std::sort's order of equal sums, scan/expiry_seg.cpp:456, 842): random
lists with many ties, structured lists, and lists built by McIlroy's adversary ("A Killer Adversary for Quicksort",
1999) against a transcription of libstdc++'s introsort loop -- they drive the loop to its depth limit, i.e. into the
heap sort."""
import numpy as np


def _introsort_loop_with(cmp_gt, n):
    v = list(range(n))

    def swap(i, j):
        v[i], v[j] = v[j], v[i]

    def loop(first, last, depth):
        while last - first > 16:
            if depth == 0:
                return
            depth -= 1
            a, b, c = first + 1, first + (last - first) // 2, last - 1
            ea, eb, ec = v[a], v[b], v[c]
            if cmp_gt(ea, eb):
                swap(first, b if cmp_gt(eb, ec) else (c if cmp_gt(ea, ec) else a))
            else:
                swap(first, a if cmp_gt(ea, ec) else (c if cmp_gt(eb, ec) else b))
            lo, hi = first + 1, last
            while True:
                while cmp_gt(v[lo], v[first]):
                    lo += 1
                hi -= 1
                while cmp_gt(v[first], v[hi]):
                    hi -= 1
                if not lo < hi:
                    break
                swap(lo, hi)
                lo += 1
            loop(lo, last, depth)
            last = lo

    loop(0, n, 2 * (n.bit_length() - 1))


def killer(n):
    """keys that make every partition of the introsort loop maximally lopsided"""
    gas = n
    val = [gas] * n
    st = {"solid": 0, "cand": 0}

    def freeze(x):
        val[x] = st["solid"]
        st["solid"] += 1

    def cmp_gt(x, y):
        if val[x] == gas and val[y] == gas:
            freeze(x if x == st["cand"] else y)
        if val[x] == gas:
            st["cand"] = x
        elif val[y] == gas:
            st["cand"] = y
        return val[x] > val[y]

    _introsort_loop_with(cmp_gt, n)
    return np.array(val, np.int64)


def random_lists(rng, count, max_len=430):
    out = []
    for _ in range(count):
        n = int(rng.integers(0, max_len + 1))
        hi = int(rng.choice([2, 3, 5, 10, 50, 1000, 10 ** 6]))
        out.append(rng.integers(0, hi, n).astype(np.int64))
    return out


def structured_lists(sizes=(17, 18, 33, 64, 100, 257, 419, 420)):
    out = []
    for n in sizes:
        out += [np.arange(n), np.arange(n)[::-1].copy(), np.zeros(n, np.int64), np.arange(n) % 3,
                np.r_[np.arange(n // 2), np.arange(n - n // 2)]]
    return [np.asarray(k, np.int64) for k in out]


def adversarial_lists(sizes=(17, 33, 64, 100, 200, 420)):
    out = []
    for n in sizes:
        k = killer(n)
        out += [k, k.max() - k]
    return out

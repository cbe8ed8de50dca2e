"""Where the reference's two code paths disagree.

On an AVX2 / SSE4.1 host the reference's `levenshtein_simd_k_with_opts` runs its SIMD core, which takes a transposition by an
UNCONDITIONAL blend wherever `a[i-1] == b[j-2] && a[i-2] == b[j-1] && a[i-1] != b[j-1]` (src/levenshtein.rs:1056-1075, 2384-2388),
while the scalar routine -- the bit-exactness target of this repository -- takes it only if it is not worse than the other three
moves (`<=`, :517-525).  Each case: (a, b, costs, the scalar path's distance, what a DP with the unconditional blend returns).
The list was found by exhaustive search over two-letter strings of up to five characters (blend model: the scalar recurrence with
the transposition assigned instead of min-ed)."""
CASES = [
    (b"ab", b"aba", (1, 1, 0, 1), 1, 2),
    (b"ab", b"aaba", (1, 1, 0, 1), 2, 3),
    (b"ab", b"abba", (1, 1, 0, 1), 2, 3),
    (b"ab", b"baba", (1, 1, 0, 1), 2, 3),
    (b"ab", b"ababa", (1, 1, 0, 1), 3, 4),
]


def blend_model(a, b, mc, gc, tc):
    """full matrix, linear gaps, the SIMD core's transposition rule"""
    n, m = len(a), len(b)
    dp = [[0] * (m + 1) for _ in range(n + 1)]
    for i in range(1, n + 1):
        dp[i][0] = i * gc
    for j in range(1, m + 1):
        dp[0][j] = j * gc
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            v = min(dp[i - 1][j - 1] + (0 if a[i - 1] == b[j - 1] else mc), dp[i - 1][j] + gc, dp[i][j - 1] + gc)
            if i > 1 and j > 1 and a[i - 1] == b[j - 2] and a[i - 2] == b[j - 1] and a[i - 1] != b[j - 1]:
                v = dp[i - 2][j - 2] + tc
            dp[i][j] = v
    return dp[n][m]

"""Adapter: the product's host API (triple_accel_amd, HIP kernels behind the C ABI) in the shape
kat_runner.py expects.  Used by the -m gpu parity tests."""
import triple_accel_amd as T


class Product:
    supports_trace = True

    @staticmethod
    def hamming(a, b):
        return T.hamming(a, b)

    @staticmethod
    def hamming_search_with_opts(needle, haystack, k, search_type):
        return [tuple(m) for m in T.hamming_search_simd_with_opts(needle, haystack, k, search_type)]

    @staticmethod
    def levenshtein_k_with_opts(a, b, k, trace_on, costs):
        r = T.levenshtein_simd_k_with_opts(a, b, k, trace_on, T.EditCosts(*costs))
        if r is None:
            return (None, None)
        return (r[0], None if r[1] is None else [tuple(e) for e in r[1]])

    @staticmethod
    def levenshtein_full(a, b, trace_on, costs):
        r = T.levenshtein_simd_k_with_opts(a, b, 0xFFFFFFFF, trace_on, T.EditCosts(*costs))
        return (r[0], None if r[1] is None else [tuple(e) for e in r[1]])

    levenshtein = staticmethod(T.levenshtein)
    rdamerau = staticmethod(T.rdamerau)
    levenshtein_exp = staticmethod(T.levenshtein_exp)
    rdamerau_exp = staticmethod(T.rdamerau_exp)

    @staticmethod
    def levenshtein_exp_with_opts(a, b, trace_on, costs):
        r = T.levenshtein_exp_with_opts(a, b, trace_on, T.EditCosts(*costs))
        return (r[0], None if r[1] is None else [tuple(e) for e in r[1]])

    @staticmethod
    def levenshtein_search_with_opts(needle, haystack, k, search_type, costs, anchored):
        return [tuple(m) for m in T.levenshtein_search_simd_with_opts(needle, haystack, k, search_type,
                                                                       T.EditCosts(*costs), anchored)]

    @staticmethod
    def default_search_k(n):
        return (n >> 1) + (n & 1)

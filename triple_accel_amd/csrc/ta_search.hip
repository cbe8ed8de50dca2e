// ta_search.hip -- search entry points (placeholder until the kernels land).
#include <hip/hip_runtime.h>

#include "ta_internal.h"

extern "C" {
int ta_levenshtein_search_simd_with_opts(const uint8_t *, size_t, const uint8_t *, size_t, uint32_t, int,
                                         const ta_edit_costs *, int, ta_match **, size_t *) { return TA_ERR_UNSUPPORTED; }
int ta_levenshtein_search(const uint8_t *, size_t, const uint8_t *, size_t, ta_match **, size_t *) { return TA_ERR_UNSUPPORTED; }
int ta_hamming_search_simd_with_opts(const uint8_t *, size_t, const uint8_t *, size_t, uint32_t, int, ta_match **, size_t *) { return TA_ERR_UNSUPPORTED; }
int ta_hamming_search(const uint8_t *, size_t, const uint8_t *, size_t, ta_match **, size_t *) { return TA_ERR_UNSUPPORTED; }
int ta_levenshtein_search_dev(const uint8_t *, size_t, const uint8_t *, size_t, uint32_t, const ta_edit_costs *, int,
                              uint64_t, uint64_t, ta_match *, size_t, uint64_t *, void *) { return TA_ERR_UNSUPPORTED; }
int ta_hamming_search_dev(const uint8_t *, size_t, const uint8_t *, size_t, uint32_t, uint64_t, ta_match *, size_t,
                          uint64_t *, void *) { return TA_ERR_UNSUPPORTED; }
size_t ta_search_fold_best(ta_match *, size_t, uint32_t, int) { return 0; }
}

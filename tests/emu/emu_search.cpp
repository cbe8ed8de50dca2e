// emu_search.cpp -- runs the search tile function (lev_search_body.h) on the host over a tiled haystack.
// TESTS ONLY: checks the tile/halo decomposition against the monolithic scalar oracle without a GPU.
#include <stdint.h>
#include <stdlib.h>

#include <vector>

#include "lev_search_body.h"

using namespace ta;

struct Hit { uint64_t start, end; uint32_t k, pad; };

template <int N>
static void run_tiles(const uint8_t *needle, uint32_t n, const uint8_t *hay, uint64_t h, const SearchCosts &C,
                      bool trans, uint64_t tile, uint64_t halo, std::vector<Hit> &hits) {
    for (uint64_t eb = 0; eb < h; eb += tile) {
        uint64_t ee = eb + tile < h ? eb + tile : h;
        uint64_t cb = eb > halo ? eb - halo : 0;
        auto emit = [&](uint64_t end, uint32_t len, uint32_t cost) { hits.push_back(Hit{end - len, end, cost, 0}); };
        if (trans) lev_search_tile<N, true>(hay, needle, n, C, cb, eb, ee, emit);
        else lev_search_tile<N, false>(hay, needle, n, C, cb, eb, ee, emit);
    }
}

static void run_tiles_mem(const uint8_t *needle, uint32_t n, const uint8_t *hay, uint64_t h, const SearchCosts &C,
                          bool trans, uint64_t tile, uint64_t halo, std::vector<Hit> &hits) {
    // emulate the GPU layout: all tiles' columns interleaved with stride = number of tiles
    const uint64_t tiles = (h + tile - 1) / tile;
    std::vector<uint32_t> col((size_t)(6 * (uint64_t)(n + 1) * (tiles ? tiles : 1)), 0xABABABABu);
    uint64_t t = 0;
    for (uint64_t eb = 0; eb < h; eb += tile, t++) {
        uint64_t ee = eb + tile < h ? eb + tile : h;
        uint64_t cb = eb > halo ? eb - halo : 0;
        lev_search_tile_mem(hay, needle, n, C, trans, col.data() + t, tiles, cb, eb, ee,
                            [&](uint64_t end, uint32_t len, uint32_t cost) { hits.push_back(Hit{end - len, end, cost, 0}); });
    }
}

extern "C" int emu_lev_search(const uint8_t *needle, uint32_t n, const uint8_t *hay, uint64_t h, uint32_t k,
                              uint32_t mc, uint32_t gc, uint32_t sg, int has_t, uint32_t tc, int anchored,
                              uint64_t tile, uint64_t halo, Hit *out, uint64_t cap, uint64_t *count) {
    SearchCosts C{k, mc, gc, sg, tc, (uint32_t)(anchored ? 1 : 0)};
    std::vector<Hit> hits;
    if (n == 0) return 1;
    if (n > 32) run_tiles_mem(needle, n, hay, h, C, has_t != 0, tile, halo, hits);
    else if (n <= 8) run_tiles<8>(needle, n, hay, h, C, has_t != 0, tile, halo, hits);
    else if (n <= 16) run_tiles<16>(needle, n, hay, h, C, has_t != 0, tile, halo, hits);
    else run_tiles<32>(needle, n, hay, h, C, has_t != 0, tile, halo, hits);
    *count = hits.size();
    for (uint64_t i = 0; i < hits.size() && i < cap; i++) out[i] = hits[i];
    return 0;
}

// emu_search.cpp -- runs the search tile function (lev_search_body.h) on the host over a tiled haystack.
// TESTS ONLY: checks the tile/halo decomposition against the monolithic scalar oracle without a GPU.
#include <stdint.h>
#include <stdlib.h>

#include <vector>

#include "emu_wave.h"
#include "lev_search_body.h"
#include "lev_search_wave_body.h"

using namespace ta;

struct Hit { uint64_t start, end; uint32_t k, pad; };

static int g_packed = 0;          // 0: cost + length in two registers, 1: packed key, 2: packed key on a whole wavefront per tile
extern "C" void emu_search_set_packed(int on) { g_packed = on; }

// the wavefront-per-block form (lev_search_wave_body.h): tiles are the blocks, n <= 64, tile + halo <= 256 columns
static int run_tiles_wave(const uint8_t *needle, uint32_t n, const uint8_t *hay, uint64_t h, const SearchCosts &C,
                          bool trans, uint64_t tile, uint64_t halo, std::vector<Hit> &hits) {
    if (n > 64 || tile + halo > SRCH_WAVE_MAX_COLS || C.anchored) return 3;
    uint8_t nd[64 + 16] = {0};
    for (uint32_t i = 0; i < n; i++) nd[i] = needle[i];
    for (uint64_t eb = 0; eb < h; eb += tile) {
        uint64_t ee = eb + tile < h ? eb + tile : h;
        uint64_t cb = eb > halo ? eb - halo : 0;
        auto flush = [&](const VB &hit, const V32 &keys, const V32 &cols) {      // lane t: emitted column t of the block
            for (uint32_t q = 0; q < 64; q++) {
                if (!hit.v[q]) continue;
                const uint64_t end = cb + cols.v[q] + 1;
                hits.push_back(Hit{end - (0xFFFFu - (keys.v[q] & 0xFFFFu)), end, keys.v[q] >> 16, 0});
            }
        };
        uint8_t row_lds[64 * 4];
        memset(row_lds, 0xA5, sizeof(row_lds));
        if (trans) lev_search_block_wave<EmuWave, true>(hay, nd, n, C, cb, eb, ee, row_lds, flush);
        else lev_search_block_wave<EmuWave, false>(hay, nd, n, C, cb, eb, ee, row_lds, flush);
    }
    return 0;
}

template <int N>
static void run_tiles(const uint8_t *needle, uint32_t n, const uint8_t *hay, uint64_t h, const SearchCosts &C,
                      bool trans, uint64_t tile, uint64_t halo, std::vector<Hit> &hits) {
    for (uint64_t eb = 0; eb < h; eb += tile) {
        uint64_t ee = eb + tile < h ? eb + tile : h;
        uint64_t cb = eb > halo ? eb - halo : 0;
        auto emit = [&](uint64_t end, uint32_t len, uint32_t cost) { hits.push_back(Hit{end - len, end, cost, 0}); };
        if (g_packed == 1) {
            if (trans) lev_search_tile_packed<N, true>(hay, needle, n, C, cb, eb, ee, emit);
            else lev_search_tile_packed<N, false>(hay, needle, n, C, cb, eb, ee, emit);
        } else {
            if (trans) lev_search_tile<N, true>(hay, needle, n, C, cb, eb, ee, emit);
            else lev_search_tile<N, false>(hay, needle, n, C, cb, eb, ee, emit);
        }
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

extern "C" int emu_search_anchored_packed_ok(uint64_t h, uint32_t n, uint32_t mc, uint32_t gc, uint32_t sg) {
    return srch_anchored_packed_ok(h, n, mc, gc, sg) ? 1 : 0;
}

extern "C" uint32_t emu_search_filter_k(uint32_t k, uint32_t mc, uint32_t gc, uint32_t sg, int has_t, uint32_t tc) {
    return srch_filter_k(k, mc, gc, sg, has_t != 0, tc);
}

extern "C" int emu_lev_search(const uint8_t *needle, uint32_t n, const uint8_t *hay, uint64_t h, uint32_t k,
                              uint32_t mc, uint32_t gc, uint32_t sg, int has_t, uint32_t tc, int anchored,
                              uint64_t tile, uint64_t halo, Hit *out, uint64_t cap, uint64_t *count) {
    SearchCosts C{k, mc, gc, sg, tc, (uint32_t)(anchored ? 1 : 0)};
    std::vector<Hit> hits;
    if (n == 0) return 1;
    if (g_packed == 2) {
        int rc = run_tiles_wave(needle, n, hay, h, C, has_t != 0, tile, halo, hits);
        if (rc) return rc;
    } else if (n > 32) run_tiles_mem(needle, n, hay, h, C, has_t != 0, tile, halo, hits);
    else if (g_packed == 1) {
        switch (n) {
#define TA_N(x) case x: run_tiles<x>(needle, n, hay, h, C, has_t != 0, tile, halo, hits); break;
            TA_N(1) TA_N(2) TA_N(3) TA_N(4) TA_N(5) TA_N(6) TA_N(7) TA_N(8) TA_N(9) TA_N(10) TA_N(11) TA_N(12)
            TA_N(13) TA_N(14) TA_N(15) TA_N(16) TA_N(17) TA_N(18) TA_N(19) TA_N(20) TA_N(21) TA_N(22) TA_N(23) TA_N(24)
            TA_N(25) TA_N(26) TA_N(27) TA_N(28) TA_N(29) TA_N(30) TA_N(31) TA_N(32)
#undef TA_N
        }
    } else if (n <= 8) run_tiles<8>(needle, n, hay, h, C, has_t != 0, tile, halo, hits);
    else if (n <= 16) run_tiles<16>(needle, n, hay, h, C, has_t != 0, tile, halo, hits);
    else run_tiles<32>(needle, n, hay, h, C, has_t != 0, tile, halo, hits);
    *count = hits.size();
    for (uint64_t i = 0; i < hits.size() && i < cap; i++) out[i] = hits[i];
    return 0;
}

// ---- bit-parallel candidate filter (lev_filter_body.h): flagged 64-column blocks of a tiled scan
#include "lev_filter_body.h"

template <int NWF>
static void filter_n(const uint8_t *needle, uint32_t n, const uint8_t *hay, uint64_t h, uint32_t k, int has_t, uint64_t tile,
                     uint64_t halo, std::vector<uint64_t> &blocks) {
    std::vector<uint32_t> peq(256 * NWF);
    for (uint32_t c = 0; c < 256; c++)
        for (uint32_t w = 0; w < (uint32_t)NWF; w++) peq[c * NWF + w] = lev_filter_peq_word(needle, n, NWF, c, w);
    for (uint64_t eb = 0; eb < h; eb += tile) {
        const uint64_t ee = eb + tile < h ? eb + tile : h, cb = eb > halo ? eb - halo : 0;
        auto pq = [&](uint32_t c, uint32_t (&Eq)[NWF]) { for (int w = 0; w < NWF; w++) Eq[w] = peq[c * NWF + w]; };
        auto mk = [&](uint64_t b) { blocks.push_back(b); };
        if (has_t) lev_filter_tile_n<NWF, true>(hay, pq, n, k, cb, eb, ee, mk);
        else lev_filter_tile_n<NWF, false>(hay, pq, n, k, cb, eb, ee, mk);
    }
}

extern "C" int emu_lev_filter(const uint8_t *needle, uint32_t n, const uint8_t *hay, uint64_t h, uint32_t k, int has_t,
                              uint64_t tile, uint64_t halo, int force_words, uint64_t *blocks_out, uint64_t cap, uint64_t *count) {
    if (n == 0 || n > 512 || tile == 0 || tile % FILTER_BLOCK) return 1;
    std::vector<uint64_t> blocks;
    uint32_t nwf = (n + 31) / 32;
    const bool lower_bound = force_words == -1;       // lev_filter_tile_lb: the score settled per 32 columns (needles <= 32 bytes)
    if (force_words > (int)nwf) nwf = (uint32_t)force_words;
    if (nwf == 1) {
        uint32_t peq[256];
        for (uint32_t c = 0; c < 256; c++) peq[c] = lev_filter_peq(needle, n, c);
        for (uint64_t eb = 0; eb < h; eb += tile) {
            const uint64_t ee = eb + tile < h ? eb + tile : h, cb = eb > halo ? eb - halo : 0;
            auto pq = [&](uint32_t c) { return peq[c]; };
            auto mk = [&](uint64_t b) { blocks.push_back(b); };
            if (lower_bound) {
                if (has_t) lev_filter_tile_lb<true>(hay, pq, n, k, cb, eb, ee, mk);
                else lev_filter_tile_lb<false>(hay, pq, n, k, cb, eb, ee, mk);
            } else if (has_t) lev_filter_tile<true>(hay, pq, n, k, cb, eb, ee, mk);
            else lev_filter_tile<false>(hay, pq, n, k, cb, eb, ee, mk);
        }
    } else {
        switch (nwf) {
            case 2: filter_n<2>(needle, n, hay, h, k, has_t, tile, halo, blocks); break;
            case 3: filter_n<3>(needle, n, hay, h, k, has_t, tile, halo, blocks); break;
            case 4: filter_n<4>(needle, n, hay, h, k, has_t, tile, halo, blocks); break;
            case 5: filter_n<5>(needle, n, hay, h, k, has_t, tile, halo, blocks); break;
            case 6: filter_n<6>(needle, n, hay, h, k, has_t, tile, halo, blocks); break;
            case 7: filter_n<7>(needle, n, hay, h, k, has_t, tile, halo, blocks); break;
            case 8: filter_n<8>(needle, n, hay, h, k, has_t, tile, halo, blocks); break;
            case 9: case 10: case 11: case 12: filter_n<12>(needle, n, hay, h, k, has_t, tile, halo, blocks); break;
            case 13: case 14: case 15: case 16: filter_n<16>(needle, n, hay, h, k, has_t, tile, halo, blocks); break;
            default: return 2;
        }
    }
    *count = blocks.size();
    for (uint64_t i = 0; i < blocks.size() && i < cap; i++) blocks_out[i] = blocks[i];
    return 0;
}

// ---- shift-add hamming_search scan (ham_search_body.h)
#include "ham_search_body.h"

template <int NWS>
static void ham_n(const uint8_t *needle, uint32_t n, const uint8_t *hay, uint64_t h, uint32_t k, uint64_t tile, std::vector<Hit> &hits) {
    std::vector<uint32_t> tab(256 * NWS);
    for (uint32_t c = 0; c < 256; c++)
        for (uint32_t w = 0; w < (uint32_t)NWS; w++) tab[c * NWS + w] = ham_sa_table_word(needle, n, c, w);
    const uint64_t offsets = h - n + 1;
    for (uint64_t ob = 0; ob < offsets; ob += tile) {
        const uint64_t oe = ob + tile < offsets ? ob + tile : offsets;
        ham_sa_tile<NWS>(hay, [&](uint32_t c, uint32_t (&Tc)[NWS]) { for (int w = 0; w < NWS; w++) Tc[w] = tab[c * NWS + w]; }, n, k, ob, oe,
                         [&](uint64_t p, uint32_t cnt) { hits.push_back(Hit{p, p + n, cnt, 0u}); });
    }
}

// ---- bit-sliced counters (ham_bits_body.h): the kernel's tile walk -- n - 1 bytes in front of the tile, verdict per byte, recount of hits
#include "ham_bits_body.h"
template <int B>
static void ham_bits_all(const uint8_t *needle, uint32_t n, const uint8_t *hay, uint64_t h, uint32_t k, uint64_t tile, std::vector<Hit> &hits) {
    uint32_t mis[256], bias[B];
    for (uint32_t c = 0; c < 256; c++) mis[c] = ham_bits_mis(needle, n, c);
    ham_bits_bias<B>(k, n, bias);
    const uint64_t last = h - n;
    for (uint64_t b0 = 0; b0 < h; b0 += tile) {
        const uint64_t b1 = b0 + tile < h ? b0 + tile : h;
        HamBitsState<B> st;
        ham_bits_reset<B>(st, n);
        for (uint64_t i = b0 > n - 1u ? b0 - (n - 1u) : 0; i < b0; i++) ham_bits_step<B>(st, mis[hay[i]], bias);
        for (uint64_t i = b0; i < b1; i++) {
            if (ham_bits_step<B>(st, mis[hay[i]], bias) >> 31) continue;
            if (i < n - 1u) continue;
            const uint64_t pos = i - (n - 1u);
            if (pos > last) continue;
            uint32_t cnt = 0;
            for (uint32_t j = 0; j < n; j++) cnt += hay[pos + j] != needle[j];
            hits.push_back(Hit{pos, pos + n, cnt, 0u});
        }
    }
}
extern "C" int emu_ham_search_bits(const uint8_t *needle, uint32_t n, const uint8_t *hay, uint64_t h, uint32_t k, uint64_t tile,
                                   Hit *out, uint64_t cap, uint64_t *count) {
    const int B = ham_bits_planes(k);
    if (n == 0 || n > 32 || n > h || tile == 0 || !B || k >= n) return 1;
    std::vector<Hit> hits;
    switch (B) {
        case 1: ham_bits_all<1>(needle, n, hay, h, k, tile, hits); break; case 2: ham_bits_all<2>(needle, n, hay, h, k, tile, hits); break;
        case 3: ham_bits_all<3>(needle, n, hay, h, k, tile, hits); break; case 4: ham_bits_all<4>(needle, n, hay, h, k, tile, hits); break;
        default: ham_bits_all<5>(needle, n, hay, h, k, tile, hits); break;
    }
    *count = hits.size();
    for (uint64_t i = 0; i < hits.size() && i < cap; i++) out[i] = hits[i];
    return 0;
}

// ---- bit-sliced counters over a subset of the needle's positions, Q phases per dword (ham_phase_body.h): the kernel's tile walk --
// span bytes in front of the tile in steps of Q, a verdict word per step, candidates recounted over the whole needle
#include "ham_phase_body.h"
template <int B>
static void ham_phase_all(const uint8_t *needle, uint32_t n, const uint8_t *hay, uint64_t h, uint32_t k, uint64_t tile, const HamPhaseGeom &G,
                          std::vector<Hit> &hits, uint64_t *candidates) {
    uint32_t mis[256], bias[B];
    for (uint32_t c = 0; c < 256; c++) mis[c] = ham_phase_mis(needle, G, c);
    ham_phase_bias<B>(k, G, bias);
    const uint64_t last = h - n;
    const uint32_t Q = G.Q, W = G.W;
    for (uint64_t b0 = 0; b0 < h; b0 += tile) {
        const uint64_t b1 = b0 + tile < h ? b0 + tile : h;
        HamBitsState<B> st;
        ham_phase_reset<B>(st, G);
        auto step = [&](uint64_t i) -> uint32_t {
            uint32_t m = 0;
            for (int r = (int)Q - 1; r >= 0; r--) {
                const uint32_t t = mis[i + (uint32_t)r < h ? hay[i + (uint32_t)r] : 0u];
                m = Q == 1 ? t : ((m << (W & 31u)) | t);
            }
            return Q > 1 ? ham_phase_step<B, true>(st, m, bias, G.keep) : ham_phase_step<B, false>(st, m, bias, G.keep);
        };
        for (uint64_t i = b0 > G.span ? b0 - G.span : 0; i < b0; i += Q) step(i);
        for (uint64_t i = b0; i < b1; i += Q) {
            const uint32_t ov = step(i);
            for (uint32_t r = 0; r < Q; r++) {
                if ((ov >> (r * W + W - 1u)) & 1u) continue;
                const uint64_t x = i + r;
                if (x < G.span) continue;
                const uint64_t pos = x - G.span;
                if (pos > last) continue;
                (*candidates)++;
                uint32_t cnt = 0;
                for (uint32_t j = 0; j < n; j++) cnt += hay[pos + j] != needle[j];
                if (cnt <= k) hits.push_back(Hit{pos, pos + n, cnt, 0u});
            }
        }
    }
}
// q_force: 0 = the plan's Q, else 1 / 2 / 4 where the plan allows at least that many phases; rc 1 = the form does not apply
extern "C" int emu_ham_search_phase(const uint8_t *needle, uint32_t n, const uint8_t *hay, uint64_t h, uint32_t k, uint64_t tile, uint32_t q_force,
                                    Hit *out, uint64_t cap, uint64_t *count, uint32_t *plan /* Q, L, B */, uint64_t *candidates) {
    uint32_t Q = 0, L = 0; int B = 0;
    if (n == 0 || n > h || tile == 0 || (tile % 4u) || !ham_phase_plan(n, k, Q, L, B)) return 1;
    if (q_force && q_force < Q) {
        Q = q_force; L = (n + Q - 1u) / Q;
        if (L > 32u / Q) L = 32u / Q;
    }
    const HamPhaseGeom G = ham_phase_geom(Q, L);
    plan[0] = Q; plan[1] = L; plan[2] = (uint32_t)B;
    *candidates = 0;
    std::vector<Hit> hits;
    switch (B) {
        case 1: ham_phase_all<1>(needle, n, hay, h, k, tile, G, hits, candidates); break; case 2: ham_phase_all<2>(needle, n, hay, h, k, tile, G, hits, candidates); break;
        case 3: ham_phase_all<3>(needle, n, hay, h, k, tile, G, hits, candidates); break; case 4: ham_phase_all<4>(needle, n, hay, h, k, tile, G, hits, candidates); break;
        default: ham_phase_all<5>(needle, n, hay, h, k, tile, G, hits, candidates); break;
    }
    *count = hits.size();
    for (uint64_t i = 0; i < hits.size() && i < cap; i++) out[i] = hits[i];
    return 0;
}

// ---- SWAR form (ham_swar_body.h): the per-lane function over every 16-byte-aligned lane position, as the kernel walks them
#include "ham_swar_body.h"
template <int NW>
static void ham_swar_all(const uint8_t *needle, uint32_t n, const uint8_t *hay, uint64_t h, uint32_t k, uint32_t delta, std::vector<Hit> &hits) {
    uint32_t nd12[NW], tail_mask, tail_pad;
    ham_swar_needle<NW>(needle, n, nd12, tail_mask, tail_pad);
    const uint64_t last = h - n, end = (uint64_t)delta + h;
    for (uint64_t byte0 = 0; byte0 < end; byte0 += 16) {
        uint32_t w[4 + NW];
        for (int i = 0; i < 4 + NW; i++) {                  // the kernel's loads: 16 bytes at a time, zeros when the load starts past the end
            uint32_t v = 0;
            const uint64_t q0 = byte0 + 16u * (uint64_t)(i >> 2);
            if (q0 <= end)
                for (int b = 0; b < 4; b++) {
                    const uint64_t x = byte0 + 4u * (uint64_t)i + (uint64_t)b;
                    // bytes in front of the haystack (x < delta) and its 16 bytes of slack: whatever memory holds -- garbage here
                    v |= (uint32_t)((x >= delta && x < end) ? hay[x - delta] : (uint8_t)(0x5A + x)) << (8 * b);
                }
            w[i] = v;
        }
        uint32_t cnt[16];
        ham_swar_lane<NW>(w, nd12, tail_mask, tail_pad, cnt);
        for (uint32_t o = 0; o < 16; o++) {
            const uint64_t x = byte0 + o;
            if ((cnt[o] >> 3) > k || x < delta || x - delta > last) continue;
            hits.push_back(Hit{x - delta, x - delta + n, cnt[o] >> 3, 0u});
        }
    }
}
extern "C" int emu_ham_search_swar(const uint8_t *needle, uint32_t n, const uint8_t *hay, uint64_t h, uint32_t k, uint32_t delta,
                                   Hit *out, uint64_t cap, uint64_t *count) {
    if (n == 0 || n > 64 || n > h || delta > 15) return 1;
    std::vector<Hit> hits;
    switch ((n + 3) / 4) {
        case 9: ham_swar_all<9>(needle, n, hay, h, k, delta, hits); break; case 10: ham_swar_all<10>(needle, n, hay, h, k, delta, hits); break;
        case 11: ham_swar_all<11>(needle, n, hay, h, k, delta, hits); break; case 12: ham_swar_all<12>(needle, n, hay, h, k, delta, hits); break;
        case 13: ham_swar_all<13>(needle, n, hay, h, k, delta, hits); break; case 14: ham_swar_all<14>(needle, n, hay, h, k, delta, hits); break;
        case 15: ham_swar_all<15>(needle, n, hay, h, k, delta, hits); break; case 16: ham_swar_all<16>(needle, n, hay, h, k, delta, hits); break;
        case 1: ham_swar_all<1>(needle, n, hay, h, k, delta, hits); break; case 2: ham_swar_all<2>(needle, n, hay, h, k, delta, hits); break;
        case 3: ham_swar_all<3>(needle, n, hay, h, k, delta, hits); break; case 4: ham_swar_all<4>(needle, n, hay, h, k, delta, hits); break;
        case 5: ham_swar_all<5>(needle, n, hay, h, k, delta, hits); break; case 6: ham_swar_all<6>(needle, n, hay, h, k, delta, hits); break;
        case 7: ham_swar_all<7>(needle, n, hay, h, k, delta, hits); break; case 8: ham_swar_all<8>(needle, n, hay, h, k, delta, hits); break;
        default: return 1;
    }
    *count = hits.size();
    for (uint64_t i = 0; i < hits.size() && i < cap; i++) out[i] = hits[i];
    return 0;
}

extern "C" int emu_ham_search(const uint8_t *needle, uint32_t n, const uint8_t *hay, uint64_t h, uint32_t k, uint64_t tile, int words,
                              Hit *out, uint64_t cap, uint64_t *count) {
    if (n == 0 || n > 32 || n > h || tile == 0) return 1;
    std::vector<Hit> hits;
    int nws = (int)((n + 3) / 4);
    if (words > nws) nws = words;
    if (nws <= 2) ham_n<2>(needle, n, hay, h, k, tile, hits);
    else if (nws <= 4) ham_n<4>(needle, n, hay, h, k, tile, hits);
    else ham_n<8>(needle, n, hay, h, k, tile, hits);
    *count = hits.size();
    for (uint64_t i = 0; i < hits.size() && i < cap; i++) out[i] = hits[i];
    return 0;
}

// CPU-only sanitizer target for the HOST half of libmfar_hip.so: the stage-1 chunk-table builder and the payload / workspace
// layouts (multifield-adaptive-retrieval_amd/csrc/mfar_tables.h -- the very code mfar_hip.hip compiles).  Built by
// tests/test_host_tables.py with g++ -fsanitize=address,undefined -D_GLIBCXX_ASSERTIONS (vector indexing is bounds-checked)
// and run over the bench shapes plus a seeded fuzz of (rows per field, fields, depth k, compute units, workgroups per CU, waves,
// sample sizing).  Every table must satisfy the invariants the kernels rely on; a violation aborts with the offending shape.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "mfar_tables.h"

static long long n_checked = 0;
#define REQUIRE(cond)                                                                                         \
    do {                                                                                                      \
        if (!(cond)) {                                                                                        \
            std::fprintf(stderr, "FAILED %s (line %d): F=%d k=%d n_cu=%d wgs=%d waves=%d solo=%d forced=%d stm=%d\n", #cond, __LINE__, F, k, \
                         n_cu, wgs, waves, (int)solo, (int)forced, stm);                                      \
            for (int f_ = 0; f_ < F; ++f_) std::fprintf(stderr, "  field %d: tiles %d rows %lld\n", f_, g.n_tiles[f_], g.n_rows[f_]); \
            std::abort();                                                                                     \
        }                                                                                                     \
    } while (0)

static void check_table(const S1GeomHost& g, int F, int n_cu, int k, bool solo, int stm, bool forced, int waves, int wgs, int sdiv, int atgt) {
    S1TableHost t;
    s1_build_table(g, F, n_cu, k, solo, stm, forced, waves, wgs, sdiv, atgt, t, 0);
    ++n_checked;
    const int cap = std::max(1, std::min(128, (64 * 256) / k));
    const int l2cap = std::max(1, std::min(cap, 8192 / k));
    REQUIRE(t.k == k && t.wgs == wgs);
    REQUIRE((int)t.fchunk.size() == F + 1 && (int)t.samp_n.size() == F && t.fchunk[0] == 0);
    int n_scanned = 0;
    for (int f = 0; f < F; ++f) n_scanned += g.n_tiles[f] > 0;
    REQUIRE(t.n_chunks == (int)t.chunks.size() && t.fchunk[F] == t.n_chunks && t.n_chunks >= n_scanned);
    long long tiles = 0;
    int max_cf = 0, stride = 0;
    bool need_two = false;
    for (int f = 0; f < F; ++f) {
        const int c0 = t.fchunk[f], c1 = t.fchunk[f + 1];
        if (g.n_tiles[f] == 0) {                                       // a field the table does not scan (switched off): nothing at all
            REQUIRE(c1 == c0 && t.samp_n[f] == 0);
            continue;
        }
        REQUIRE(c1 > c0);                                              // every scanned field is scanned by at least one workgroup
        REQUIRE(c1 - c0 <= std::max(1, g.n_tiles[f]));                 // no empty chunks
        REQUIRE((long long)(c1 - c0) <= (long long)cap * l2cap);        // what a (two-level) merge can hold
        max_cf = std::max(max_cf, c1 - c0);
        need_two = need_two || (c1 - c0 > cap);
        int tl = 0;
        for (int c = c0; c < c1; ++c) {
            const S1Chunk& ck = t.chunks[c];
            REQUIRE(ck.f == f && ck.n_rows == (int)g.n_rows[f] && ck.base == g.base[f]);
            REQUIRE(ck.t0 == (c == c0 ? 0 : t.chunks[c - 1].t1));      // contiguous cover of the field's tiles, in order
            REQUIRE(ck.t1 > ck.t0);
            REQUIRE(ck.ns >= 1 && ck.ns <= ck.t1 - ck.t0);             // the sample pass never walks past its chunk
            REQUIRE(ck.tl0 == tl);                                     // sample output slots: dense, in chunk order
            tl += ck.ns;
        }
        REQUIRE(t.chunks[c1 - 1].t1 == g.n_tiles[f]);
        REQUIRE(t.samp_n[f] == waves * tl);
        stride = std::max(stride, waves * tl);
        tiles += g.n_tiles[f];
    }
    REQUIRE(t.max_chunks == max_cf && t.samp_stride == stride && t.total_tiles == tiles);
    REQUIRE(t.thresholded_tiles >= 0 && t.thresholded_tiles <= tiles && t.sample_tiles >= 1);
    REQUIRE(t.two_level == need_two);
    if (t.two_level) {
        REQUIRE((int)t.fgroup.size() == F + 1 && t.fgroup[0] == 0 && t.fgroup[F] == t.n_groups);
        REQUIRE((int)t.gfield.size() == t.n_groups && (int)t.gchunk.size() == t.n_groups + 1 && t.gchunk[t.n_groups] == t.n_chunks);
        int mg = 0, mgc = 0;
        for (int f = 0; f < F; ++f) {
            if (g.n_tiles[f] == 0) {
                REQUIRE(t.fgroup[f + 1] == t.fgroup[f]);
                continue;
            }
            REQUIRE(t.fgroup[f + 1] > t.fgroup[f] && t.gchunk[t.fgroup[f]] == t.fchunk[f]);
            mg = std::max(mg, t.fgroup[f + 1] - t.fgroup[f]);
            for (int gi = t.fgroup[f]; gi < t.fgroup[f + 1]; ++gi) {
                const int hi = gi + 1 < t.fgroup[f + 1] ? t.gchunk[gi + 1] : t.fchunk[f + 1];
                REQUIRE(t.gfield[gi] == f && hi > t.gchunk[gi] && hi - t.gchunk[gi] <= cap);      // level 1 holds a group's lists
                mgc = std::max(mgc, hi - t.gchunk[gi]);
            }
        }
        REQUIRE(mg == t.max_groups && mg <= l2cap && mgc == t.max_group_chunks);                    // level 2 holds a field's groups
    }
    if (!solo) {
        const long long want = (long long)wgs * n_cu;
        long long floors = 0;
        for (int f = 0; f < F; ++f) floors += g.n_tiles[f] == 0 ? 0 : std::min<long long>(g.n_tiles[f], (3LL * k + 7) / 8 + 4);
        REQUIRE(t.n_chunks <= want + floors + F);                      // the grid stays about one wave of workgroups
    }
}

static void check_layouts(int Q, int F, int k1, int k2) {
    const PayloadLayout P = payload_layout(Q, F, k1);
    const long long C = (long long)F * k1;
    const long long po[] = {P.hdr, P.ids, P.scores, P.cand, P.ncand, P.x, P.total};
    const long long ps[] = {64, (long long)Q * F * k1 * 8, (long long)Q * F * k1 * 4, Q * C * 8, (long long)Q * 4, Q * C * F * 4};
    for (int i = 0; i < 6; ++i)
        if (po[i] % 256 || po[i + 1] < po[i] + ps[i]) { std::fprintf(stderr, "payload layout Q=%d F=%d k1=%d section %d\n", Q, F, k1, i); std::abort(); }
    const MergeWsLayout M = merge_ws_layout(Q, F, k1);
    const long long mo[] = {M.lids, M.lsc, M.cand, M.ncand, M.x, M.total};
    const long long ms[] = {(long long)Q * F * k1 * 8, (long long)Q * F * k1 * 4, Q * C * 8, (long long)Q * 4, Q * C * F * 4};
    for (int i = 0; i < 5; ++i)
        if (mo[i] % 256 || mo[i + 1] < mo[i] + ms[i]) { std::fprintf(stderr, "merge ws layout Q=%d F=%d k1=%d section %d\n", Q, F, k1, i); std::abort(); }
    const ListsLayout L = lists_layout(Q, F, k1);
    if (L.ids != 0 || L.scores % 256 || L.scores < (long long)Q * F * k1 * 8 || L.total < L.scores + (long long)Q * F * k1 * 4 || L.total % 256) std::abort();
    const TopkLayout T = topk_layout(Q, k2);
    if (T.scores < (long long)Q * k2 * 8 || T.ncand < T.scores + (long long)Q * k2 * 4 || T.flag < T.ncand + (long long)Q * 4 || T.total < T.flag + 4 ||
        T.scores % 256 || T.ncand % 256 || T.total % 256 || T.flag % 4)
        std::abort();
    ++n_checked;
}

static S1GeomHost geom(const std::vector<long long>& rows, int E) {
    S1GeomHost g;
    long long total = 0;
    for (long long n : rows) {
        long long blk = (n + 63) / 64;
        blk = std::max(4LL, ((blk + 3) / 4) * 4);
        g.n_rows.push_back(n);
        g.base.push_back(total);
        g.n_tiles.push_back((int)(blk / 4));
        total += blk * 64 * E;
    }
    return g;
}

int main(int argc, char** argv) {
    const int n_fuzz = argc > 1 ? std::atoi(argv[1]) : 20000;
    // the bench shapes: document slabs (every field D rows) and screen slabs (unique rows per field, structured corpus)
    const std::vector<std::vector<long long>> shapes = {
        std::vector<long long>(8, 1000000), std::vector<long long>(22, 129375), std::vector<long long>(5, 700244), std::vector<long long>(8, 957192),
        std::vector<long long>(16, 1250000), std::vector<long long>(8, 125000), {1000000}, {1}, {0}, std::vector<long long>(32, 300),
        {920000, 230000, 920000, 10, 920000, 1000000, 920000, 920000}, {129375, 38000, 3000, 10, 129375, 2, 1, 64, 65, 255, 256, 257}};
    for (const auto& rows : shapes)
        for (int k : {1, 10, 100, 128, 164, 192})
            for (int wgs : {1, 2, 3, 8})
                for (int solo = 0; solo < 2; ++solo)
                    for (int stm : {1, 2}) check_table(geom(rows, 768), (int)rows.size(), 256, k, solo != 0, stm, false, 4, wgs, 12, 130);
    std::mt19937_64 rng(0xdeadbeef);
    auto U = [&](long long lo, long long hi) { return lo + (long long)(rng() % (unsigned long long)(hi - lo + 1)); };
    for (int it = 0; it < n_fuzz; ++it) {
        const int F = (int)U(1, 32);
        std::vector<long long> rows(F);
        const int kind = (int)U(0, 3);
        for (auto& n : rows) {
            switch (kind) {
                case 0: n = U(0, 2000); break;
                case 1: n = U(0, 3000000); break;
                case 2: n = U(0, 1) ? U(0, 300) : U(100000, 2000000); break;          // collapsed low-cardinality fields beside big ones
                default: n = 1LL << U(0, 21); break;
            }
        }
        const int k = (int)U(1, 192), n_cu = (int)U(1, 304), wgs = (int)U(1, 8), waves = U(0, 1) ? 4 : 8;
        const bool solo = U(0, 3) == 0, forced = U(0, 7) == 0;
        S1GeomHost gg = geom(rows, 32 * (int)U(1, 64));
        if (U(0, 3) == 0) {                                          // some fields switched off (not scanned), at least one left
            const int keep = (int)U(0, F - 1);
            for (int f = 0; f < F; ++f)
                if (f != keep && U(0, 1)) gg.n_tiles[f] = 0;
        }
        check_table(gg, F, n_cu, k, solo, forced ? (int)U(1, 9) : (int)U(1, 2), forced, waves, wgs, (int)U(1, 24), (int)U(1, 400));
        check_layouts((int)U(0, 128), F, (int)U(1, 128), (int)U(1, 128));
    }
    std::printf("OK %lld tables and layouts checked\n", n_checked);
    return 0;
}

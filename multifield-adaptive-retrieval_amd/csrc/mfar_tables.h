// mfar_tables.h -- host-side geometry of the scorer: the stage-1 chunk tables and the byte layouts of the multi-GPU payloads.
// Plain C++ (no HIP): included by the device headers AND compiled on the CPU with -fsanitize=address,undefined by
// tests/host/ (tests/test_host_tables.py fuzzes the table builder over rows / fields / dim / depth / grid size).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

// One workgroup of a stage-1 pass scans one CHUNK: a contiguous run of 256-row tiles of one field.  The chunk table is built
// on the host from the fields' row counts (fields differ when the scanned slab holds each field's UNIQUE rows,
// mfar_screen.h): every field gets a share of the grid proportional to its tiles, chunks of a field are consecutive.
struct S1Chunk {
    int f;                  // field
    int t0, t1;             // tiles [t0, t1) of the field
    int n_rows;             // valid rows of the field (rows beyond are zero padding)
    long long base;         // element offset of the field's first tile inside the slab
    int tl0;                // index of this chunk's first tile among the SAMPLED tiles of its field (sample pass output slot)
    int ns;                 // tiles of this chunk the light sample pass scans (>= 1; per field: small fields are sampled deeper)
};

// Geometry of one scanned slab (per field) and one chunk table built from it for a list depth / grid size.
struct S1GeomHost {
    std::vector<long long> n_rows, base;   // per field: valid rows, element offset of the field inside the slab
    std::vector<int> n_tiles;              // per field: 256-row tiles
};
struct S1TableHost {
    int k = -1, wgs = -1;             // built for this list depth / grid size
    std::vector<S1Chunk> chunks;
    std::vector<int> fchunk, samp_n;  // [F + 1], [F]
    int n_chunks = 0, max_chunks = 0, samp_stride = 0, sample_tiles = 1;
    long long total_tiles = 0;
    long long thresholded_tiles = 0;  // tiles of the fields whose sample publishes at least k values (= yields a threshold)
    // two-level merge: a field cut into more chunks than one merge workgroup can hold (few fields, or a single-field pass)
    // is merged in GROUPS of consecutive chunks first (mfar_select.h MergeParams)
    bool two_level = false;
    std::vector<int> gchunk, fgroup, gfield;   // [n_groups + 1] chunk boundaries, [F + 1] groups of a field, [n_groups] field of a group
    int n_groups = 0, max_group_chunks = 0, max_groups = 0;
};

// Chunk table of a scanned slab for list depth k (mfar_stage1.h).  A field's share of the grid follows its tiles; the list
// merge holds n_chunks * k keys of one field, which caps the chunks of a field.
//   n_cu    compute units of the device;  wgs  workgroups per CU the grid is sized for
//   waves   waves per workgroup of the pass: wave blocks published per sampled tile
//   sample_div / append_target   sample-pass sizing (MFAR_SAMPLE_DIV = 12, MFAR_APPEND_TARGET = 130)
//   group_chunks   > 0: merge in two levels whenever a field has more chunks than this, in groups of this many chunks -- two
//                  launches of many small register-resident merges instead of one launch whose workgroups hold 48 keys per thread
// A field with n_tiles == 0 is NOT SCANNED by this table: it gets no chunk (fchunk[f] == fchunk[f + 1]), no sample slot and no
// share of the grid -- the fields the certified screen has switched off (mfar_hip.hip "AUTO-OFF"); the list merge writes their
// lists empty (MergeParams::skip_mask).
static inline void s1_build_table(const S1GeomHost& g, int F, int n_cu, int k, bool solo, int sample_tiles_max, bool sample_forced,
                                  int waves, int wgs, int sample_div, int append_target, S1TableHost& t, int group_chunks = 0) {
    const long long want = (long long)wgs * n_cu;
    const int cap = std::max(1, std::min(128, (64 * 256) / k));
    long long total_tiles = 0;
    for (int f = 0; f < F; ++f) total_tiles += g.n_tiles[f];
    t.chunks.clear();
    t.fchunk.assign(F + 1, 0);
    t.samp_n.assign(F, 0);
    t.max_chunks = 0;
    std::vector<int> cf(F);
    const int l2cap = std::max(1, std::min(cap, 8192 / k));   // lists the second level merges per field (register-resident keys)
    t.two_level = false;
    // SMALL FIELDS (low-cardinality fields collapse to a few unique rows: STaRK-prime `type` has ten texts).  The threshold of a
    // field is the k-th best of the values its sample publishes, 8 per sampled tile; a field that cannot publish k values even
    // when sampled whole has NO threshold, and then every row of a tile survives into the lists (measured: 0.235 ms per tile and
    // workgroup in the wide pass against ~0.03 ms with a threshold).  One workgroup walking such a field tile after tile held
    // up the whole launch (structured 1 M x 8 corpus, 6 tiles of a ten-text field in one chunk: scan 4.3 ms instead of 1.7).
    //   * a field without a possible threshold is cut into one-tile chunks (its cost is then bounded by one tile);
    //   * any field gets enough chunks to sample min(its tiles, tiles that publish 3 k values) at <= 4 tiles per chunk, so its
    //     sample workgroups stay as short as everybody's; the whole of such a field may be sampled (its bytes do not matter).
    // The chunks these rules add are taken from the largest fields, so the grid stays one wave of workgroups.
    const long long need_tiles = (3LL * k + 7) / 8;            // sampled tiles that publish 3 k values
    std::vector<int> floor_cf(F, 0);                           // > 0: the field was cut by one of the two rules
    {
        std::vector<long long> want_cf(F);
        long long extra = 0, spare = 0;
        for (int f = 0; f < F; ++f) {
            if (g.n_tiles[f] == 0) {               // not scanned
                cf[f] = 0;
                want_cf[f] = 0;
                continue;
            }
            const long long tiles = std::max(1, g.n_tiles[f]);
            const long long lim = std::min<long long>((long long)cap * l2cap, tiles);
            long long c = solo ? want : (want * g.n_tiles[f] + total_tiles / 2) / std::max(1LL, total_tiles);
            c = std::max(1LL, std::min(c, lim));
            const long long fl = std::max(1LL, std::min(8 * tiles < k ? tiles : (std::min(tiles, need_tiles) + 3) / 4, lim));
            cf[f] = (int)c;
            want_cf[f] = fl;
            if (fl > c) extra += fl - c;
            else spare += c - fl;
        }
        // the extra chunks come out of the fields that have more than their own floor; when nobody has (many equal fields),
        // only the fields without a possible threshold are cut (they must be) and the grid grows by those few workgroups
        for (int f = 0; f < F; ++f) {
            const bool hard = g.n_tiles[f] > 0 && 8LL * g.n_tiles[f] < k;
            if (want_cf[f] > cf[f] && (hard || solo || spare >= extra)) {
                cf[f] = (int)want_cf[f];
                floor_cf[f] = cf[f];
            }
        }
        if (!solo && extra > 0 && spare >= extra)
            for (int f = 0; f < F; ++f)
                if (!floor_cf[f] && cf[f] > want_cf[f]) cf[f] -= (int)(((cf[f] - want_cf[f]) * extra + spare - 1) / spare);
        // exactly one wave of workgroups where the rules allow it: a grid of a few workgroups more leaves them waiting for the
        // first to finish (a mid-size field's short chunks finish early: 515 workgroups, scan 2.36 ms instead of 1.85), a few
        // less idles CUs.  Short of a wave: the field with the longest chunks gets one more; over: the field with the shortest gives one up.
        if (!solo) {
            long long sum = 0;
            for (int f = 0; f < F; ++f) sum += cf[f];
            for (; sum != want; sum += sum < want ? 1 : -1) {
                int best = -1;
                double key = 0.0;
                for (int f = 0; f < F; ++f) {
                    if (cf[f] == 0) continue;      // a field that is not scanned takes no part in the balance
                    const long long tiles = std::max(1, g.n_tiles[f]);
                    const long long lim = std::min<long long>((long long)cap * l2cap, tiles);
                    const double tpc = (double)tiles / cf[f];
                    if (sum < want ? (cf[f] < lim && !floor_cf[f] && (best < 0 || tpc > key))
                                   : (cf[f] > std::max<long long>(1, want_cf[f]) && !floor_cf[f] && (best < 0 || tpc < key))) {
                        best = f;
                        key = tpc;
                    }
                }
                if (best < 0) break;
                cf[best] += sum < want ? 1 : -1;
            }
        }
    }
    long long n_chunks = 0;
    for (int f = 0; f < F; ++f) {
        n_chunks += cf[f];
        if (cf[f] > cap || (group_chunks > 0 && cf[f] > group_chunks)) t.two_level = true;
        t.max_chunks = std::max(t.max_chunks, cf[f]);
    }
    // tiles per workgroup in the sample pass, per field: more tiles = tighter starting thresholds = fewer appends in the full
    // pass, at the price of reading those tiles twice; at most 1/12 of a chunk and 4096 published values per (query, field)
    // (measured at 1 M x 8: 2 tiles pay off for the HBM-bound 16-bit passes, 1 for the MFMA-bound fp32 pass).
    // The threshold is the k-th best of the field's sampled rows, so a chunk expects k * (its rows) / (sampled rows of the field)
    // appends per query: about 85 at 1 M x 8.  Long chunks of many-field shards (1.25 M x 16: 140 tiles per chunk, 32 chunks
    // per field) would see ~280 with that fixed size -- past the compaction trigger, and every compaction drains the whole
    // workgroup's prefetch ring (measured there: selection epilogue 1.85 of 6.2 ms) -- so the sample grows until a chunk
    // expects no more than ~130 appends (MFAR_APPEND_TARGET; measured there: 75 .. 130 within 3 %, stage 1 6.8 -> 5.6 ms).
    // Short chunks (small shards, many fields) and the small fields above: the sample yields a threshold only when it
    // publishes at least k values per (query, field) and a useful one from about 3 k; without a threshold every list compacts
    // on nearly every tile.  Spend up to a sixth of a chunk on it -- the whole chunk in a field that was cut for this.
    std::vector<int> ns(F, 1);
    t.sample_tiles = 1;
    for (int f = 0; f < F; ++f) {
        if (cf[f] == 0) continue;
        const long long tpc = std::max(1LL, (long long)g.n_tiles[f] / cf[f]);      // tiles of the field's shortest chunk
        const long long tpc_hi = std::max(1LL, ((long long)g.n_tiles[f] + cf[f] - 1) / cf[f]);   // ... of its longest
        long long v = std::max(1LL, std::min<long long>(sample_tiles_max, tpc / sample_div));
        if (!sample_forced) {
            const long long want_tiles = ((long long)k * tpc + (long long)append_target * cf[f] / 2) / ((long long)append_target * cf[f]);
            v = std::max(v, std::min(want_tiles, std::max(1LL, tpc / sample_div)));
            const long long st_cap = cf[f] == floor_cf[f] ? tpc : std::max(1LL, tpc / 6);
            while (v < st_cap && 2LL * waves * cf[f] * v < 3LL * k) ++v;
            // no threshold at all (fewer than k values) costs 8 x a normal tile in the full pass: rather sample short chunks whole
            // (90 % empty 129 k x 22: 56 tiles per field in 23 chunks, 184 values from one tile each -- stage 1 1.34 ms)
            while (v < tpc_hi && 2LL * waves * std::min<long long>(g.n_tiles[f], cf[f] * v) < (long long)k) ++v;
        }
        while (v > 1 && 2LL * waves * cf[f] * v > 4096) --v;
        ns[f] = (int)v;
        t.sample_tiles = std::max(t.sample_tiles, ns[f]);
    }
    t.samp_stride = 0;
    t.thresholded_tiles = 0;
    for (int f = 0; f < F; ++f) {
        t.fchunk[f] = (int)t.chunks.size();
        int tl = 0;
        for (int c = 0; c < cf[f]; ++c) {
            S1Chunk ck = {};
            ck.f = f;
            ck.t0 = (int)(((long long)c * g.n_tiles[f]) / cf[f]);
            ck.t1 = (int)(((long long)(c + 1) * g.n_tiles[f]) / cf[f]);
            ck.n_rows = (int)g.n_rows[f];
            ck.base = g.base[f];
            ck.tl0 = tl;
            ck.ns = std::max(1, std::min(ns[f], ck.t1 - ck.t0));
            tl += std::min(ck.ns, ck.t1 - ck.t0);
            t.chunks.push_back(ck);
        }
        t.samp_n[f] = waves * tl;
        if (2LL * waves * tl >= k) t.thresholded_tiles += g.n_tiles[f];
        t.samp_stride = std::max(t.samp_stride, waves * tl);
    }
    t.fchunk[F] = (int)t.chunks.size();
    t.n_chunks = (int)t.chunks.size();
    t.total_tiles = total_tiles;
    t.gchunk.clear();
    t.gfield.clear();
    t.fgroup.assign(F + 1, 0);
    t.max_group_chunks = t.max_groups = 0;
    if (t.two_level) {
        for (int f = 0; f < F; ++f) {
            t.fgroup[f] = (int)t.gfield.size();
            int gs = (cf[f] + l2cap - 1) / l2cap;                        // chunks per group (<= cap by construction)
            if (group_chunks > 0) gs = std::max(gs, std::min(std::min(group_chunks, cap), cf[f]));
            for (int c0 = 0; c0 < cf[f]; c0 += gs) {
                t.gchunk.push_back(t.fchunk[f] + c0);
                t.gfield.push_back(f);
                t.max_group_chunks = std::max(t.max_group_chunks, std::min(gs, cf[f] - c0));
            }
            t.max_groups = std::max(t.max_groups, (int)t.gfield.size() - t.fgroup[f]);
        }
        t.fgroup[F] = (int)t.gfield.size();
        t.gchunk.push_back(t.n_chunks);
        t.n_groups = (int)t.gfield.size();
    }
    t.k = k;
    t.wgs = wgs;
}

// ---- byte layouts of the multi-GPU payloads (every section 256-byte aligned)
struct PayloadLayout {
    long long hdr, ids, scores, cand, ncand, x, total;
};
static inline long long mfar_up256(long long v) { return (v + 255) & ~255LL; }
static inline PayloadLayout payload_layout(int Q, int F, int k1) {
    PayloadLayout L;
    const long long C = (long long)F * k1;
    L.hdr = 0;
    L.ids = mfar_up256(64);                      // sizeof(PayloadHeader)
    L.scores = mfar_up256(L.ids + (long long)Q * F * k1 * 8);
    L.cand = mfar_up256(L.scores + (long long)Q * F * k1 * 4);
    L.ncand = mfar_up256(L.cand + (long long)Q * C * 8);
    L.x = mfar_up256(L.ncand + (long long)Q * 4);
    L.total = mfar_up256(L.x + (long long)Q * C * F * 4);
    return L;
}
struct MergeWsLayout {
    long long lids, lsc, cand, ncand, x, total;
};
static inline MergeWsLayout merge_ws_layout(int Q, int F, int k1) {
    MergeWsLayout L;
    const long long C = (long long)F * k1;
    L.lids = 0;
    L.lsc = mfar_up256(L.lids + (long long)Q * F * k1 * 8);
    L.cand = mfar_up256(L.lsc + (long long)Q * F * k1 * 4);
    L.ncand = mfar_up256(L.cand + (long long)Q * C * 8);
    L.x = mfar_up256(L.ncand + (long long)Q * 4);
    L.total = mfar_up256(L.x + (long long)Q * C * F * 4);
    return L;
}
struct ListsLayout {
    long long ids, scores, total;
};
static inline ListsLayout lists_layout(int Q, int F, int k1) {
    ListsLayout L;
    L.ids = 0;
    L.scores = mfar_up256((long long)Q * F * k1 * 8);
    L.total = mfar_up256(L.scores + (long long)Q * F * k1 * 4);
    return L;
}
struct TopkLayout {
    long long ids, scores, ncand, flag, total;
};
static inline TopkLayout topk_layout(int Q, int k2) {
    TopkLayout L;
    L.ids = 0;
    L.scores = mfar_up256((long long)Q * k2 * 8);
    L.ncand = mfar_up256(L.scores + (long long)Q * k2 * 4);
    L.flag = L.ncand + (long long)Q * 4;        // one int32: this rank's certificate flag of the batch (travels with the top-k)
    L.total = mfar_up256(L.flag + 4);
    return L;
}

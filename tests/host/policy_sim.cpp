// CPU-only check of the certified screen's adaptive policy (multifield-adaptive-retrieval_amd/csrc/mfar_policy.h -- the very code
// mfar_hip.hip drives from the certificate flags of finished launches).  Built by tests/test_host_policy.py with
// g++ -fsanitize=address,undefined; every scenario aborts with a message on the first violated expectation.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "mfar_policy.h"

#define REQUIRE(cond)                                                        \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::fprintf(stderr, "FAILED %s (line %d)\n", #cond, __LINE__);  \
            std::abort();                                                    \
        }                                                                    \
    } while (0)

// one launch: plan, "run" it against per-field failure probabilities, feed the flags back (optionally `lag` launches late)
struct Sim {
    ScreenPolicy pol;
    int F;
    std::mt19937_64 rng{0xdeadbeef};
    long long n_exact_field_launches = 0, n_screen_field_launches = 0, n_reported = 0, n_launches = 0;
    explicit Sim(int F_) : F(F_) {}
    bool launch(const double* p_fail, bool strict = false) {
        uint32_t exact = 0, skip = 0;
        pol.plan(F, &exact, &skip);
        REQUIRE((skip & ~exact) == 0);                      // the screen never leaves out a field the exact pass does not cover
        REQUIRE(skip == exact || skip == 0);                // ... and a probe screens everything
        int flags[MFAR_POLICY_MAX_FIELDS] = {0}, probe[MFAR_POLICY_MAX_FIELDS] = {0};
        int any = 0;
        uint32_t screened = 0, probed = 0;
        for (int f = 0; f < F; ++f) {
            const uint32_t bit = 1u << f;
            const bool fails = std::uniform_real_distribution<double>(0, 1)(rng) < p_fail[f];
            if (exact & bit) n_exact_field_launches++;
            if (!(skip & bit)) n_screen_field_launches++;
            if (!(exact & bit)) {
                screened |= bit;
                flags[f] = fails;
                any |= fails;
            } else if (!(skip & bit)) {
                probed |= bit;
                probe[f] = fails;
            }
        }
        if (any && !pol.inline_repair) n_reported++;        // the caller pays a drain + redo
        n_launches++;
        return pol.feed(F, flags, probe, any, screened, probed, strict);
    }
};

int main() {
    {   // a clean corpus: nothing ever changes
        Sim s(8);
        const double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 1000; ++i) REQUIRE(!s.launch(p));
        REQUIRE(s.pol.off_mask == 0 && !s.pol.inline_repair && s.pol.n_probes == 0 && s.n_exact_field_launches == 0);
    }
    {   // two of eight fields fail every launch: exactly those are switched off after 12 launches, inline repair meanwhile, probes keep
        // them off; once they are off the remaining fields never fail, so reporting comes back
        Sim s(8);
        const double p[8] = {0, 0, 1, 0, 0, 1, 0, 0};
        for (int i = 0; i < 11; ++i) s.launch(p);
        REQUIRE(s.pol.off_mask == 0 && s.pol.inline_repair);                 // (inline from the 4th failed launch on)
        REQUIRE(s.n_reported == 4);
        s.launch(p);
        REQUIRE(s.pol.off_mask == ((1u << 2) | (1u << 5)) && s.pol.n_off == 2);
        for (int i = 0; i < 640; ++i) s.launch(p);
        REQUIRE(s.pol.off_mask == ((1u << 2) | (1u << 5)) && s.pol.n_on == 0 && s.pol.n_probes == 10);
        REQUIRE(!s.pol.inline_repair);                                        // the ON fields are clean: back to reporting
        REQUIRE(s.n_reported == 4);
        // the data changes (rows rewritten): two clean probes switch each field back on
        const double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 64 * 2; ++i) s.launch(q);
        REQUIRE(s.pol.off_mask == 0 && s.pol.n_on == 2);
    }
    {   // every field fails: all off; a launch is then the exact pass and nothing else, except one probe in probe_every launches
        Sim s(8);
        const double p[8] = {1, 1, 1, 1, 1, 1, 1, 1};
        for (int i = 0; i < 12; ++i) s.launch(p);
        REQUIRE(s.pol.off_mask == 0xFFu);
        const long long s0 = s.n_screen_field_launches;
        for (int i = 0; i < 6400; ++i) s.launch(p);
        REQUIRE(s.pol.off_mask == 0xFFu && s.n_screen_field_launches - s0 == 100 * 8);      // 100 probes x 8 fields, nothing else screened
    }
    {   // a field that fails in 30 % of the launches is cheaper screened + repaired than scanned exactly: it stays on
        Sim s(4);
        const double p[4] = {0.3, 0, 0, 0};
        for (int i = 0; i < 5000; ++i) s.launch(p);
        REQUIRE(s.pol.n_off <= 2);                             // (12 of 16 at p = 0.3: about once in 10^4 windows)
        REQUIRE(s.pol.inline_repair);                          // ... and its repairs run on the device, not through drains
    }
    {   // strict (bf16): only 16 of 16 switches off
        Sim s(2);
        const double p[2] = {0.9, 0};
        for (int i = 0; i < 300; ++i) s.launch(p, true);
        const long long offs = s.pol.n_off;
        Sim t(2);
        for (int i = 0; i < 300; ++i) t.launch(p, false);
        REQUIRE(t.pol.n_off >= 1 && offs <= t.pol.n_off);
        Sim u(2);
        const double p1[2] = {1, 0};
        for (int i = 0; i < 15; ++i) u.launch(p1, true);
        REQUIRE(u.pol.off_mask == 0);
        u.launch(p1, true);
        REQUIRE(u.pol.off_mask == 1u);
    }
    {   // mode 0: nothing is ever switched off; switching the mode off clears the set
        Sim s(3);
        s.pol.set_mode(0);
        const double p[3] = {1, 1, 1};
        for (int i = 0; i < 100; ++i) s.launch(p);
        REQUIRE(s.pol.off_mask == 0 && s.n_exact_field_launches == 0 && s.pol.inline_repair);
        s.pol.set_mode(1);
        for (int i = 0; i < 20; ++i) s.launch(p);
        REQUIRE(s.pol.off_mask == 7u);
        s.pol.set_mode(0);
        REQUIRE(s.pol.off_mask == 0);
    }
    {   // inline repair is reversible: a burst of failures, then a clean stretch of 16 launches
        Sim s(4);
        const double bad[4] = {0.5, 0.5, 0, 0}, good[4] = {0, 0, 0, 0};
        for (int i = 0; i < 10; ++i) s.launch(bad);
        REQUIRE(s.pol.inline_repair);
        for (int i = 0; i < 16; ++i) s.launch(good);
        REQUIRE(!s.pol.inline_repair);
    }
    {   // 32 fields, random failure rates, feedback consumed: invariants only (the REQUIREs inside launch())
        Sim s(32);
        double p[32];
        for (int f = 0; f < 32; ++f) p[f] = (f % 5 == 0) ? 1.0 : (f % 7 == 0 ? 0.5 : 0.0);
        for (int i = 0; i < 3000; ++i) s.launch(p);
        for (int f = 0; f < 32; ++f)
            if (f % 5 == 0) REQUIRE((s.pol.off_mask >> f) & 1u);
            else if (p[f] == 0.0) REQUIRE(!((s.pol.off_mask >> f) & 1u));
    }
    {   // TIER 2 (round 6): armed by the first launch with a failed FIRST certificate (whether or not tier 2 then finished the lists), disarmed
        // after 256 clean launches in a row; the flags AUTO-OFF sees are those after tier 2 -- lists tier 2 finishes switch nothing off
        ScreenPolicy pol;
        int clean[MFAR_POLICY_MAX_FIELDS] = {0}, probe[MFAR_POLICY_MAX_FIELDS] = {0};
        for (int i = 0; i < 100; ++i) pol.feed(4, clean, probe, 0, 0xFu, 0u, false, 0);
        REQUIRE(!pol.t2_armed);
        pol.feed(4, clean, probe, 0, 0xFu, 0u, false, 1);                  // first certificates failed, tier 2 finished every list
        REQUIRE(pol.t2_armed && pol.off_mask == 0 && !pol.inline_repair);
        for (int i = 0; i < 1000; ++i) pol.feed(4, clean, probe, 0, 0xFu, 0u, false, 1);
        REQUIRE(pol.t2_armed && pol.off_mask == 0 && !pol.inline_repair && pol.n_off == 0);
        for (int i = 0; i < 255; ++i) pol.feed(4, clean, probe, 0, 0xFu, 0u, false, 0);
        REQUIRE(pol.t2_armed);
        pol.feed(4, clean, probe, 0, 0xFu, 0u, false, 0);
        REQUIRE(!pol.t2_armed);                                            // 256 clean launches: a clean corpus pays nothing again
        int bad[MFAR_POLICY_MAX_FIELDS] = {0};
        bad[2] = 1;
        pol.feed(4, bad, probe, 1, 0xFu, 0u, false, 0);                    // a hard failure (tier 2 not even tried) arms it as well
        REQUIRE(pol.t2_armed);
    }
    {   // DEEP SCAN (auto mode): a field whose first certificates fail in 8 of 16 evaluated launches is promoted; an overflowing deep list
        // demotes it at once and blocks it for deep_block launches; after deep_renew launches a deep field is evaluated afresh; mode 0 = never
        ScreenPolicy pol;
        pol.deep_mode = 1;
        int t1[MFAR_POLICY_MAX_FIELDS] = {0}, ok[MFAR_POLICY_MAX_FIELDS] = {0}, failed[MFAR_POLICY_MAX_FIELDS] = {0};
        t1[1] = 1;
        for (int i = 0; i < 7; ++i) pol.feed_deep(4, t1, ok, 0xFu, 0u);
        REQUIRE(pol.deep_mask == 0);
        pol.feed_deep(4, t1, ok, 0xFu, 0u);
        REQUIRE(pol.deep_mask == 2u && pol.n_deep_on == 1);
        for (int i = 0; i < 1023; ++i) pol.feed_deep(4, t1, ok, 0xFu & ~pol.deep_mask, pol.deep_mask);
        REQUIRE(pol.deep_mask == 2u);
        pol.feed_deep(4, t1, ok, 0xFu & ~pol.deep_mask, pol.deep_mask);   // renewal: evaluated afresh ...
        REQUIRE(pol.deep_mask == 0);
        for (int i = 0; i < 8; ++i) pol.feed_deep(4, t1, ok, 0xFu, 0u);   // ... and promoted again while it keeps failing
        REQUIRE(pol.deep_mask == 2u && pol.n_deep_on == 2);
        failed[1] = 1;
        pol.feed_deep(4, t1, failed, 0xFu & ~pol.deep_mask, pol.deep_mask);      // a deep list overflowed: demoted at once
        REQUIRE(pol.deep_mask == 0 && pol.n_deep_demoted == 1);
        for (int i = 0; i < 4000; ++i) pol.feed_deep(4, t1, ok, 0xFu, 0u);
        REQUIRE(pol.deep_mask == 0);                                       // blocked
        for (int i = 0; i < 200; ++i) pol.feed_deep(4, t1, ok, 0xFu, 0u);
        REQUIRE(pol.deep_mask == 2u);                                      // the block has run out
        pol.deep_mode = 0;
        pol.feed_deep(4, t1, ok, 0xFu, 0u);
        REQUIRE(pol.deep_mask == 0);
        ScreenPolicy off;                                                  // default: never
        for (int i = 0; i < 100; ++i) off.feed_deep(4, t1, ok, 0xFu, 0u);
        REQUIRE(off.deep_mask == 0 && off.n_deep_on == 0);
    }
    {   // TIER 2's RESCAN is armed separately: only once a list has asked for it (its candidates were not in the launch's own chunk lists),
        // disarmed after 256 launches in which none did -- a corpus whose failed lists are all served from the first scan never has the
        // full-width rescan kernel in its launches' tails
        ScreenPolicy pol;
        for (int i = 0; i < 1000; ++i) pol.feed_rescan(false);
        REQUIRE(!pol.t2_rescan_armed && pol.n_rescan_armed == 0);
        pol.feed_rescan(true);
        REQUIRE(pol.t2_rescan_armed && pol.n_rescan_armed == 1);
        for (int i = 0; i < 255; ++i) pol.feed_rescan(false);
        REQUIRE(pol.t2_rescan_armed);
        pol.feed_rescan(true);                                             // (asked again: the count starts over)
        for (int i = 0; i < 255; ++i) pol.feed_rescan(false);
        REQUIRE(pol.t2_rescan_armed);
        pol.feed_rescan(false);
        REQUIRE(!pol.t2_rescan_armed && pol.n_rescan_armed == 1);
        pol.feed_rescan(true);
        REQUIRE(pol.t2_rescan_armed && pol.n_rescan_armed == 2);
    }
    std::printf("OK policy scenarios\n");
    return 0;
}

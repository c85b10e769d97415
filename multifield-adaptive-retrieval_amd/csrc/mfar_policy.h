// mfar_policy.h -- the adaptive policy of the certified screen as a host-only state machine (plain C++: mfar_hip.hip drives it from the
// certificate flags of finished launches; tests/host/policy_sim.cpp runs it on the CPU under ASan / UBSan, tests/test_host_policy.py).
//
// The certificate is data dependent.  A list fails when more than k' - k of its field's unique rows sit within ~2 eps of the k-th best score:
// clusters of near-duplicate rows (not bit-identical, so the unique-row build keeps them apart) do that list after list, and every failure
// costs the exact pass of that field ON TOP of the screen; a caller that only asked for a report also pays a pipeline drain and a second
// launch.  Three decisions bound the worst case at the price of the exact pass; none of them latches, none changes a result bit:
//   * AUTO-OFF, per field: >= off_fails failures in the field's last 16 screened launches -> the field is switched off (the screen's chunk
//     table leaves it out, the exact pass writes its lists from the begin phase).  Every probe_every-th launch that has a switched-off field
//     screens it anyway (a PROBE: certificate evaluated, lists discarded); on_clean clean probes in a row switch it back on.
//   * inline repair: >= 4 of the last 16 screened launches had a failure among the fields that were ON -> mfar_stage1_finish repairs on the
//     device even when asked to report only; <= 1 of the last 16 (a full window) -> back to reporting.
//   * the first observed failure tells the caller to activate ROW MODE for eligible fields (feed() returns true).
//   * TIER 2 armed / disarmed (below).
#pragma once
#include <cstdint>

#define MFAR_POLICY_MAX_FIELDS 32
struct ScreenPolicy {
    int mode = 1;             // 0 never switch a field off, 1 auto
    int off_fails = 12;       // failures of the last 16 screened launches that switch a field off
    int probe_every = 64;     // launches (with a switched-off field) between probes
    int on_clean = 2;         // consecutive clean probes that switch a field back on
    unsigned short hist[MFAR_POLICY_MAX_FIELDS] = {0};      // per field: 1 = failed, newest in bit 0
    unsigned char clean[MFAR_POLICY_MAX_FIELDS] = {0};
    unsigned short any_hist = 0;
    int any_n = 0;
    long long launches = 0, n_off = 0, n_on = 0, n_probes = 0;
    uint32_t off_mask = 0;    // fields that are switched off now
    bool inline_repair = false;
    // TIER 2 (mfar_screen.h "threshold rescan") is ARMED by the first launch with a failed first certificate -- from then on its (idle
    // when nothing fails) kernels follow every certificate -- and disarmed after t2_disarm_after clean launches in a row: a corpus whose
    // lists all certify never pays for their launches.  The flags fed above are those AFTER tier 2: a field is switched off only when
    // tier 2 cannot finish its lists either.
    bool t2_armed = false;
    int t2_clean = 0, t2_disarm_after = 256;
    // ... and tier 2's RESCAN separately: tier 2 normally reads a failed list's candidates out of the chunk lists the launch's own scan
    // wrote; the rescan of a field -- a full-width scan kernel in the tail of the launch, which cannot start before the NEXT launch's scan
    // lets go of the register file, even when it has nothing to do -- is only enqueued while a list of the last t2_disarm_after launches
    // asked for it.  A list that asks while it is not armed goes to the exact pass (and arms it).
    bool t2_rescan_armed = false;
    int t2_rescan_clean = 0;
    long long n_rescan_armed = 0;
    void feed_rescan(bool wanted) {
        if (wanted) {
            if (!t2_rescan_armed) n_rescan_armed++;
            t2_rescan_armed = true;
            t2_rescan_clean = 0;
        } else if (t2_rescan_armed && ++t2_rescan_clean >= t2_disarm_after) t2_rescan_armed = false;
    }
    // DEEP SCAN (mfar_screen.h): a field whose lists failed their FIRST certificate in >= deep_on of its last 16 evaluated launches gets no
    // first attempt any more -- its one scan collects the complete candidate set from a sample-derived threshold and tier 2's back half
    // finishes the lists (one scan instead of scan + rescan).  After deep_renew launches as a deep field it is evaluated afresh.
    // A deep field whose lists overflow (its sample-derived threshold is looser than the rescan's: clusters of near-duplicates put thousands
    // of rows above it) is DEMOTED at once -- an overflowing list sends its whole field to the exact pass -- and may not be promoted again
    // for deep_block launches.
    int deep_mode = 0;        // 0 never, 1 auto
    int deep_on = 8, deep_renew = 1024, deep_block = 4096;
    unsigned short t1hist[MFAR_POLICY_MAX_FIELDS] = {0};
    int deep_age[MFAR_POLICY_MAX_FIELDS] = {0}, deep_hold[MFAR_POLICY_MAX_FIELDS] = {0};
    uint32_t deep_mask = 0;
    long long n_deep_on = 0, n_deep_demoted = 0;
    //   t1_fields[f] != 0: a list of field f failed its first certificate;  evaluated: fields that HAD a first certificate in that launch;
    //   was_deep: fields that ran as deep fields;  failed[f] != 0: a list of field f went to the exact pass in the end
    void feed_deep(int F, const int* t1_fields, const int* failed, uint32_t evaluated, uint32_t was_deep) {
        for (int f = 0; f < F && f < MFAR_POLICY_MAX_FIELDS; ++f) {
            const uint32_t bit = 1u << f;
            if (deep_hold[f] > 0) deep_hold[f]--;
            if (evaluated & bit) {
                t1hist[f] = (unsigned short)((t1hist[f] << 1) | (t1_fields[f] ? 1 : 0));
                if (deep_mode && !(deep_mask & bit) && deep_hold[f] == 0 && __builtin_popcount(t1hist[f]) >= deep_on) {
                    deep_mask |= bit;
                    deep_age[f] = 0;
                    n_deep_on++;
                }
            }
            if ((was_deep & bit) && (deep_mask & bit)) {
                if (failed[f]) {
                    deep_mask &= ~bit;
                    t1hist[f] = 0;
                    deep_hold[f] = deep_block;
                    n_deep_demoted++;
                } else if (++deep_age[f] >= deep_renew) {
                    deep_mask &= ~bit;
                    t1hist[f] = 0;
                }
            }
        }
        if (!deep_mode) deep_mask = 0;
    }

    // Begin of a screened all-fields launch over F fields: *exact = fields whose lists the exact pass writes in this launch, *skip = fields
    // its screen leaves out (equal, except in a probe launch, which screens everything).
    void plan(int F, uint32_t* exact, uint32_t* skip) {
        const uint32_t all = F >= 32 ? 0xFFFFFFFFu : ((1u << F) - 1u);
        *exact = mode ? (off_mask & all) : 0u;
        bool probe = false;
        if (*exact) {
            probe = ++launches % probe_every == 0;
            if (probe) n_probes++;
        }
        *skip = probe ? 0u : *exact;
    }

    // The flags of a FINISHED launch: flags[f] != 0 = a list of field f failed (fields in `screened`: screened for real, i.e. ON at the
    // time), probe_flags[f] != 0 = the quiet probe of switched-off field f failed (fields in `probed`), any != 0 = some ON field failed.
    //   strict   a field is only switched off when it failed in ALL of its last 16 launches (bf16 indexes: their exact pass is the VALU
    //            chain pass, ~30x a screened scan: switching off saves the field's share of the screened scan -- its chunk table leaves the
    //            field out like an fp32 index's -- and its certificate, never the chain pass)
    // Returns true when the launch had a failure among the ON fields.
    //   t1       the launch had a list that failed its FIRST certificate (whether or not tier 2 then finished it): arms tier 2
    bool feed(int F, const int* flags, const int* probe_flags, int any, uint32_t screened, uint32_t probed, bool strict, int t1 = 0) {
        if (screened) {
            if (t1 || any) {
                t2_armed = true;
                t2_clean = 0;
            } else if (t2_armed && ++t2_clean >= t2_disarm_after) t2_armed = false;
        }
        for (int f = 0; f < F && f < MFAR_POLICY_MAX_FIELDS; ++f) {
            const uint32_t bit = 1u << f;
            if (screened & bit) {
                hist[f] = (unsigned short)((hist[f] << 1) | (flags[f] ? 1 : 0));
                const int need = strict ? 16 : off_fails;
                if (mode && !(off_mask & bit) && __builtin_popcount(hist[f]) >= need) {
                    off_mask |= bit;
                    clean[f] = 0;
                    n_off++;
                }
            }
            if ((probed & bit) && (off_mask & bit)) {
                if (probe_flags[f]) clean[f] = 0;
                else if (++clean[f] >= on_clean) {
                    off_mask &= ~bit;
                    hist[f] = 0;
                    n_on++;
                }
            }
        }
        if (!screened) return false;
        any_hist = (unsigned short)((any_hist << 1) | (any ? 1 : 0));
        if (any_n < 16) any_n++;
        const int n_bad = __builtin_popcount(any_hist);
        if (n_bad >= 4) inline_repair = true;
        else if (n_bad <= 1 && any_n >= 16) inline_repair = false;
        return any != 0;
    }

    void set_mode(int m) {
        mode = m;
        if (!m) off_mask = 0;
    }
};

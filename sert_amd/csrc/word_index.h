// Host-side builder of the per-batch inverted index word -> batch rows.
//
// The reference's backward of the embedding lookup is Theano's
// AdvancedIncSubtensor1 into zeros_like(R_w) (autodiff of sert/models.py:180):
// a scatter-add with heavy duplicates (Zipfian tokens).  Batches are fixed
// slices [i*B, (i+1)*B) of the data set (models.py:322-326), so the sorted
// (word -> rows) lists can be built ONCE at upload; the device then does an
// order-fixed segmented gather-reduce instead of fp32 atomics: deterministic
// and free of hot-row serialisation.
//
// Levels: a word with more than kSegChunk occurrences is split into chunks whose
// partial sums are reduced again at the next level (tree), so no single wave
// walks more than kSegChunk rows.
#pragma once
#include <stdint.h>
#include <algorithm>
#include <vector>

namespace sert {

constexpr int kSegChunk = 64;
// HEAVY words (vectorspace): a word with more than kHeavyMinCount occurrences in a batch leaves the
// tree.  Its gradient row is a count-weighted sum over ALL batch rows, sum_i cnt[i][h] src[i, :],
// which one streaming pass over src computes for up to kHeavyMax words at once (kernels_seg.h:
// segsum_heavy) -- instead of fetching the same B rows once per occurrence through three tree levels.
// Zipfian tokens put over half of a batch's tokens on a dozen words (stop words, the padding id, the
// clipped tail of the synthetic stream): at C2 the tree then walks 280 k entries instead of 655 k.
constexpr int kHeavyMax = 16;            // counts of one batch row: 16 bytes, one load
constexpr int kHeavyMinCount = 4096;     // = kSegChunk^2: without its heavy words a tree has two levels
constexpr int kHeavyRowsPerBlock = 256;
constexpr int kSegMaxLevels = 6;
constexpr int kBundleItems = 8;
constexpr int kFusedMaxChunks = 32;   // level-1 chunk items one workgroup of segsum_upper_fused combines: one per lane group
                                      // (64, two per group, measured SLOWER than the separate level-2 launch: 19 vs 13 us)

struct SegItem {       // 16 bytes, read as int4 on the device
    int32_t begin;     // first entry (level 0: index into rows[]; level>0: partial row)
    int32_t end;       // one past the last entry
    int32_t dst;       // >= 0: destination row in the gradient table (final)
                       // <  0: partial row -(dst+1) of this level's output space
    int32_t slot;      // final items: rank of the word among the batch's distinct words
};

struct BatchIndex {
    int64_t rows_off = 0;                 // offset of this batch's level-0 entries in rows[]
    int nlevels = 0;
    int64_t item_off[kSegMaxLevels] = {}; // offset into items[]
    int32_t item_cnt[kSegMaxLevels] = {};
    int64_t part_off[kSegMaxLevels] = {}; // partial-row offset of level l's OUTPUT space
    int64_t part_rows = 0;                // total partial rows of this batch
    int64_t uw_off = 0;                   // offset of this batch's distinct words in uwords[]
    int32_t num_distinct = 0;             // U: distinct words of the batch
    // three-level trees only: the words whose level-1 chunks are summed again at level 2 ("heavy":
    // more than kSegChunk^2 occurrences), so that levels 1 and 2 can run as ONE launch
    // (segsum_upper_fused): {first level-1 item, number of chunk items, word, 0} per heavy word.
    int64_t heavy_off = 0;                // offset into heavy[] (in entries of four ints)
    int32_t heavy_cnt = 0;
    bool fused_upper_ok = false;          // nlevels == 3 and every heavy word has <= kFusedMaxChunks chunks
    // words summed densely (kHeavyMax per batch at most), outside the tree
    int32_t dense_cnt = 0;
    int32_t dense_word[kHeavyMax] = {};
    int32_t dense_slot[kHeavyMax] = {};   // (loglinear) their ranks among the batch's distinct words
    // row-grouped level 0 (build_word_index: row_groups > 1): the level-0 items are stored per XCD list -- list x
    // holds the items of the row groups x, x + 8, x + 16, ... in that order -- at item_off[0] + xcd_off[x]
    int32_t row_groups = 1;
    int32_t xcd_off[8] = {};
    int32_t xcd_cnt[8] = {};
    // level-0 BUNDLES (row_groups == 1): consecutive items handed to one lane group together -- up to kBundleItems items
    // holding up to kSegChunk entries in all -- so that the short items (half of a Zipfian batch's words occur once: one
    // row load in flight per lane group) travel eight to a group with eight loads in flight (kernels_seg.h:
    // segsum_rows_bundled).  bundles[bundle_off + k] = first item of bundle k, relative to item_off[0]; bundle_cnt + 1
    // values.  The items' entry ranges are consecutive, every item is still summed left to right: the same bits.
    int64_t bundle_off = 0;
    int32_t bundle_cnt = 0;
    // level 0 SORTED by length, longest items first, and every level-0 item's `slot` = its FIRST source row
    // (build_word_index: sort_level0; vectorspace only, where `slot` is otherwise unused): the eight items of a workgroup
    // are then equally long -- a workgroup lasts as long as its longest item --, and a one-entry item (half of a Zipfian
    // batch's words) fetches its row straight after the descriptor, one dependent round trip less
    bool slot_is_row = false;
};

struct WordIndex {
    std::vector<int32_t> rows;            // all batches, level-0 entries (source row ids)
    std::vector<SegItem> items;           // all batches, all levels
    std::vector<BatchIndex> batches;
    std::vector<int32_t> heavy;           // all batches, four ints per heavy word (BatchIndex::heavy_off)
    std::vector<int32_t> bundles;         // all batches, level-0 bundles (BatchIndex::bundle_off)
    int64_t max_part_rows = 0;
    // distinct words per batch (sorted) and, per token position, the rank of its word
    // among them: a gathered row is then computed ONCE per distinct word (loglinear)
    std::vector<int32_t> uwords;          // all batches
    std::vector<int32_t> slots;           // all batches, B*n per batch (want_slots only)
    std::vector<int32_t> rows_div;        // rows[] / n: the batch row of every level-0 entry (want_slots only)
    int32_t max_distinct = 0;
    // per batch: bit w = 1 iff word w occurs in the batch (words_per_batch 32-bit words each).
    // The optimiser takes the gradient of every other row as zero without reading it, and the
    // rows no token of the batch points to can be updated while the batch is still in flight.
    std::vector<uint32_t> touched_bits;
    int64_t bit_words = 0;
    // dense heavy words: per batch and batch row, kHeavyMax occurrence counts (uint8); empty if no
    // batch has any
    std::vector<uint8_t> dense_counts;    // [num_batches][B][kHeavyMax]
    bool any_dense = false;
    // (vectorspace) per batch and token position: the dense slot of the position's word, 255 = none -- the forward's gather
    // takes a dense word's row from LDS instead of fetching it once per occurrence (kernels_vs.h: vs_gather_mean_hot)
    std::vector<uint8_t> dense_tok_slot;  // [num_batches][B * n]
};

// ids: (num_batches*B*n) token ids of the complete batches, IdT wide.
// row_of_pos: entry value = pos / n (vectorspace: row of dh) or pos (loglinear: row of dG).
//
// row_groups > 1 (vectorspace word gradient, batches whose source matrix dh does not fit one XCD's L2): level 0
// is cut by SOURCE ROW as well as by word.  The batch rows form `row_groups` equal ranges; a level-0 item is
// the (<= kSegChunk) occurrences of one word inside ONE range, and the items of range g are handed to workgroups
// that run on XCD g % 8 (kernels_seg.h: XcdLists), so that a range's slice of dh (<= ~2.5 MB) is fetched into that
// XCD's L2 once and every further fetch of its rows is an L2 hit -- instead of 655 k row fetches spread uniformly
// over a 33.5 MB matrix that no L2 holds (C2: 264 MB crossed the fabric for a job whose compulsory traffic is
// 56 MB).  A word whose occurrences all sit in one item is stored finally by level 0; every other word's items
// write partial rows, numbered word-major (range order, then chunk order, inside a word), which the upper levels
// sum in that order: a fixed association, as before -- only a different one.
template <typename IdT>
bool build_word_index(const IdT* ids, int64_t num_batches, int B, int n, int vocab,
                      bool row_is_pos, WordIndex& out, bool want_slots = false, bool dense_heavy = false,
                      int row_groups = 1, bool sort_level0 = false) {
    const int64_t T = (int64_t)B * n;
    if (want_slots || row_is_pos || row_groups < 1) row_groups = 1;
    const int rows_per_group = (B + row_groups - 1) / row_groups;
    // vectorspace (rows = batch rows): the heavy words LEAVE the tree.  loglinear (want_slots: the tree
    // also carries per-position scalars and, in the per-token cross-check mode, the word gradient): they
    // stay in it, FLAGGED -- chunk items get slot = -1, and the V_e-wide per-word sums skip every item of
    // a flagged word (kernels_seg.h: SKIP_DENSE) because segsum_heavy computes those rows densely
    const bool flag_only = want_slots;
    dense_heavy = dense_heavy && (want_slots || !row_is_pos) && n <= 255;
    std::vector<int8_t> heavy_slot;       // word -> its slot among the batch's dense words, or -1
    if (dense_heavy) {
        heavy_slot.assign((size_t)vocab, (int8_t)-1);
        out.dense_counts.assign((size_t)(num_batches * B) * kHeavyMax, (uint8_t)0);
        if (!want_slots) out.dense_tok_slot.assign((size_t)(num_batches * T), (uint8_t)255);
    }
    out.rows.resize((size_t)(num_batches * T));
    out.batches.resize((size_t)num_batches);
    if (want_slots) out.slots.resize((size_t)(num_batches * T));
    std::vector<int32_t> slot_of((size_t)(want_slots ? vocab : 0), 0);
    std::vector<int32_t> count((size_t)vocab + 1, 0);
    std::vector<int32_t> touched;
    touched.reserve((size_t)std::min<int64_t>(T, vocab));
    std::vector<int32_t> start;  // per touched word
    out.bit_words = ((((int64_t)vocab + 31) / 32) + 3) / 4 * 4;
    out.touched_bits.assign((size_t)(num_batches * out.bit_words), 0u);
    for (int64_t bi = 0; bi < num_batches; ++bi) {
        const IdT* x = ids + bi * T;
        BatchIndex& bx = out.batches[(size_t)bi];
        bx.rows_off = bi * T;
        // counting sort by word id (stable in position)
        touched.clear();
        for (int64_t p = 0; p < T; ++p) {
            const int64_t wid64 = (int64_t)x[p];
            if (wid64 >= vocab) return false;  // token id outside the vocabulary
            const int32_t wid = (int32_t)wid64;
            if (count[wid]++ == 0) touched.push_back(wid);
        }
        std::sort(touched.begin(), touched.end());
        {
            uint32_t* bits = out.touched_bits.data() + bi * out.bit_words;
            for (int32_t wid : touched) bits[wid >> 5] |= 1u << (wid & 31);
        }
        // dense heavy words: the (at most kHeavyMax) words with more than kHeavyMinCount occurrences,
        // heaviest first; only worth a pass over all rows if they hold a good share of the tokens
        bx.dense_cnt = 0;
        if (dense_heavy) {
            std::vector<std::pair<int32_t, int32_t>> cand;   // (count, word)
            for (int32_t wid : touched)
                if (count[wid] > kHeavyMinCount) cand.push_back({count[wid], wid});
            std::sort(cand.begin(), cand.end(), [](const std::pair<int32_t, int32_t>& a, const std::pair<int32_t, int32_t>& b) {
                return a.first != b.first ? a.first > b.first : a.second < b.second;
            });
            if ((int)cand.size() > kHeavyMax) cand.resize(kHeavyMax);
            int64_t held = 0;
            for (const auto& c : cand) held += c.first;
            if (!cand.empty() && held * 8 >= T) {   // >= 1/8 of the batch's tokens
                for (size_t h = 0; h < cand.size(); ++h) {
                    bx.dense_word[h] = cand[h].second;
                    bx.dense_slot[h] = (int32_t)(std::lower_bound(touched.begin(), touched.end(), cand[h].second) - touched.begin());
                    heavy_slot[(size_t)cand[h].second] = (int8_t)h;
                }
                bx.dense_cnt = (int32_t)cand.size();
                uint8_t* dc = out.dense_counts.data() + (size_t)(bi * B) * kHeavyMax;
                uint8_t* ts = out.dense_tok_slot.empty() ? nullptr : out.dense_tok_slot.data() + (size_t)(bi * T);
                for (int64_t p = 0; p < T; ++p) {
                    const int8_t h = heavy_slot[(size_t)x[p]];
                    if (h >= 0) {
                        ++dc[(size_t)(p / n) * kHeavyMax + h];
                        if (ts) ts[p] = (uint8_t)h;
                    }
                }
                out.any_dense = true;
            }
        }
        bx.uw_off = (int64_t)out.uwords.size();
        bx.num_distinct = (int32_t)touched.size();
        out.max_distinct = std::max(out.max_distinct, bx.num_distinct);
        if (want_slots) {
            out.uwords.insert(out.uwords.end(), touched.begin(), touched.end());
            for (size_t t = 0; t < touched.size(); ++t) slot_of[(size_t)touched[t]] = (int32_t)t;
            int32_t* sl = out.slots.data() + bi * T;
            for (int64_t p = 0; p < T; ++p) sl[p] = slot_of[(size_t)x[p]];
        }
        // exclusive offsets, stored back into count[] as write cursors
        start.resize(touched.size() + 1);
        int32_t acc = 0;
        for (size_t t = 0; t < touched.size(); ++t) {
            start[t] = acc;
            const int32_t c = count[touched[t]];
            count[touched[t]] = acc;   // cursor
            acc += c;
        }
        start[touched.size()] = acc;
        int32_t* rows = out.rows.data() + bx.rows_off;
        for (int64_t p = 0; p < T; ++p) {
            const int32_t wid = (int32_t)x[p];
            rows[count[wid]++] = (int32_t)(row_is_pos ? p : p / n);
        }
        for (int32_t wid : touched) count[wid] = 0;
        if (want_slots) {
            if (out.rows_div.size() != out.rows.size()) out.rows_div.resize(out.rows.size());
            int32_t* rd = out.rows_div.data() + bx.rows_off;
            for (int64_t e = 0; e < T; ++e) rd[e] = rows[e] / n;
        }

        // level 0 items
        struct Seg { int32_t begin, end, word, slot; bool dense; };
        std::vector<Seg> segs;
        segs.reserve(touched.size());
        int level = 0;
        int64_t part_base = 0;
        bx.row_groups = 1;
        if (row_groups > 1) {
            // ---- row-grouped level 0 (see above) ----
            struct GSeg { int32_t begin, end, t; };                       // entries [begin, end) of the group's own list
            std::vector<std::vector<int32_t>> grows((size_t)row_groups);
            std::vector<std::vector<GSeg>> gsegs((size_t)row_groups);
            std::vector<int32_t> nitems_of(touched.size(), 0);            // level-0 items per word
            for (size_t t = 0; t < touched.size(); ++t) {
                const bool dense = bx.dense_cnt > 0 && heavy_slot[(size_t)touched[t]] >= 0;
                if (dense) continue;                                       // summed densely, not by the tree
                int32_t e = start[t];
                while (e < start[t + 1]) {                                 // (a word's rows ascend: so do its groups)
                    const int g = rows[e] / rows_per_group;
                    std::vector<int32_t>& gr = grows[(size_t)g];
                    const int32_t b0 = (int32_t)gr.size();
                    while (e < start[t + 1] && rows[e] / rows_per_group == g) gr.push_back(rows[e++]);
                    gsegs[(size_t)g].push_back({b0, (int32_t)gr.size(), (int32_t)t});
                    nitems_of[t] += ((int32_t)gr.size() - b0 + kSegChunk - 1) / kSegChunk;
                }
            }
            for (int h = 0; h < bx.dense_cnt; ++h) heavy_slot[(size_t)bx.dense_word[h]] = (int8_t)-1;
            // partial rows of the words with more than one item, word-major
            std::vector<int32_t> pbase(touched.size(), -1);
            int32_t nparts = 0;
            for (size_t t = 0; t < touched.size(); ++t)
                if (nitems_of[t] > 1) {
                    pbase[t] = nparts;
                    segs.push_back({nparts, nparts + nitems_of[t], touched[t], (int32_t)t, false});   // level 1's input
                    nparts += nitems_of[t];
                }
            // the entries, group after group (a dense word's entries are simply absent)
            std::vector<int32_t> gbase((size_t)row_groups, 0);
            int32_t acc_e = 0;
            for (int g = 0; g < row_groups; ++g) {
                gbase[(size_t)g] = acc_e;
                std::copy(grows[(size_t)g].begin(), grows[(size_t)g].end(), rows + acc_e);
                acc_e += (int32_t)grows[(size_t)g].size();
            }
            // the items, XCD list after XCD list
            bx.item_off[0] = (int64_t)out.items.size();
            bx.part_off[0] = 0;
            std::vector<int32_t> cursor(touched.size(), 0);               // partial rows a word has handed out so far
            // (a word's partial rows must be numbered in GROUP order: number them in a first sweep over the groups)
            std::vector<std::vector<int32_t>> first_part((size_t)row_groups);
            for (int g = 0; g < row_groups; ++g) {
                first_part[(size_t)g].resize(gsegs[(size_t)g].size());
                for (size_t k = 0; k < gsegs[(size_t)g].size(); ++k) {
                    const GSeg& sg = gsegs[(size_t)g][k];
                    first_part[(size_t)g][k] = cursor[(size_t)sg.t];
                    cursor[(size_t)sg.t] += (sg.end - sg.begin + kSegChunk - 1) / kSegChunk;
                }
            }
            for (int x = 0; x < 8; ++x) {
                bx.xcd_off[x] = (int32_t)((int64_t)out.items.size() - bx.item_off[0]);
                for (int g = x; g < row_groups; g += 8)
                    for (size_t k = 0; k < gsegs[(size_t)g].size(); ++k) {
                        const GSeg& sg = gsegs[(size_t)g][k];
                        int32_t q = 0;
                        for (int32_t b = sg.begin; b < sg.end; b += kSegChunk, ++q) {
                            const int32_t e2 = std::min(sg.end, b + kSegChunk);
                            const int32_t dst = pbase[(size_t)sg.t] < 0 ? touched[(size_t)sg.t]
                                                                       : -(pbase[(size_t)sg.t] + first_part[(size_t)g][k] + q + 1);
                            out.items.push_back({gbase[(size_t)g] + b, gbase[(size_t)g] + e2, dst, (int32_t)sg.t});
                        }
                    }
                bx.xcd_cnt[x] = (int32_t)((int64_t)out.items.size() - bx.item_off[0]) - bx.xcd_off[x];
            }
            bx.item_cnt[0] = (int32_t)((int64_t)out.items.size() - bx.item_off[0]);
            bx.row_groups = row_groups;
            part_base = nparts;
            level = 1;
        } else {
        for (size_t t = 0; t < touched.size(); ++t) {
            const bool dense = bx.dense_cnt > 0 && heavy_slot[(size_t)touched[t]] >= 0;
            if (dense && !flag_only) continue;   // summed densely, not by the tree
            segs.push_back({start[t], start[t + 1], touched[t], (int32_t)t, dense});
        }
        for (int h = 0; h < bx.dense_cnt; ++h) heavy_slot[(size_t)bx.dense_word[h]] = (int8_t)-1;
        }
        while (!segs.empty() && level < kSegMaxLevels) {
            bx.item_off[level] = (int64_t)out.items.size();
            bx.part_off[level] = part_base;
            std::vector<Seg> next;
            int32_t nparts = 0;
            for (const Seg& s : segs) {
                const int32_t len = s.end - s.begin;
                if (len <= kSegChunk || level == kSegMaxLevels - 1) {
                    out.items.push_back({s.begin, s.end, s.word, s.slot});
                } else {
                    const int32_t first = nparts;
                    for (int32_t b = s.begin; b < s.end; b += kSegChunk) {
                        const int32_t e = std::min(s.end, b + kSegChunk);
                        out.items.push_back({b, e, -(nparts + 1), s.dense ? -1 : 0});
                        ++nparts;
                    }
                    next.push_back({first, nparts, s.word, s.slot, s.dense});
                }
            }
            bx.item_cnt[level] = (int32_t)((int64_t)out.items.size() - bx.item_off[level]);
            part_base += nparts;
            segs.swap(next);
            ++level;
        }
        bx.nlevels = level;
        bx.part_rows = part_base;
        bx.slot_is_row = false;
        if (sort_level0 && !want_slots && !row_is_pos && bx.row_groups == 1 && level >= 1 && bx.item_cnt[0] > 0) {
            // (an item's destination -- a table row or a numbered partial row -- does not depend on where the item stands)
            SegItem* l0 = out.items.data() + bx.item_off[0];
            for (int32_t k = 0; k < bx.item_cnt[0]; ++k) l0[k].slot = rows[l0[k].begin];
            std::stable_sort(l0, l0 + bx.item_cnt[0], [](const SegItem& a, const SegItem& b) { return a.end - a.begin > b.end - b.begin; });
            bx.slot_is_row = true;
        }
        bx.heavy_off = (int64_t)out.heavy.size() / 4;
        bx.heavy_cnt = 0;
        bx.fused_upper_ok = false;
        if (level == 3) {
            // level-2 items (all final) own consecutive runs of level-1 chunk items, in the same order
            const SegItem* l1 = out.items.data() + bx.item_off[1];
            const SegItem* l2 = out.items.data() + bx.item_off[2];
            bool ok = true;
            int32_t i1 = 0;
            for (int32_t j = 0; j < bx.item_cnt[2] && ok; ++j) {
                const SegItem& parent = l2[j];
                while (i1 < bx.item_cnt[1] && l1[i1].dst >= 0) ++i1;          // skip the final level-1 items
                const int32_t first = i1;
                int32_t n = 0;
                while (i1 < bx.item_cnt[1] && l1[i1].dst < 0 && -(l1[i1].dst + 1) < parent.end) {
                    ok = ok && (-(l1[i1].dst + 1) == parent.begin + n);        // its partial rows, in order
                    ++i1; ++n;
                }
                ok = ok && parent.dst >= 0 && n == parent.end - parent.begin && n >= 1 && n <= kFusedMaxChunks;
                out.heavy.push_back(first); out.heavy.push_back(n); out.heavy.push_back(parent.dst); out.heavy.push_back(0);
            }
            bx.heavy_cnt = bx.item_cnt[2];
            bx.fused_upper_ok = ok;
        }
        if (part_base > out.max_part_rows) out.max_part_rows = part_base;
        // level-0 bundles (see BatchIndex): only where the items partition ONE consecutive entry range in order
        bx.bundle_off = (int64_t)out.bundles.size();
        bx.bundle_cnt = 0;
        if (bx.row_groups == 1 && bx.nlevels >= 1 && bx.item_cnt[0] > 0) {
            const SegItem* l0 = out.items.data() + bx.item_off[0];
            bool consecutive = true;
            for (int32_t k = 1; k < bx.item_cnt[0] && consecutive; ++k) consecutive = l0[k].begin == l0[k - 1].end;
            if (consecutive) {
                int32_t in_bundle = 0, entries = 0;
                for (int32_t k = 0; k < bx.item_cnt[0]; ++k) {
                    const int32_t len = l0[k].end - l0[k].begin;
                    if (in_bundle == 0 || in_bundle == kBundleItems || entries + len > kSegChunk) {
                        out.bundles.push_back(k);
                        ++bx.bundle_cnt;
                        in_bundle = 0;
                        entries = 0;
                    }
                    ++in_bundle;
                    entries += len;
                }
                out.bundles.push_back(bx.item_cnt[0]);
            }
        }
    }
    return true;
}

}  // namespace sert

// Data-parallel exchange of the word table BY ROWS (new: the reference is single-device).
//
// Rank q owns rows [q R, (q+1) R) of R_w for good: their Adam / Adadelta state lives there and nowhere
// else, and the reference's dense update (L2 + optimiser on every row every step, sert/models.py:764-795,
// :548-549) runs there.  A training batch is a static slice of the data set, so WHICH rows a rank's batch
// touches is known at upload (word_index.h: touched_bits).  The bitmaps of all ranks are exchanged once,
// and from them every rank derives, per batch, the same lists in the same (ascending row) order:
//
//   serve[r]  rows I own that rank r's batch touches   -> I send r their PARAMETERS before its forward,
//                                                          r sends me their GRADIENT rows after its backward
//   fetch[q]  rows rank q owns that my batch touches   -> the mirror image
//
// so a step moves two all-to-alls of touched rows -- no index travels, no row nobody touches travels --
// instead of a reduce-scatter plus an all-gather of the whole table (ZeRO-1, kept as the fallback):
// at C2 on 8 ranks 2 x 23 MB per rank and step instead of 2 x 45 MB.  The owner adds the gradient rows
// it receives to its own in RANK order (a fixed association: the result does not depend on arrival
// order) and updates all of its rows; rows no rank touched take a zero gradient without reading it.
#pragma once
#include <algorithm>
#include <vector>
#include "common.h"

namespace sert {

// out[i, :] = table[rows[i], :]   (d4 float4 per row; one 16-lane group per row and 16 columns)
__global__ __launch_bounds__(256) void xchg_pack_rows(const float* __restrict__ table, const int32_t* __restrict__ rows,
                                                      int nrows, int d4, float4* __restrict__ out) {
    const int64_t total = (int64_t)nrows * d4;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(t / d4), c = (int)(t - (int64_t)i * d4);
        out[t] = reinterpret_cast<const float4*>(table)[(size_t)rows[i] * d4 + c];
    }
}

// table[rows[i], :] = in[i, :]
__global__ __launch_bounds__(256) void xchg_unpack_rows(float* __restrict__ table, const int32_t* __restrict__ rows,
                                                        int nrows, int d4, const float4* __restrict__ in) {
    const int64_t total = (int64_t)nrows * d4;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(t / d4), c = (int)(t - (int64_t)i * d4);
        reinterpret_cast<float4*>(table)[(size_t)rows[i] * d4 + c] = in[t];
    }
}

// g[urows[u], :] = sum over the contributions ent[ptr[u] .. ptr[u+1]) of row u, IN THAT ORDER (ascending
// source rank): ent >= 0 = row of the receive buffer, ent < 0 = this rank's own gradient row (already in g).
__global__ __launch_bounds__(256) void xchg_reduce_rows(const float4* __restrict__ recv, const int32_t* __restrict__ ptr,
                                                        const int32_t* __restrict__ ent, const int32_t* __restrict__ urows,
                                                        int nu, int d4, float* __restrict__ g) {
    const int64_t total = (int64_t)nu * d4;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int u = (int)(t / d4), c = (int)(t - (int64_t)u * d4);
        float4* dst = reinterpret_cast<float4*>(g) + (size_t)urows[u] * d4 + c;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = ptr[u]; e < ptr[u + 1]; ++e) {
            const int src = ent[e];
            const float4 v = src >= 0 ? recv[(size_t)src * d4 + c] : *dst;
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        *dst = a;
    }
}

// ---- the static lists (host side, built once per uploaded training split) -----------------------------
struct RowExchangeBatch {
    // offsets into the concatenated device arrays below
    int64_t serve_off = 0, fetch_off = 0, union_off = 0, ent_off = 0, ptr_off = 0;
    int32_t serve_total = 0, fetch_total = 0, nunion = 0, nent = 0;
    std::vector<int32_t> serve_cnt, fetch_cnt;   // [world] rows per peer (0 for this rank)
};

struct RowExchangeLists {
    std::vector<RowExchangeBatch> batches;
    std::vector<int32_t> serve_rows, fetch_rows, union_rows, ent, ptr;
    std::vector<uint32_t> union_bits;   // per batch: owned_bit_words words, bit j = local row j touched by any rank
    int64_t owned_bit_words = 0;
    int32_t max_xfer_rows = 0;          // largest serve_total / fetch_total of any batch
};

// Appends the lists of `num_batches` consecutive batches to `out` (called once per group of batches, so
// that the gathered bitmaps of a long data set never exist all at once).
// allbits: [world][num_batches][bit_words] touched bitmaps of every rank for these batches (rank-major).
inline void build_row_exchange(const uint32_t* allbits, int world, int rank, int64_t num_batches, int64_t bit_words,
                               int64_t rows_per_rank, int64_t vocab, RowExchangeLists& out) {
    const size_t first = out.batches.size();
    out.batches.resize(first + (size_t)num_batches);
    out.owned_bit_words = ((rows_per_rank + 31) / 32 + 3) / 4 * 4;
    out.union_bits.resize((first + (size_t)num_batches) * (size_t)out.owned_bit_words, 0u);
    auto words_of = [&](int r, int64_t b) -> const uint32_t* { return allbits + ((size_t)r * num_batches + b) * bit_words; };
    // calls f(row) for every set bit of `bits` inside [lo, hi), ascending (cost ~ words + set bits)
    auto for_each_bit = [&](const uint32_t* bits, int64_t lo, int64_t hi, auto&& f) {
        for (int64_t wd = lo >> 5; wd <= (hi - 1) >> 5 && lo < hi; ++wd) {
            uint32_t x = bits[wd];
            while (x) {
                const int k = __builtin_ctz(x);
                x &= x - 1;
                const int64_t w = (wd << 5) + k;
                if (w >= lo && w < hi) f(w);
            }
        }
    };
    const int64_t my_lo = std::min(vocab, (int64_t)rank * rows_per_rank), my_hi = std::min(vocab, my_lo + rows_per_rank);
    std::vector<int32_t> pos((size_t)world);
    std::vector<uint32_t> uni;
    for (int64_t b = 0; b < num_batches; ++b) {
        RowExchangeBatch& xb = out.batches[first + (size_t)b];
        xb.serve_cnt.assign((size_t)world, 0);
        xb.fetch_cnt.assign((size_t)world, 0);
        xb.serve_off = (int64_t)out.serve_rows.size();
        xb.fetch_off = (int64_t)out.fetch_rows.size();
        xb.union_off = (int64_t)out.union_rows.size();
        xb.ent_off = (int64_t)out.ent.size();
        xb.ptr_off = (int64_t)out.ptr.size();
        // serve lists: peer-major, ascending row; a row's slot inside the receive buffer of the gradient
        // phase = base of its peer's segment (segments in ascending peer order) + its rank in the list
        std::vector<int32_t> seg_base((size_t)world, 0);
        for (int r = 0; r < world; ++r) {
            if (r == rank) continue;
            for_each_bit(words_of(r, b), my_lo, my_hi, [&](int64_t w) { out.serve_rows.push_back((int32_t)w); ++xb.serve_cnt[(size_t)r]; });
        }
        {
            int32_t acc = 0;
            for (int r = 0; r < world; ++r) { seg_base[(size_t)r] = acc; acc += xb.serve_cnt[(size_t)r]; }
            xb.serve_total = acc;
        }
        // fetch lists: owner-major, ascending row
        for (int q = 0; q < world; ++q) {
            if (q == rank) continue;
            const int64_t lo = std::min(vocab, (int64_t)q * rows_per_rank), hi = std::min(vocab, lo + rows_per_rank);
            for_each_bit(words_of(rank, b), lo, hi, [&](int64_t w) { out.fetch_rows.push_back((int32_t)w); ++xb.fetch_cnt[(size_t)q]; });
        }
        xb.fetch_total = 0;
        for (int q = 0; q < world; ++q) xb.fetch_total += xb.fetch_cnt[(size_t)q];
        // union of the owned rows anyone touches, with their contributions in rank order
        uni.assign((size_t)bit_words, 0u);
        for (int r = 0; r < world; ++r) {
            const uint32_t* bw = words_of(r, b);
            for (int64_t wd = my_lo >> 5; wd <= (my_hi - 1) >> 5 && my_lo < my_hi; ++wd) uni[(size_t)wd] |= bw[wd];
        }
        std::fill(pos.begin(), pos.end(), 0);
        uint32_t* ub = out.union_bits.data() + (first + (size_t)b) * (size_t)out.owned_bit_words;
        for_each_bit(uni.data(), my_lo, my_hi, [&](int64_t w) {
            out.union_rows.push_back((int32_t)w);
            out.ptr.push_back((int32_t)(out.ent.size() - (size_t)xb.ent_off));
            for (int r = 0; r < world; ++r) {
                if (!((words_of(r, b)[w >> 5] >> (w & 31)) & 1u)) continue;
                if (r == rank) out.ent.push_back(-1);
                else out.ent.push_back(seg_base[(size_t)r] + pos[(size_t)r]++);
            }
            const int64_t j = w - my_lo;
            ub[j >> 5] |= 1u << (j & 31);
        });
        out.ptr.push_back((int32_t)(out.ent.size() - (size_t)xb.ent_off));
        xb.nunion = (int32_t)(out.union_rows.size() - (size_t)xb.union_off);
        xb.nent = (int32_t)(out.ent.size() - (size_t)xb.ent_off);
        out.max_xfer_rows = std::max(out.max_xfer_rows, std::max(xb.serve_total, xb.fetch_total));
    }
}

}  // namespace sert

// Stable key/value radix sort of the (entity, pair) keys (hipCUB / rocPRIM).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

namespace sert {

size_t sort_pairs_temp_bytes(int n, int end_bit) {
    size_t bytes = 0;
    int32_t* p = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, p, p, p, p, n, 0, end_bit, nullptr);
    return bytes;
}

int sort_pairs(void* tmp, size_t tmp_bytes, const int32_t* keys_in, int32_t* keys_out,
               const int32_t* vals_in, int32_t* vals_out, int n, int end_bit, hipStream_t s) {
    return (int)hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys_in, keys_out, vals_in,
                                                   vals_out, n, 0, end_bit, s);
}

}  // namespace sert

// Key/value radix sort used by the clustered intersect path (ray order per pass).
// rocPRIM is header-only; it lives in its own translation unit to keep the other files'
// compile times short.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>

namespace tfrt {

size_t sort_pairs_temp_bytes(size_t n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                  (const int32_t*)nullptr, (int32_t*)nullptr, n, 0, 32,
                                  (hipStream_t)0);
  return bytes;
}

int sort_pairs_u32_i32(void* tmp, size_t bytes, const uint32_t* keys_in, uint32_t* keys_out,
                       const int32_t* vals_in, int32_t* vals_out, size_t n, int end_bit,
                       hipStream_t st) {
  return (int)rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0,
                                        (unsigned)end_bit, st);
}

}  // namespace tfrt

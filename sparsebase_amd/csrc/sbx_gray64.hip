// sbx_gray64.hip — sbx_gray.hip once more, for 64-bit index arrays: row_ptr, col and degree_out are read and written as
// int64 (X in that file; four columns = two 16-byte loads), everything inside stays as it is.  reorder/gray_reorder.cc:106-424
// for the <int64, int64, ...> type tuples the reference pre-instantiates (CMakeLists.txt:15-16).
#define SBX_GRAY_I64 1
#include "sbx_gray.hip"

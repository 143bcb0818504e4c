// sbx_rcm64.hip — sbx_rcm.hip once more, for 64-bit index arrays: row_ptr, col and the inverse permutation are read and
// written as int64 (X in that file), everything inside stays as it is.  reorder/rcm_reorder.cc:22-166 for the
// <int64, int64, ...> type tuples the reference pre-instantiates (CMakeLists.txt:15-16).
#define SBX_RCM_I64 1
#include "sbx_rcm.hip"

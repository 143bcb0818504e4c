// Umbrella header of the MI355X-native reorder / convert / permute path.
#ifndef SPARSEBASE_SPARSEBASE_H_
#define SPARSEBASE_SPARSEBASE_H_
#include "sparsebase/bases/reorder_base.h"
#include "sparsebase/context/cpu_context.h"
#include "sparsebase/context/hip_context.h"
#include "sparsebase/converter/converter_order_one.h"
#include "sparsebase/converter/converter_order_two.h"
#include "sparsebase/format/array.h"
#include "sparsebase/feature/bandwidth.h"
#include "sparsebase/feature/degree_distribution.h"
#include "sparsebase/feature/degrees.h"
#include "sparsebase/feature/profile.h"
#include "sparsebase/format/coo.h"
#include "sparsebase/format/csc.h"
#include "sparsebase/format/csr.h"
#include "sparsebase/format/hip_formats.h"
#include "sparsebase/utils/logger.h"
// last: the reader needs the complete conversion graph
#include "sparsebase/bases/iobase.h"

namespace sparsebase {
// pre-0.3 spelling used by north_star / older call sites (SURVEY.md "Naming drift")
namespace preprocess {
template <typename IDType>
using Reorder = reorder::Reorderer<IDType>;
template <typename IDType>
using ReorderPreprocessType = reorder::Reorderer<IDType>;
template <typename I, typename N, typename V>
using RCMReorder = reorder::RCMReorder<I, N, V>;
template <typename I, typename N, typename V>
using DegreeReorder = reorder::DegreeReorder<I, N, V>;
template <typename I, typename N, typename V>
using GrayReorder = reorder::GrayReorder<I, N, V>;
}  // namespace preprocess
}  // namespace sparsebase
#endif

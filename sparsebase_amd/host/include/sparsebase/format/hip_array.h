#ifndef SPARSEBASE_FORMAT_HIP_ARRAY_H_
#define SPARSEBASE_FORMAT_HIP_ARRAY_H_
#include "sparsebase/format/format_order_one.h"
#endif

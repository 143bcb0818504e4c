#ifndef SPARSEBASE_FORMAT_HIP_COO_H_
#define SPARSEBASE_FORMAT_HIP_COO_H_
#include "sparsebase/format/coo.h"
#endif

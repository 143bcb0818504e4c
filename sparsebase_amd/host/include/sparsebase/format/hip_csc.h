#ifndef SPARSEBASE_FORMAT_HIP_CSC_H_
#define SPARSEBASE_FORMAT_HIP_CSC_H_
#include "sparsebase/format/hip_formats.h"
#endif

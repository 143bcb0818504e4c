#ifndef SPARSEBASE_FORMAT_HIP_CSR_H_
#define SPARSEBASE_FORMAT_HIP_CSR_H_
#include "sparsebase/format/csr.h"
#endif

// CPUContext lives next to its base in context.h; this header keeps the reference's include path working.
#ifndef SPARSEBASE_CONTEXT_CPU_CONTEXT_H_
#define SPARSEBASE_CONTEXT_CPU_CONTEXT_H_
#include "sparsebase/context/context.h"
#endif

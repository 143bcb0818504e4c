// sparsebase/config.h — build switches of the MI355X-native host layer.
// Mirrors the option header the reference generates (src/sparsebase/config.h.in:4-11):
// header-only, HIP device path on, no optional third-party orderings.
#ifndef SPARSEBASE_CONFIG_H_
#define SPARSEBASE_CONFIG_H_
#define _HEADER_ONLY
#define USE_HIP
#endif
